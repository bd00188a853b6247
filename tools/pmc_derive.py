#!/usr/bin/env python3
"""Derived figures per kernel from a tools/pmc_summary.py dump (the passes of tools/pmc_passes.sh).
usage: pmc_derive.py <summary.txt> [kernel substring ...]      formulas as in profiles/r02_v20_pmc_summary.txt"""
import re
import sys

acc = {}
for line in open(sys.argv[1]):
    m = re.match(r"(\S.*?)\s+calls=(\d+) (.*)", line)
    if not m:
        continue
    k = acc.setdefault(m.group(1).strip(), {})
    k.setdefault("calls", int(m.group(2)))
    for kv in m.group(3).split():
        if "=" in kv:
            name, v = kv.split("=")
            k.setdefault(name, float(v))  # the first pass that holds a counter wins
want = sys.argv[2:] or ["k_traverse<false, false>", "k_traverse<true, false>", "k_shade"]
for name, c in acc.items():
    if not any(w in name for w in want):
        continue
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8.0  # summed over the 8 XCDs
    out = ["%-26s calls %d, kernel cycles %.4g" % (name, c["calls"], g)]
    if "SQ_INSTS_VALU" in c and g:
        out.append("VALU busy %.1f %%" % (100 * 4 * c["SQ_INSTS_VALU"] / (1024 * g)))
        out.append("lane utilisation %.1f %%" % (100 * c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_INSTS_VALU"])))
    if "SQ_WAIT_ANY" in c:
        out.append("waves waiting %.1f %%" % (100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
        out.append("resident waves %.0f" % (4 * c["SQ_WAVE_CYCLES"] / g / c["calls"] * c["calls"]))
    if "TCC_HIT_sum" in c:
        out.append("L2 hit %.1f %%" % (100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
    if "TCP_TCC_READ_REQ_sum" in c and "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        out.append("L1 miss %.1f %%" % (100 * c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]))
    if "TCP_TCC_READ_REQ_LATENCY_sum" in c:
        out.append("L1->L2 latency %.0f cycles" % (c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]))
    if "TA_TA_BUSY_sum" in c and g:
        out.append("TA busy %.1f %%, TCP stalled on pending %.1f %%" % (100 * c["TA_TA_BUSY_sum"] / (256 * g), 100 * c["TCP_PENDING_STALL_CYCLES_sum"] / (256 * g)))
    if "FETCH_SIZE" in c:
        out.append("HBM bytes (2 x FETCH + WRITE) %.4g GB" % ((2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0)) * 1024 / 1e9))
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"):
        if n in c:
            out.append("%s %.4g" % (n, c[n]))
    print("; ".join(out))
