#!/usr/bin/env python3
"""Build-time guard on the traversal kernel's registers (hipcc -Rpass-analysis=kernel-resource-usage output on stdin or
in the file given).  Two 1024-thread workgroups of k_traverse share a CU only with <= 64 VGPRs, <= 80 SGPRs, no spills
and <= 80 KB of LDS per workgroup; one register more halves the resident waves and every launch takes ~1.45x as long
(DESIGN.md 5, round 3), while the compiler's own "Occupancy" remark still says 8.  Exit status 1 when the product
kernels (the non-counting instantiations) break a limit; `--warn` only reports (probe builds with -D switches)."""
import re
import sys

LIMITS = {"TotalSGPRs": 80, "VGPRs": 64, "SGPRs Spill": 0, "VGPRs Spill": 0, "LDS Size [bytes/block]": 81920}
args = [a for a in sys.argv[1:] if a != "--warn"]
text = open(args[0]).read() if args else sys.stdin.read()
bad, cur, seen = [], None, 0
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        continue
    if not cur or not re.search(r"k_traverseILb[01]ELb0E", cur):  # <ANY, COUNT = false>
        continue
    for key, limit in LIMITS.items():
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m:
            seen += 1
            if int(m.group(1)) > limit:
                bad.append("%s: %s = %s > %d" % (cur, key, m.group(1), limit))
for line in text.splitlines():  # whatever the compiler said besides the remarks
    if "remark:" not in line and line.strip() and not line.lstrip().startswith(("|", "^")) and not re.match(r"^\s*\d+ \|", line):
        print(line, file=sys.stderr)
if seen < 2 * len(LIMITS):
    print("check_resources: the traversal kernels were not found in the compiler's remarks", file=sys.stderr)
    sys.exit(0 if "--warn" in sys.argv else 1)
for b in bad:
    print("check_resources: " + b, file=sys.stderr)
sys.exit(1 if bad and "--warn" not in sys.argv else 0)
