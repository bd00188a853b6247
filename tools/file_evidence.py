#!/usr/bin/env python3
"""Files the outputs of tools/r3_evidence.sh under profiles/ (the judged copies): file_evidence.py <gpurun_out tag> <old profiles tag> <new profiles tag>
e.g. file_evidence.py r03_v29e r03_v28 r03_v29.  Renames profiles/<old>_* to <new>_* (git mv), overwrites them with the new outputs,
regenerates the A / B sections of the PMC summary and profiles/hbm_traffic.json (keeping its limited_by texts), and rewrites the
<old>_ references in the documents.  The numbers quoted in DESIGN.md / README.md / BASELINE.md stay a manual edit."""
import glob, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, old, new = sys.argv[1:4]
E = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")
if old != new:
    for f in sorted(glob.glob(os.path.join(P, old + "_*"))):
        subprocess.check_call(["git", "mv", f, os.path.join(P, new + os.path.basename(f)[len(old):])], cwd=ROOT)
copies = {"bench.json": "bench.json", "trace_bench.json": "bench_under_rocprof.json", "kernel_stats.csv": "kernel_stats.csv",
          "pmc_traffic.txt": "pmc_fetch_write.txt", "pytest_gpu.log": "pytest_gpu.log", "rows.jsonl": "rows_gpu.jsonl",
          "per_round_teapot_640spp.txt": "per_round_teapot_640spp.txt", "per_round_sr2_128spp.txt": "per_round_sr2_128spp.txt",
          "launch_floor.txt": "launch_floor.txt", "bench_force_dist.json": "bench_force_dist_1rank_nccl.json",
          "bench_inproc1.json": "bench_in_process_1gpu.json", "bench_inproc2_shared.json": "bench_in_process_2handles_shared_gpu.json",
          "pmc_teapot_summary.txt": "pmc_teapot_raw.txt", "pmc_sr2_summary.txt": "pmc_sr2_raw.txt"}
for src, dst in copies.items():
    shutil.copy(os.path.join(E, src), os.path.join(P, "%s_%s" % (new, dst)))
derive = lambda f: "".join(l + "\n" for l in subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_derive.py"), f], text=True).splitlines()
                            if "k_traverse" in l or "k_shade" in l)
sp = os.path.join(P, new + "_pmc_summary.txt")
s = open(sp).read().replace(old + "_", new + "_")
i, j = s.index("# ---- A:"), s.index("# ---- C:")
ver = new.split("_")[-1]
s = (s[:i] + "# ---- A: cornell_teapot 1080x1080, %s\n" % ver + derive(os.path.join(P, new + "_pmc_teapot_raw.txt"))
     + "# ---- B: sr2 3840x2160, %s\n" % ver + derive(os.path.join(P, new + "_pmc_sr2_raw.txt")) + s[j:])
s = re.sub(r"of the v\d+ kernels", "of the %s kernels" % ver, s, count=1)
open(sp, "w").write(s)
tj = os.path.join(P, "hbm_traffic.json")
cur, fresh = json.load(open(tj)), json.load(open(os.path.join(E, "hbm_traffic.json")))
for k, v in fresh.items():
    if isinstance(v, dict):
        v["source"] = v.get("source", "").replace(tag + "_pmc_sr2_summary", new + "_pmc_summary").replace(tag + "_", new + "_")
        if k in cur and "limited_by" in cur[k]:
            v["limited_by"] = cur[k]["limited_by"].replace(old + "_", new + "_")
json.dump(fresh, open(tj, "w"), indent=1)
for doc in ("DESIGN.md", "README.md", "BASELINE.md", "INTEGRATION.md"):
    p = os.path.join(ROOT, doc)
    t = open(p).read()
    if old + "_" in t:
        open(p, "w").write(t.replace(old + "_", new + "_"))
print("filed", tag, "as", new)
