#!/usr/bin/env python3
"""profiles/hbm_traffic.json entry for one workload from the FETCH_SIZE and WRITE_SIZE rocprofv3 passes of bench.py.
usage: pmc_traffic_json.py <scene@WxH> <fetch pass dir> <write pass dir> <bench json of a pass> <source label> [limited_by text]
HBM bytes of the extend kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE doubled for gfx950, MI355X_MICROARCH.md
'HBM'), divided by the extend rays of the profiled run (warm-up + timed steps: every launch is counted by the profiler)."""
import collections, csv, glob, json, os, sys

key, fetch_dir, write_dir, bench_json, source = sys.argv[1:6]
limited = sys.argv[6] if len(sys.argv) > 6 else None


def total(d, counter, kernel_part):
    t, calls = 0.0, set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_part in r["Kernel_Name"] and r["Counter_Name"] == counter:
                t += float(r["Counter_Value"])
                calls.add(r["Dispatch_Id"])
    return t, len(calls)


fetch, n1 = total(fetch_dir, "FETCH_SIZE", "k_traverse<false")
write, n2 = total(write_dir, "WRITE_SIZE", "k_traverse<false")
b = json.loads(open(bench_json).read().strip().splitlines()[-1])
steps, warm = b["steps"], b["warmup"]
rays = (b["extend_rays"] - b.get("culled_camera_rays", 0)) * (steps + warm) / steps  # rays the kernel traversed; the profiler saw the warm-up step too
per_ray = (2 * fetch + write) * 1024 / rays
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "hbm_traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
entry = {"extend_bytes_per_ray": round(per_ray, 1), "source": source,
         "fetch_size_kb": fetch, "write_size_kb": write, "extend_launches_profiled": n1, "extend_rays_profiled": rays}
if limited:
    entry["limited_by"] = limited
data[key] = entry
json.dump(data, open(path, "w"), indent=2)
print(key, entry)
