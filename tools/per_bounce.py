#!/usr/bin/env python3
"""Per-bounce ray counts and traversal kernel times of one batch.  Run under
rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/per_bounce.py [scene] [spp], then
python3 tools/per_bounce.py --parse DIR to print the table."""
import os, sys, glob, csv, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    meta = json.load(open(os.path.join(sys.argv[2], "counts.json")))
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    B = len(meta["extend"])
    for key, tag, counts in (("k_traverse<false", "extend", meta["extend"]), ("k_traverse<true", "shadow", meta["shadow"]),
                             ("k_shade", "shade", meta["extend"]), ("k_resolve", "resolve", meta["shadow"])):
        ks = [r for r in rows if key in r["Kernel_Name"]][-B:]
        tot = 0.0
        for b, r in enumerate(ks):
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            tot += ms
            print("%-8s bounce %2d  %10d  %8.3f ms  %7.2f G/s" % (tag, b + 1, counts[b], ms, counts[b] / ms / 1e6 if ms else 0))
        print("%-8s total %.3f ms" % (tag, tot))
    sys.exit(0)
import numpy as np, torch
import pathtracer_amd as pt
scene = pt.LoadedScene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "%s.obj" % (sys.argv[1] if len(sys.argv) > 1 else "cornell_teapot")))
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = scene.upload()
ph = scene.pinhole(1080, 1080)
acc = torch.zeros(1080 * 1080 * 3, device="cuda")
for i in range(2):
    dev.render_device(ph, spp, acc.data_ptr(), first_sample=spp * i)
torch.cuda.synchronize()
e, s = dev.last_batch_counts()
out = os.environ.get("PB_OUT", ".")
json.dump({"extend": e.tolist(), "shadow": (s * max(scene.desc.n_lights, 1)).tolist()}, open(os.path.join(out, "counts.json"), "w"))
