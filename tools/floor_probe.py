#!/usr/bin/env python3
"""Launch-size sweep of the flat extend entry point on late-bounce rays (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import pyoracle as orc
import pathtracer_amd as pt
path = os.path.join(ROOT, "scenes", "cornell_teapot.obj")
ref = orc.Scene(path)
rays, _ = ref.bounce_rays(512, 512, bounces=10, seed=3, threads=16)
rng = np.random.default_rng(0)
rays = rays[rng.permutation(len(rays))]
print("rays", len(rays))
dev = pt.LoadedScene(path).upload()
d_rays = torch.from_numpy(rays.view(np.uint8).copy()).cuda()
d_hits = torch.zeros(len(rays) * 16, dtype=torch.uint8, device="cuda")
for n in (64, 1000, 10000, 100000, 500000, 1000000, 2000000, len(rays)):
    n = min(n, len(rays))
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dev.intersect_device(d_rays.data_ptr(), n, d_hits.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("n %8d  min %.3f ms  median %.3f ms  %.2f Grays/s" % (n, min(ts), sorted(ts)[2], n / min(ts) / 1e6))
