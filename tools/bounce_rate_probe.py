#!/usr/bin/env python3
"""Are the late bounces slow because their launches are small, or because their rays are expensive?
The oracle's extend rays of a 10-bounce render, split by bounce; every bounce's set is tiled to the same ~32 M rays and
timed through the flat entry point, so the launch size is the same for all of them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import pyoracle as orc
import pathtracer_amd as pt
path = os.path.join(ROOT, "scenes", "cornell_teapot.obj")
ref = orc.Scene(path)
rays, _ = ref.bounce_rays(1080, 1080, bounces=10, seed=3, threads=16)
ext = rays[rays["tmax"] > 2.0]
o = ext["o"]
is_cam = (o == o[0]).all(1)
idx = np.arange(len(ext)); start = np.maximum.accumulate(np.where(is_cam, idx, 0)); bounce = idx - start
dev = pt.LoadedScene(path).upload(flags=pt.FLAG_COUNT_WORK if "--count" in sys.argv else 0)
def run(tag, r):
    d_rays = torch.from_numpy(np.ascontiguousarray(r).view(np.uint8).copy()).cuda()
    d_hits = torch.zeros(len(r) * 16, dtype=torch.uint8, device="cuda")
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); dev.intersect_device(d_rays.data_ptr(), len(r), d_hits.data_ptr(), stream=torch.cuda.current_stream().cuda_stream); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("%-26s original %8d rays, timed %9d  %.3f ms  %.2f Grays/s" % (tag, tag_n, len(r), min(ts), len(r) / min(ts) / 1e6), flush=True)
for bn in range(10):
    r = ext[bounce == bn]
    tag_n = len(r)
    if tag_n == 0:
        continue
    reps = max(1, (1 << 25) // tag_n)
    run("bounce %d" % (bn + 1), np.tile(r, reps))
    if bn in (5, 7, 9):
        run("bounce %d untiled" % (bn + 1), r)
