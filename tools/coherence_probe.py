#!/usr/bin/env python3
"""Does ordering bounce-2 rays by direction help the extend kernel?  (diagnostic, flat entry point)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import pyoracle as orc
import pathtracer_amd as pt
path = os.path.join(ROOT, "scenes", "cornell_teapot.obj")
ref = orc.Scene(path)
rays, _ = ref.bounce_rays(1080, 1080, bounces=3, seed=3, threads=16)
ext = rays[rays["tmax"] > 2.0]
cam = ext[0]["o"] if "o" in ext.dtype.names else None
names = ext.dtype.names
print(names)
o = np.stack([ext[names[0]], ext[names[1]], ext[names[2]]], 1) if len(names) == 8 else ext["o"]
d = np.stack([ext[names[4]], ext[names[5]], ext[names[6]]], 1) if len(names) == 8 else ext["d"]
is_cam = (o == o[0]).all(1)
idx = np.arange(len(ext)); start = np.maximum.accumulate(np.where(is_cam, idx, 0)); bounce = idx - start
dev = pt.LoadedScene(path).upload()
def run(tag, r):
    d_rays = torch.from_numpy(np.ascontiguousarray(r).view(np.uint8).copy()).cuda()
    d_hits = torch.zeros(len(r) * 16, dtype=torch.uint8, device="cuda")
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); dev.intersect_device(d_rays.data_ptr(), len(r), d_hits.data_ptr(), stream=torch.cuda.current_stream().cuda_stream); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("%-34s n %8d  %.3f ms  %.2f Grays/s" % (tag, len(r), min(ts), len(r) / min(ts) / 1e6))
for bn in (0, 1, 2):
    sel = bounce == bn
    r, oo, dd = ext[sel], o[sel], d[sel]
    r = np.tile(r, 32); oo = np.tile(oo, (32, 1)); dd = np.tile(dd, (32, 1))  # ~37 M rays
    run("bounce %d pixel order" % (bn + 1), r)
    octant = (dd[:, 0] < 0) * 1 + (dd[:, 1] < 0) * 2 + (dd[:, 2] < 0) * 4
    run("bounce %d by octant (stable)" % (bn + 1), r[np.argsort(octant, kind="stable")])
    # octant within blocks of 64 Ki rays (what a per-segment binning could do)
    blk = np.arange(len(r)) // 65536
    run("bounce %d by octant within 64Ki" % (bn + 1), r[np.lexsort((octant, blk))])
    for chunk in (256, 1024, 4096):  # what a bucketing inside one k_shade workgroup pass could do
        blk = np.arange(len(r)) // chunk
        run("bounce %d by octant within %d" % (bn + 1, chunk), r[np.lexsort((octant, blk))])
    ax = np.abs(dd).argmax(1) * 2 + (dd[np.arange(len(dd)), np.abs(dd).argmax(1)] < 0)
    run("bounce %d by major axis (6 bins)" % (bn + 1), r[np.argsort(ax, kind="stable")])
    run("bounce %d random" % (bn + 1), r[np.random.default_rng(0).permutation(len(r))])
