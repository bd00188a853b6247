#!/usr/bin/env python3
"""How long does one traversal launch take as a function of its ray count?  ptx_intersect_batch_device on camera-like
rays, HIP events around 20 launches each.  usage: python tools/launch_floor.py [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pathtracer_amd as pt

name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
scene = pt.LoadedScene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", name + ".obj"))
dev = scene.upload()
ph = scene.pinhole(2048, 2048)
pos = np.array(ph.position[:], np.float32); plane = np.array(ph.plane[:], np.float32)
dx = np.array(ph.dx[:], np.float32); dy = np.array(ph.dy[:], np.float32)
N = 1 << 22
side = 2048
v, u = np.divmod(np.arange(N), side)
d = plane[None] + (u[:, None] / side) * dx[None] + (v[:, None] / side) * dy[None]
rays = pt.make_rays(np.tile(pos, (N, 1)), d.astype(np.float32))
d_rays = torch.from_numpy(rays.view(np.uint8).copy()).cuda()
d_hits = torch.zeros(N * 16, dtype=torch.uint8, device="cuda")
for n in (64, 1024, 16384, 65536, 262144, 1 << 20, 1 << 22):
    for _ in range(3):
        dev.intersect_device(d_rays.data_ptr(), n, d_hits.data_ptr())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        dev.intersect_device(d_rays.data_ptr(), n, d_hits.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("%-8s %8d rays  %8.1f us per launch  %7.2f Grays/s" % (name, n, ms * 1e3, n / ms / 1e6))
