#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point ptx_intersect_batch (rays in, hits out over PCIe) beside the
device-resident rate of ptx_intersect_batch_device on the same rays."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pathtracer_amd as pt

scene = pt.LoadedScene(os.path.join(ROOT, "scenes", "cornell_teapot.obj"))
dev = scene.upload()
tris = scene.arrays()["triangles"].reshape(-1, 3)
lo, hi = tris.min(0), tris.max(0)
rng = np.random.default_rng(1)
n = 1 << 24
o = (lo + rng.random((n, 3), dtype=np.float32) * (hi - lo)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
rays = pt.make_rays(o, d)
dev.intersect(rays[:1 << 20])
t0 = time.perf_counter(); hits = dev.intersect(rays); t1 = time.perf_counter()
d_rays = torch.from_numpy(rays.view(np.uint8)).cuda(); d_hits = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
dev.intersect_device(d_rays.data_ptr(), n, d_hits.data_ptr()); torch.cuda.synchronize()
t2 = time.perf_counter(); dev.intersect_device(d_rays.data_ptr(), n, d_hits.data_ptr()); torch.cuda.synchronize(); t3 = time.perf_counter()
assert d_hits.cpu().numpy().tobytes() == hits.tobytes()
print("%d random rays: host buffers (PCIe in + out, pageable) %.3f s = %.2f Grays/s; device-resident %.4f s = %.2f Grays/s" % (n, t1 - t0, n / (t1 - t0) / 1e9, t3 - t2, n / (t3 - t2) / 1e9))
