#!/usr/bin/env python3
"""Per-bounce ray counts and kernel times of one batch (run under rocprofv3 --kernel-trace, or alone for counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pathtracer_amd as pt
scene = pt.LoadedScene("scenes/%s.obj" % (sys.argv[1] if len(sys.argv) > 1 else "cornell_teapot"))
dev = scene.upload(flags=pt.FLAG_TIME_KERNELS)
ph = scene.pinhole(1080, 1080)
acc = torch.zeros(1080 * 1080 * 3, device="cuda")
for i in range(3):
    dev.render_device(ph, 4, acc.data_ptr(), first_sample=4 * i)
torch.cuda.synchronize()
e, s = dev.last_batch_counts()
print("extend counts", e.tolist())
print("shadow counts", s.tolist())
print(dev.stats())
