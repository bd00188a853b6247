#!/usr/bin/env python3
"""How long do render calls take when the queue set has to grow between calls?  (bench.py's config-3 figure)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pathtracer_amd as pt

scene = pt.LoadedScene(os.path.join(ROOT, "scenes", "cornell_teapot.obj"))
dev = scene.upload(flags=pt.FLAG_TIME_KERNELS)
ph = scene.pinhole(1080, 1080)
buf = torch.zeros(1080 * 1080 * 3, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for spp in (64, 640, 640, 1024, 1024, 512, 1024):
    dev.stats_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.render_device(ph, spp, buf.data_ptr(), seed=0, first_sample=0, stream=stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = dev.stats()
    k = st["extend_ms"] + st["shadow_ms"] + st["shade_ms"] + st["generate_ms"] + st["other_ms"]
    print("spp %4d: call %.3f s + sync %.3f s = %.3f s; kernels %.3f s; launches %d; held %.1f GB; %.0f Msamples/s" % (
        spp, t1 - t0, t2 - t1, t2 - t0, k / 1e3, st["extend_launches"], dev.memory() / 1e9, spp * 1080 * 1080 / (t2 - t0) / 1e6), flush=True)
