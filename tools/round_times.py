#!/usr/bin/env python3
"""Per-round ray counts and traversal kernel times (HIP events, ptx_round_times) of one batch after a warm-up batch.
usage: python tools/round_times.py [scene] [WxH] [spp]     (PTX_LIB selects a library variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt

name = sys.argv[1] if len(sys.argv) > 1 else "cornell_teapot"
W, H = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1080x1080").split("x"))
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 128
scene = pt.LoadedScene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", name + ".obj"))
dev = scene.upload(flags=pt.FLAG_TIME_KERNELS)
dev.set_batch_samples(spp)
ph = scene.pinhole(W, H)
acc = torch.zeros(H * W * 3, device="cuda")
dev.render_device(ph, spp, acc.data_ptr(), first_sample=0)
torch.cuda.synchronize()
dev.stats_reset()
dev.render_device(ph, spp, acc.data_ptr(), first_sample=spp)
torch.cuda.synchronize()
e, s = dev.last_batch_counts()
L = max(scene.desc.n_lights, 1)
n = int(dev.stats()["extend_launches"])
te, ts = dev.round_times(n)
print("%s %dx%d %d spp: %d rounds" % (name, W, H, spp, n))
for r in range(min(n, len(e))):
    print("round %2d  extend %10d rays %8.3f ms %7.2f G/s   shadow %10d rays %8.3f ms %7.2f G/s" % (
        r + 1, e[r], te[r], e[r] / te[r] / 1e6 if te[r] else 0, s[r] * L, ts[r], s[r] * L / ts[r] / 1e6 if ts[r] else 0))
print("total extend %.3f ms shadow %.3f ms" % (sum(te), sum(ts)))
