#!/usr/bin/env python3
"""Inner nodes and triangle tests per extend ray by bounce group, from counting renders with different max_bounces
(same seed: a render with fewer bounces traces a prefix of every path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt
W = H = 1080
spp = 8
prev = None
for mb in (1, 2, 3, 5, 7, 10):
    scene = pt.LoadedScene("scenes/cornell_teapot.obj", max_bounces=mb)
    dev = scene.upload(flags=pt.FLAG_COUNT_WORK | pt.FLAG_NO_BOX_CLIP)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    dev.render_device(scene.pinhole(W, H), spp, acc.data_ptr(), seed=0, first_sample=0, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s = dev.stats()
    cur = (s["extend_rays"], s["extend_inner_nodes"], s["extend_tri_tests"], s["shadow_rays"], s["shadow_inner_nodes"], s["shadow_tri_tests"])
    if prev:
        d = [a - b for a, b in zip(cur, prev[1])]
        print("bounces %d..%d: extend rays %10d  nodes/ray %6.2f  tris/ray %5.2f   shadow rays %10d nodes/ray %6.2f tris/ray %5.2f" % (
            prev[0] + 1, mb, d[0], d[1] / max(d[0], 1), d[2] / max(d[0], 1), d[3], d[4] / max(d[3], 1), d[5] / max(d[3], 1)), flush=True)
    else:
        print("bounce 1: extend rays %10d  nodes/ray %6.2f  tris/ray %5.2f   shadow rays %10d nodes/ray %6.2f tris/ray %5.2f" % (
            cur[0], cur[1] / cur[0], cur[2] / cur[0], cur[3], cur[4] / max(cur[3], 1), cur[5] / max(cur[3], 1)), flush=True)
    prev = (mb, cur)
    dev.close()
