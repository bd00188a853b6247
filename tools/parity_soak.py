#!/usr/bin/env python3
"""A longer run of the config-2 parity check than the test suite's: for several seeds and every shipped scene, the kdtree-tester-cli ray set
(camera rays bounced through the scene) plus synthetic rays, GPU hit records against the oracle's pointer-tree traversal, bit for bit; and one
small frame per scene against the oracle's render.  usage: python tools/parity_soak.py [seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT), sys.path.insert(0, os.path.join(ROOT, "oracle")), sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pathtracer_amd as pt
import pyoracle as orc
from test_gpu_parity import synthetic_rays

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
total = 0
for name in ("cornell", "cornell_teapot", "cornell_textured", "sr2", "cubesplosion"):
    path = os.path.join(ROOT, "scenes", name + ".obj")
    ref = orc.Scene(path)
    tree = ref.tree()
    tris = ref.export()["triangles"].reshape(-1, 3, 3)
    scene = pt.LoadedScene(path)
    dev = scene.upload()
    t0 = time.time()
    for seed in range(1, seeds + 1):
        res = 64 + 16 * seed
        bounced, _ = ref.bounce_rays(res, res + 8, 10, seed=100 + seed, threads=8, check=False)
        rays = np.concatenate([bounced, synthetic_rays(tris.min((0, 1)), tris.max((0, 1)), 1 << 16, 0xBEEF0000 + seed)])
        want = tree.intersect(rays, threads=8)
        got = dev.intersect(rays)
        assert got.tobytes() == want.tobytes(), (name, seed)
        total += len(rays)
    w, h = 72, 56
    img = dev.render(scene.pinhole(w, h), 6, seed=77)
    want_img, _ = ref.render(w, h, threads=1, iterations_per_thread=6, mode=1, seed=77)
    assert img.tobytes() == want_img.tobytes(), name
    dev.close()
    print("%-18s %d seeds ok, frame ok (%.1f s)" % (name, seeds, time.time() - t0), flush=True)
print("parity soak: %d rays bit-exact" % total)
