#!/usr/bin/env python3
"""Times the host stages either side of the render: load+build, build alone with 1 / all
threads, tone map, PNG."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pathtracer_amd as pt
if len(sys.argv) > 1 and sys.argv[1] == "build":
    tr = pt.LoadedScene(os.path.join(ROOT, "scenes", sys.argv[2] + ".obj")).arrays()["triangles"].copy()
    t = time.time(); kd = pt.build_kdtree(tr); print("%.3f" % (time.time() - t)); sys.exit(0)
for n in ["cornell_teapot", "sr2", "cubesplosion"]:
    t = time.time(); sc = pt.LoadedScene(os.path.join(ROOT, "scenes", n + ".obj")); t1 = time.time() - t
    one = subprocess.run([sys.executable, __file__, "build", n], env=dict(os.environ, PTX_BUILD_THREADS="1"), capture_output=True, text=True).stdout.strip()
    par = subprocess.run([sys.executable, __file__, "build", n], capture_output=True, text=True).stdout.strip()
    print("%-16s load+build %.3f s   build: 1 thread %s s, parallel %s s" % (n, t1, one, par))
y, x = np.mgrid[0:1080, 0:1080]
img = ((np.sin(x / 90.) + np.cos(y / 70.)) * 60 + 128 + np.random.default_rng(0).normal(0, 6, (1080, 1080)))[..., None].repeat(3, 2).clip(0, 255)
acc = (img / 255.0).astype(np.float32) ** 2.2 * 64
for i in range(2):
    t = time.time(); rgb = pt.to_rgb8(acc, 64); t3 = time.time() - t
    t = time.time(); pt.write_png("/tmp/probe.png", rgb); t4 = time.time() - t
print("tone map %.3f s, png %.3f s (%d bytes)" % (t3, t4, os.path.getsize("/tmp/probe.png")))
