#!/usr/bin/env python3
"""CPU-only study of the traversal's step mix on real bounce rays (test/diagnostic tool; uses the oracle
only to produce rays).  Reports how many leaf visits are empty and how they are reached."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle as orc
import pathtracer_amd as pt

name = sys.argv[1] if len(sys.argv) > 1 else "cornell_teapot"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
obj = os.path.join(ROOT, "scenes", name + ".obj")
sc = pt.LoadedScene(obj)
a = sc.arrays()
nodes = np.asarray(a["nodes"]).view(np.uint32).reshape(-1, 2)
w0 = nodes[:, 0].copy(); w1 = nodes[:, 1].copy()
split = w0.view(np.float32)
ref = orc.Scene(obj)
rays, _ = ref.bounce_rays(64, 64, bounces=10, seed=3)
rng = np.random.default_rng(1)
rays = rays[rng.permutation(len(rays))[:N]]
kinds = collections.Counter()
n_leaf = int(((w1 & 3) == 3).sum()); n_empty = int((((w1 & 3) == 3) & ((w1 >> 2) == 0)).sum())
print("nodes", len(w0), "leaves", n_leaf, "empty leaves", n_empty)
f32 = np.float32
tot = collections.Counter()
for r in rays:
    o = (r["o"] if "o" in r.dtype.names else r[0]); d = (r["d"] if "d" in r.dtype.names else r[2])
    t1 = f32(r["tmin"]); t_end = f32(r["tmax"]); t2 = t_end
    node = 0; stack = []; how = "root"
    tot["rays"] += 1
    while True:
        k = int(w1[node] & 3)
        if k != 3:
            tot["inner"] += 1
            dist = split[node]; da = f32(d[k]); oa = f32(o[k])
            left, right = node + 1, int(w1[node] >> 2)
            if da == 0:
                node = left if oa <= dist else right; how = "single"
            else:
                with np.errstate(all="ignore"):
                    t = f32(f32(dist - oa) / da)
                near, far = (left, right) if da >= 0 else (right, left)
                if t > t2: node = near; how = "single"
                elif t < t1: node = far; how = "single"
                else:
                    stack.append((far, t)); node = near; t2 = t; how = "near_of_both"
        else:
            cnt = int(w1[node] >> 2)
            tot["leaf"] += 1
            if cnt == 0:
                tot["empty_" + how] += 1
            else:
                tot["full_" + how] += 1; tot["tris"] += cnt
            # no hit testing here: follow the no-hit path until the ray would really end is not possible
            # without triangles, so use the oracle's answer for the stop: approximate by stopping at the
            # first non-empty leaf with probability ~ hit; instead we simply continue (upper bound walk).
            if t2 == t_end or not stack: break
            far, t = stack.pop(); node = far; t1 = t; t2 = stack[-1][1] if stack else t_end; how = "popped"
for k in sorted(tot): print("%-22s %9d  %.3f/ray" % (k, tot[k], tot[k] / tot["rays"]))
