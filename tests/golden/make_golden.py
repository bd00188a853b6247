#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (oracle/), which is itself pinned by the
reference's known-answer tests (tests/test_oracle_kat.py).  The reference is Rust and cannot be built in
this image, so these vectors come from the restatement, not from the reference binary.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Per scene: 2048 bounced rays (kdtree-tester-cli methodology) + 2048 synthetic rays with their
(index,t,u,v) from the POINTER-tree traversal, the flat-tree digest and statistics, and a 24x24, 2-spp
radiance image in counter-sampler mode (seed 2024).  raylog.npz: the 33-byte ray records (raylogger.rs:54-66) of
a 16x12 render, iterations 1 and 2, max_bounces 6, seed 2024, for cornell (one light) and sr2 (four)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as orc  # noqa: E402

SCENES = ["cornell", "cornell_textured", "cube", "cornell_teapot", "cubesplosion", "sr2"]


def synthetic(lo, hi, n, seed):
    rng = np.random.default_rng(seed)
    c, h = (lo + hi) / 2, (hi - lo) / 2 * 1.1
    o = (c + rng.uniform(-1, 1, (n, 3)) * h).astype(np.float32)
    z = rng.uniform(-1, 1, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    r = np.sqrt(1 - z * z)
    d = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)
    rays = orc.make_rays(o, d)
    rays["d"][::16, 0] = 0.0
    rays["d"][5::16, 1] = -0.0
    rays["tmax"][3::4] = 1.0
    return rays


def main():
    for name in SCENES:
        path = os.path.join(ROOT, "scenes", name + ".obj")
        sc = orc.Scene(path)
        tree = sc.tree()
        bounced, fails = sc.bounce_rays(24, 24, 10, seed=7, threads=4)
        assert fails == 0
        rng = np.random.default_rng(1)
        if len(bounced) > 2048:
            bounced = bounced[np.sort(rng.choice(len(bounced), 2048, replace=False))]
        tris = sc.export()["triangles"].reshape(-1, 3, 3)
        rays = np.concatenate([bounced, synthetic(tris.min((0, 1)), tris.max((0, 1)), 2048, 0x5EED0001)])
        hits = tree.intersect(rays)
        nodes, leaf = tree.to_flat()
        digest = hashlib.sha256(nodes.tobytes() + leaf.tobytes()).hexdigest()
        img, _ = sc.render(24, 24, threads=1, iterations_per_thread=2, mode=1, seed=2024)
        st = tree.stats()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rays=rays, hits=hits, image=img,
                            tree_sha256=np.array(digest), stats=np.array([st[k] for k in sorted(st)], np.int64),
                            stats_keys=np.array(sorted(st)))
        print(name, len(rays), "rays", st, digest[:12])


def raylogs():
    out = {}
    for name in ("cornell", "sr2"):
        sc = orc.Scene(os.path.join(ROOT, "scenes", name + ".obj"), max_bounces=6)
        out[name] = sc.render_raylog(16, 12, seed=2024, first_sample=1, count=2)
        print(name, len(out[name]) // 33, "ray records")
    np.savez_compressed(os.path.join(HERE, "raylog.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "raylog":
        raylogs()
        sys.exit(0)
    main()
