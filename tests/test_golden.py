"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).

CPU part: the oracle and the product's host code still reproduce them (guards against silent drift of
either).  GPU part (-m gpu): the HIP path reproduces them bit for bit."""
import hashlib
import os

import numpy as np
import pytest

import pathtracer_amd as pt
import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = ["cornell", "cornell_textured", "cube", "cornell_teapot", "cubesplosion", "sr2"]


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_golden(name):
    g = load(name)
    sc = orc.Scene(os.path.join(ROOT, "scenes", name + ".obj"))
    tree = sc.tree()
    assert tree.intersect(g["rays"]).tobytes() == g["hits"].tobytes()
    nodes, leaf = tree.to_flat()
    assert hashlib.sha256(nodes.tobytes() + leaf.tobytes()).hexdigest() == str(g["tree_sha256"])
    img, _ = sc.render(24, 24, threads=1, iterations_per_thread=2, mode=1, seed=2024)
    assert img.tobytes() == g["image"].tobytes()
    # brute force agrees with the tree within the tester's tolerances (checked_intersection.rs:10-25)
    brute = tree.brute(g["rays"][:512])
    h = g["hits"][:512]
    assert (brute["index"] == h["index"]).all()
    assert np.abs(brute["t"] - h["t"]).max() < 1e-6


@pytest.mark.parametrize("name", SCENES)
def test_product_host_tree_matches_golden_digest(name):
    g = load(name)
    a = pt.LoadedScene(os.path.join(ROOT, "scenes", name + ".obj")).arrays()
    assert hashlib.sha256(a["nodes"].tobytes() + a["leaf_indices"].tobytes()).hexdigest() == str(g["tree_sha256"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_gpu_reproduces_golden(name):
    g = load(name)
    scene = pt.LoadedScene(os.path.join(ROOT, "scenes", name + ".obj"))
    dev = scene.upload()
    assert dev.intersect(g["rays"]).tobytes() == g["hits"].tobytes()
    img = dev.render(scene.pinhole(24, 24), 2, seed=2024)
    assert img.tobytes() == g["image"].tobytes()
    dev.close()


RAYLOG = dict(w=16, h=12, seed=2024, first_sample=1, count=2, max_bounces=6)


@pytest.mark.parametrize("name", ["cornell", "sr2"])
def test_oracle_reproduces_golden_raylog(name):
    want = np.load(os.path.join(GOLD, "raylog.npz"))[name]
    sc = orc.Scene(os.path.join(ROOT, "scenes", name + ".obj"), max_bounces=RAYLOG["max_bounces"])
    got = sc.render_raylog(RAYLOG["w"], RAYLOG["h"], seed=RAYLOG["seed"], first_sample=RAYLOG["first_sample"], count=RAYLOG["count"])
    assert got.tobytes() == want.tobytes()
    recs = want.view(pt.RAYLOG_DTYPE)
    assert len(recs) > 400 and set(np.unique(recs["iteration"])) == {1, 2} and recs["px"].max() == 15 and recs["py"].max() == 11


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "sr2"])
def test_gpu_reproduces_golden_raylog(name):
    want = np.load(os.path.join(GOLD, "raylog.npz"))[name]
    scene = pt.LoadedScene(os.path.join(ROOT, "scenes", name + ".obj"), max_bounces=RAYLOG["max_bounces"])
    dev = scene.upload()
    dev.raylog_enable(len(want) // 33 + 1)
    dev.render(scene.pinhole(RAYLOG["w"], RAYLOG["h"]), RAYLOG["count"], seed=RAYLOG["seed"], first_sample=RAYLOG["first_sample"])
    got, dropped = dev.raylog_read()
    assert dropped == 0 and got.tobytes() == want.tobytes()
    dev.close()
