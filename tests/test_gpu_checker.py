"""The GPU side of `kdtree-tester-cli test` (-m gpu): PTX_FLAG_CHECK_BRUTE re-intersects every ray of a render
against all triangles and reports disagreements with the KD-tree traversal (ray_bouncer.rs:35-56,
checked_intersection.rs:10-56).  A sound tree gives none; a tree with a leaf emptied on purpose gives exactly
the camera-ray fails the oracle finds for the same tree."""
import os
import subprocess

import numpy as np
import pytest

import pathtracer_amd as pt
import pyoracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
f32 = np.float32


@pytest.mark.parametrize("name", ["cornell_teapot", "sr2"])
def test_sound_tree_has_no_fails(name):
    scene = pt.LoadedScene(os.path.join(SCENES, name + ".obj"), max_bounces=6)
    dev = scene.upload(flags=pt.FLAG_CHECK_BRUTE)
    dev.render(scene.pinhole(96, 96), 2, seed=3)
    checked, fails = dev.check_report()
    st = dev.stats()
    assert checked == st["extend_rays"] + st["shadow_rays"] and checked > 96 * 96 * 2
    assert fails == 0 and len(dev.check_fail_log()) == 0
    dev.close()


def test_emptied_leaves_are_found_like_the_oracle_finds_them():
    path = os.path.join(SCENES, "cornell.obj")
    scene = pt.LoadedScene(path, max_bounces=1)
    a = scene.arrays()
    nodes = a["nodes"].copy()
    is_leaf = (nodes["w1"] & 3) == 3
    full = np.flatnonzero(is_leaf & ((nodes["w1"] >> 2) > 0))
    nodes["w1"][full[::2]] = 3  # every other non-empty leaf loses its triangles
    dev = pt.DeviceScene.from_arrays(a["triangles"], nodes, a["leaf_indices"], a["normals"], a["texcoords"], a["material_ids"],
                                     a["materials"], a["material_textures"], lights=a["lights"], max_bounces=1, flags=pt.FLAG_CHECK_BRUTE)
    w = h = 64
    dev.render(scene.pinhole(w, h), 1, seed=9)
    checked, fails = dev.check_report()
    log = dev.check_fail_log()
    assert fails == len(log) > 0
    # the oracle on the same camera rays (bounce_rays uses the same counter sampler: seed, pixel, sample 0)
    ref = orc.Scene(path, max_bounces=1)
    rays, _ = ref.bounce_rays(w, h, bounces=1, seed=9)
    cam = rays[(rays["tmax"] > 2.0)]
    assert len(cam) == w * h
    broken = orc.Tree.from_flat(a["triangles"], nodes, a["leaf_indices"], raw_triangles=True)
    kd, brute = broken.intersect(cam, threads=4), ref.tree().brute(cam, threads=4)
    both = (kd["index"] != orc.MISS) & (brute["index"] != orc.MISS)
    ok = ((kd["index"] == orc.MISS) & (brute["index"] == orc.MISS)) | (both & (kd["index"] == brute["index"]) & (np.abs(kd["t"] - brute["t"]) < 1e-6)
                                                                       & (np.abs(kd["u"] - brute["u"]) < 1e-5) & (np.abs(kd["v"] - brute["v"]) < 1e-5))
    want = cam[~ok]
    want_correct = (want["o"] + brute["t"][~ok, None] * want["d"]).astype(f32)
    cam_pos = cam["o"][0]
    got = log[(log["o"] == cam_pos).all(1)]  # camera-ray fails (shadow rays start on surfaces)
    assert len(got) == len(want) > 0
    key = lambda p: p[np.lexsort(p.T)]
    assert key(got["correct"]).tobytes() == key(want_correct).tobytes()
    assert (log["i"] == np.arange(len(log), dtype=np.uint16)).all()
    dev.close()


def test_kdtree_tester_cli(tmp_path):
    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "kdtree-tester-cli")
    out = tmp_path / "fails.bin"
    r = subprocess.run([exe, "test", "-i", os.path.join(SCENES, "cornell_teapot.obj"), "-s", "128x128", "-b", "5", "-o", str(out)],
                       capture_output=True, text=True, check=True)
    assert "Building kdtree..." in r.stdout and "Testing up to %d rays..." % (128 * 128 * 5) in r.stdout
    assert "Found 0 fails" in r.stdout and out.exists() and out.stat().st_size == 0
