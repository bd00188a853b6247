"""Ray log (tracing/src/raylogger.rs): the GPU renderer's 33-byte records against the file one
reference thread would write for the same iterations (oracle restatement, COUNTER sampler), byte for byte.
Parity unpinned beyond the restatement: the reference holds no raylog fixture (its logger is a cargo feature)."""
import os

import numpy as np
import pytest

import pathtracer_amd as pt
import pyoracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scene_path(name):
    return os.path.join(ROOT, "scenes", name)


@pytest.mark.parametrize("name,w,h,first,count", [("cornell.obj", 40, 32, 3, 2), ("sr2.obj", 24, 24, 0, 1)])
def test_raylog_equals_reference_file(name, w, h, first, count):
    ref = orc.Scene(scene_path(name), max_bounces=6)
    want = ref.render_raylog(w, h, seed=9, first_sample=first, count=count)
    scene = pt.LoadedScene(scene_path(name), max_bounces=6)
    dev = scene.upload()
    dev.raylog_enable(len(want) // 33 + 10)
    dev.set_batch_samples(1)  # two batches: the file order does not depend on the batching
    img = dev.render(scene.pinhole(w, h), count, seed=9, first_sample=first)
    got, dropped = dev.raylog_read()
    assert dropped == 0
    assert got.tobytes() == want.tobytes()
    recs = got
    assert recs["shadow"].any() and (recs["shadow"] == 0).any() and recs["intersect"].any()
    assert set(np.unique(recs["iteration"])) == set(range(first, first + count))
    # the log was emptied by the read; logging does not change the image
    again, _ = dev.raylog_read()
    assert len(again) == 0
    dev.raylog_enable(0)
    plain = dev.render(scene.pinhole(w, h), count, seed=9, first_sample=first)
    assert plain.tobytes() == img.tobytes()
    with pytest.raises(pt.PtxError):
        dev.raylog_read()
    dev.close()


def test_raylog_overflow_is_reported():
    scene = pt.LoadedScene(scene_path("cornell.obj"), max_bounces=4)
    dev = scene.upload()
    dev.raylog_enable(1000)
    dev.render(scene.pinhole(32, 32), 1, seed=1)
    got, dropped = dev.raylog_read()
    assert len(got) == 1000 and dropped > 0
    dev.close()


def test_pathtracer_cli_ray_log(tmp_path):
    import subprocess

    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "pathtracer-cli")
    log = str(tmp_path / "raylog0.bin")
    r = subprocess.run([exe, "-i", scene_path("cornell.obj"), "-o", str(tmp_path / "o.png"), "-s", "32x24", "-t", "1", "-n", "3", "-m", "5",
                        "--seed", "4", "--ray-log", log], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    want = orc.Scene(scene_path("cornell.obj"), max_bounces=5).render_raylog(32, 24, seed=4, first_sample=0, count=3)
    assert open(log, "rb").read() == want.tobytes()
    # and the image is the one a run without the log writes
    r2 = subprocess.run([exe, "-i", scene_path("cornell.obj"), "-o", str(tmp_path / "p.png"), "-s", "32x24", "-t", "1", "-n", "3", "-m", "5",
                         "--seed", "4"], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0, r2.stderr
    assert open(tmp_path / "o.png", "rb").read() == open(tmp_path / "p.png", "rb").read()
