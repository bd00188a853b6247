"""GPU parity at BASELINE.json's full sizes (-m gpu): cornell_teapot and cornell_textured at 1080 x 1080
(configs[2], configs[3]), sr2 and cubesplosion at 3840 x 2160 (configs[4]).

The oracle still finishes in seconds here when it is asked for one sample per pixel or a few million rays, so
the full-size checks are direct comparisons; on top come the size-independent properties the renderer
offers (sample ranges are additive, the number of batches / pipelines in flight never changes a bit)."""
import os

import numpy as np
import pytest

import pathtracer_amd as pt
import pyoracle as orc

from test_gpu_parity import assert_hits_equal, synthetic_rays

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEAPOT = os.path.join(ROOT, "scenes", "cornell_teapot.obj")
W = H = 1080


def test_config2_full_size_hit_records_bit_exact():
    """configs[1] at the size SURVEY 8d states: ray set A = the kdtree-tester-cli bounced set at the reference's
    defaults, 512 x 512 pixels / 10 bounces (ray_bouncer.rs:57-120; 2.69 M extend + shadow rays), ray set B = 2^24
    uniform + 2^22 segment + 2^21 axis-parallel (+-0) rays, seed 0x5EED0001: 25.8 M rays, every hit record
    (index, t, u, v) of the GPU equal to the oracle's POINTER-tree traversal bit for bit.

    Set A is regenerated here without its brute-force side (35 G triangle tests: minutes) and must hash to the set the
    CHECKED generator produced (tests/golden/config2_fullsize.json, written by tests/golden/make_config2_golden.py,
    which also recorded 0 disagreements with brute force); the oracle's hit records are pinned by the same file."""
    import hashlib
    import json

    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config2_fullsize.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    ref = orc.Scene(TEAPOT)
    bounced, _ = ref.bounce_rays(512, 512, gold["bounces"], seed=gold["seed_bounced"], threads=16, check=False)
    assert len(bounced) == gold["bounced_rays"] and sha(bounced) == gold["bounced_rays_sha256"]
    tris = ref.export()["triangles"].reshape(-1, 3, 3)
    synth = synthetic_rays(tris.min((0, 1)), tris.max((0, 1)), gold["synthetic_n"], gold["seed_synthetic"])
    assert len(synth) == gold["synthetic_rays"] == (1 << 24) + (1 << 22) + (1 << 21)
    tree = ref.tree()
    dev = pt.LoadedScene(TEAPOT).upload()
    for rays, key in ((bounced, "bounced"), (synth, "synthetic")):
        want = tree.intersect(rays, threads=16)
        got = dev.intersect(rays)
        assert_hits_equal(got, want)
        if sha(rays) == gold[key + "_rays_sha256"]:  # (the synthetic set goes through numpy's sin / cos: same image, same bits)
            assert sha(got) == gold[key + "_hits_sha256"]
        else:
            assert key == "synthetic"
        assert 0.3 < (want["index"] != 0xFFFFFFFF).mean() < 0.99
    dev.close()


def test_full_frame_one_sample_bit_exact_and_additive():
    import torch

    ref = orc.Scene(TEAPOT, max_bounces=10)
    want, _ = ref.render(W, H, threads=1, iterations_per_thread=1, mode=1, seed=5, first_sample=2)
    scene = pt.LoadedScene(TEAPOT, max_bounces=10)
    dev = scene.upload()
    ph = scene.pinhole(W, H)
    stream = torch.cuda.current_stream().cuda_stream

    def render(first, count):
        buf = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
        dev.render_device(ph, count, buf.data_ptr(), seed=5, first_sample=first, stream=stream)
        torch.cuda.synchronize()
        return buf

    one = render(2, 1)
    assert one.cpu().numpy().tobytes() == want.tobytes()  # 1.17 M paths, 10 bounces: same bits as the CPU
    # samples [0, 6) in one call == [0, 2) + [2, 3) + [3, 6) accumulated into one buffer, whatever the batching
    whole = render(0, 6)
    parts = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    dev.set_batch_samples(2)
    dev.set_pipelines(2)
    for first, count in ((0, 2), (2, 1), (3, 3)):
        dev.render_device(ph, count, parts.data_ptr(), seed=5, first_sample=first, stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(whole, parts)
    s = dev.stats()
    assert s["samples"] == (1 + 6 + 6) * W * H
    img = whole.cpu().numpy()
    assert np.isfinite(img).all() and img.min() >= 0.0 and 0.05 < img.mean() / 6 < 5.0
    dev.close()


def test_large_synthetic_tree_bit_exact():
    """A tree two orders of magnitude larger than the shipped scenes (400 k triangles, 1.8 M nodes, leaves 25 deep:
    the traversal stack spills past its LDS levels): product builder -> device layout -> 1 M rays, every hit record
    equal to the oracle walking the same tree as pointers; occlusion answers equal `is_some()`."""
    rng = np.random.default_rng(1)
    n = 400_000
    centre = rng.random((n, 1, 3), dtype=np.float32) * 10 - 5
    tris = (centre + rng.normal(0, 0.03, (n, 3, 3)).astype(np.float32)).astype(np.float32).reshape(n, 9)
    kd = pt.build_kdtree(tris)
    assert len(kd.nodes) > 1_000_000 and kd.stats()["leaf_depth_max"] > 20
    rays = synthetic_rays(np.full(3, -5, np.float32), np.full(3, 5, np.float32), 1 << 20, 0x5EED0003)
    want = orc.Tree.from_flat(tris, kd.nodes, kd.leaf_indices).intersect(rays, threads=16)
    dev = pt.DeviceScene.from_arrays(tris, kd.nodes, kd.leaf_indices)
    got = dev.intersect(rays)
    assert_hits_equal(got, want)
    assert (want["index"] != 0xFFFFFFFF).mean() > 0.2
    import torch

    d_rays = torch.from_numpy(rays.view(np.uint8).copy()).cuda()
    d_out = torch.zeros(len(rays), dtype=torch.uint8, device="cuda")
    dev.occluded_device(d_rays.data_ptr(), len(rays), d_out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().astype(bool), want["index"] != 0xFFFFFFFF)
    dev.close()


def _render_dev(dev, ph, first, count, seed, stream):
    import torch

    buf = torch.zeros(ph.height * ph.width * 3, dtype=torch.float32, device="cuda")
    dev.render_device(ph, count, buf.data_ptr(), seed=seed, first_sample=first, stream=stream)
    torch.cuda.synchronize()
    return buf


def test_config4_cornell_textured_full_frame_bit_exact():
    """configs[3] (texture lookups, refraction with total internal reflection, the metal box) at its full
    1080 x 1080: 4 samples of every pixel against the oracle (4 threads x 1 iteration = the same left fold of
    one-sample buffers), then the batching / sharding invariants at that size."""
    import torch

    path = os.path.join(ROOT, "scenes", "cornell_textured.obj")
    ref = orc.Scene(path, max_bounces=10)
    want, _ = ref.render(W, H, threads=4, iterations_per_thread=1, mode=1, seed=17, first_sample=3)
    scene = pt.LoadedScene(path, max_bounces=10)
    dev = scene.upload()
    ph = scene.pinhole(W, H)
    stream = torch.cuda.current_stream().cuda_stream
    got = _render_dev(dev, ph, 3, 4, 17, stream)
    assert got.cpu().numpy().tobytes() == want.tobytes()
    # 2048 spp sharded 1024 / 512 / 256 per GPU (configs[3]) is this at scale: the shards of [0, 24) over 1, 2, 4
    # and 8 "GPUs", each rendered on its own and summed in shard order, equal ptx_render_multi's peer fold
    whole = _render_dev(dev, ph, 0, 24, 17, stream)
    for n in (2, 8):
        acc = None
        for g in range(n):
            first, count = pt.shard_samples(0, 24, n, g)
            part = _render_dev(dev, ph, first, count, 17, stream)
            acc = part if acc is None else acc + part
        assert torch.allclose(acc, whole, rtol=2e-6, atol=1e-6)  # same samples, another order of the float sums
    img = whole.cpu().numpy().reshape(H, W, 3) / 24
    assert np.isfinite(img).all() and img.min() >= 0.0
    dev.close()


@pytest.mark.parametrize("name", ["sr2", "cubesplosion"])
def test_config5_4k_frames_bit_exact_and_invariants(name):
    """configs[4]: 3840 x 2160 (non-square camera, camera.rs:72; 8.3 M pixels; 4 resp. 2 lights).  16 samples of
    every pixel bit-exact against the oracle (16 threads x 1 iteration), then at that size: a call split into
    ranges, batches and pipelines, and path regeneration over a queue far smaller than the batch, change no bit."""
    import torch

    w, h = 3840, 2160
    path = os.path.join(ROOT, "scenes", name + ".obj")
    ref = orc.Scene(path, max_bounces=10)
    want, _ = ref.render(w, h, threads=16, iterations_per_thread=1, mode=1, seed=23, first_sample=5)
    scene = pt.LoadedScene(path, max_bounces=10)
    dev = scene.upload()
    ph = scene.pinhole(w, h)
    stream = torch.cuda.current_stream().cuda_stream
    got = _render_dev(dev, ph, 5, 16, 23, stream)
    g = got.cpu().numpy()
    assert g.tobytes() == want.tobytes()
    assert (g.reshape(h, w, 3).sum(2) > 0).mean() > 0.05  # the picture is not empty
    del want
    # ranges + batches + pipelines
    parts = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    dev.set_batch_samples(3)
    dev.set_pipelines(2)
    for first, count in ((5, 7), (12, 1), (13, 8)):
        dev.render_device(ph, count, parts.data_ptr(), seed=23, first_sample=first, stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(parts, got)
    # path regeneration: 16 x 8.3 M = 133 M paths through 6 Mi queue slots
    dev.set_pipelines(1)
    dev.set_batch_samples(0)
    dev.set_queue_slots(6 << 20)
    regen = _render_dev(dev, ph, 5, 16, 23, stream)
    assert torch.equal(regen, got)
    ext, _ = dev.last_batch_counts()
    assert len(ext) > 10  # more rounds than bounces: the segments were topped up with fresh camera rays
    # a wide call with the default budget: 96 samples = 0.8 G paths, more than one batch may hold (2^31 - 2^20 ids)
    dev.set_queue_slots(0)
    a = _render_dev(dev, ph, 0, 96, 23, stream)
    dev.set_batch_samples(32)
    b = _render_dev(dev, ph, 0, 96, 23, stream)
    assert torch.equal(a, b)
    st = dev.stats()
    assert st["samples"] == (16 + 16 + 16 + 96 + 96) * w * h
    dev.close()
