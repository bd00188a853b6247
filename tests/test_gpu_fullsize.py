"""GPU parity at BASELINE.json's full sizes (-m gpu): cornell_teapot, 1080 x 1080.

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


def test_extend_two_million_rays_bit_exact():
    """configs[1] at scale: bounced + synthetic rays, every hit record equal to the pointer-tree traversal."""
    ref = orc.Scene(TEAPOT)
    bounced, fails = ref.bounce_rays(360, 360, 10, seed=11, threads=16)
    assert fails == 0
    tris = ref.export()["triangles"].reshape(-1, 3, 3)
    rays = np.concatenate([bounced, synthetic_rays(tris.min((0, 1)), tris.max((0, 1)), 1 << 20, 0x5EED0002)])
    assert len(rays) > 2_000_000
    want = ref.tree().intersect(rays, threads=16)
    dev = pt.LoadedScene(TEAPOT).upload()
    got = dev.intersect(rays)
    assert_hits_equal(got, want)
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
