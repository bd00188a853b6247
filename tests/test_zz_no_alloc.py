"""The allocation-free entry point (-m gpu, collected last): ptx_intersect_batch_device_noalloc against
ptx_intersect_batch_device through the C ABI, bit for bit, on two calls in a row - the second with a smaller
input, so that stale cursor / spill contents of the shared library-static scratch array would show - plus the
size limit.  The entry point has no counterpart in the reference; it exists for callers that cannot call
hipMalloc (HIP graph capture, tight serving loops)."""
import os

import numpy as np
import pytest

import pathtracer_amd as pt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")


def random_rays(scene, n, seed):
    tris = scene.arrays()["triangles"].reshape(-1, 3)
    lo, hi = tris.min(0), tris.max(0)
    rng = np.random.default_rng(seed)
    o = (lo + rng.random((n, 3)) * (hi - lo)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays = pt.make_rays(o, d)
    seg = n // 4  # a quarter as segment rays with t in [0, 1] (shadow-ray shaped)
    rays["tmax"][:seg] = 1.0
    return rays


def test_noalloc_entry_matches_the_allocating_one_twice_in_a_row():
    import torch

    scene = pt.LoadedScene(os.path.join(SCENES, "sr2.obj"))  # the deepest tree shipped: stacks spill past the LDS levels
    dev = scene.upload()
    for n, seed in ((300_000, 1), (4_097, 2)):  # the second call is smaller
        rays = random_rays(scene, n, seed)
        d_rays = torch.from_numpy(rays.view(np.uint8).copy()).cuda()
        want = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
        got = torch.full((n * 16,), 0xAB, dtype=torch.uint8, device="cuda")
        dev.intersect_device(d_rays.data_ptr(), n, want.data_ptr())
        torch.cuda.synchronize()
        dev.intersect_device_noalloc(d_rays.data_ptr(), n, got.data_ptr())
        torch.cuda.synchronize()
        w = want.cpu().numpy().view(pt.HIT_DTYPE)
        g = got.cpu().numpy().view(pt.HIT_DTYPE)
        assert g.tobytes() == w.tobytes()
        assert (w["index"] != pt.MISS).mean() > 0.05
    # the defining property: the call allocates nothing - device memory free / held by the handle do not move over
    # repeated calls, also on a stream of the caller's own (what a capture would run on)
    side = torch.cuda.Stream()
    dev.intersect_device_noalloc(d_rays.data_ptr(), n, got.data_ptr(), stream=side.cuda_stream)  # (the HIP runtime sets a
    side.synchronize()                                                 # stream's own resources up at its first launch: 2 MiB)
    torch.cuda.synchronize()
    free_before, held_before = torch.cuda.mem_get_info()[0], dev.memory()
    for k in range(3):
        got.fill_(0xCD)
        torch.cuda.synchronize()  # (the fill runs on torch's stream, the call below on `side`)
        dev.intersect_device_noalloc(d_rays.data_ptr(), n, got.data_ptr(), stream=side.cuda_stream if k else None)
        side.synchronize()
        torch.cuda.synchronize()
        assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes()
    assert torch.cuda.mem_get_info()[0] == free_before and dev.memory() == held_before
    # a second handle on the same GPU shares the array (its address is per device, not per scene)
    other = scene.upload()
    other.intersect_device_noalloc(d_rays.data_ptr(), n, got.data_ptr())
    torch.cuda.synchronize()
    assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes()
    other.close()
    with pytest.raises(pt.PtxError) as e:  # beyond the stated maximum: refused, nothing launched
        dev.intersect_device_noalloc(d_rays.data_ptr(), pt.NOALLOC_MAX_RAYS + 1, got.data_ptr())
    assert e.value.code == -5 and "PTX_NOALLOC_MAX_RAYS" in str(e.value)
    dev.intersect_device_noalloc(d_rays.data_ptr(), 0, got.data_ptr())  # empty batch: nothing to do
    dev.close()
