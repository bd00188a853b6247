"""Multi-GPU behind the C ABI (-m gpu): ptx_render_multi = `render_parallel_iterations` with GPUs for threads
(tracing/src/worker.rs:84-119: private buffers, then `buffers.reduce(Add::add)`).

A one-GPU box can run every part of it: one scene (a one-rank RCCL communicator included), and several
scenes that share the GPU, whose buffers are folded with the peer path (copy + add kernel, shard order).
Also here: two handles rendering on one GPU with the default memory budget, the trim call, and overlapping
*_batch_device calls on two streams (the library serialises them)."""
import os
import subprocess

import numpy as np
import pytest

import pathtracer_amd as pt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")


def test_one_scene_list_gives_the_bits_of_ptx_render():
    scene = pt.LoadedScene(os.path.join(SCENES, "cornell.obj"), max_bounces=6)
    dev = scene.upload()
    ph = scene.pinhole(64, 48)
    start = np.random.default_rng(1).random((48, 64, 3), dtype=np.float32)  # "added into, never cleared"
    want = dev.render(ph, 5, seed=9, first_sample=2, accum=start.copy())
    for how in (pt.REDUCE_AUTO, pt.REDUCE_PEER, pt.REDUCE_RCCL):  # RCCL: ncclCommInitAll + ncclReduce with one rank
        got = pt.render_multi([dev], ph, 5, seed=9, first_sample=2, accum=start.copy(), reduce=how)
        assert got.tobytes() == want.tobytes(), how
    dev.close()


@pytest.mark.parametrize("n_shards,total", [(2, 7), (3, 8), (4, 2)])
def test_shards_on_one_gpu_fold_in_shard_order(n_shards, total):
    """N handles on the same GPU, peer reduce: the image is shard 0 rendered on top of the caller's buffer plus
    the other shards' own sums, added in shard order - bit for bit - and equals the one-GPU image up to the
    order of the float sums.  (4, 2): two of the four shards get no sample at all."""
    scene = pt.LoadedScene(os.path.join(SCENES, "cornell_textured.obj"), max_bounces=5)
    devs = [scene.upload() for _ in range(n_shards)]
    ph = scene.pinhole(56, 40)
    start = np.full((40, 56, 3), 0.25, np.float32)
    calls = []
    got = pt.render_multi(devs, ph, total, seed=4, first_sample=3, accum=start.copy(), reduce=pt.REDUCE_PEER,
                          progress=lambda done, secs: calls.append(done))
    expect = None
    for g in range(n_shards):
        first, count = pt.shard_samples(3, total, n_shards, g)
        part = devs[0].render(ph, count, seed=4, first_sample=first, accum=start.copy() if g == 0 else None) if count or g == 0 else None
        if g == 0:
            expect = part
        elif part is not None:
            expect = expect + part
    assert got.tobytes() == expect.tobytes()
    single = devs[0].render(ph, total, seed=4, first_sample=3, accum=start.copy())
    assert np.allclose(got, single, rtol=1e-5, atol=1e-6)
    assert calls and calls[-1] == total and all(b >= a for a, b in zip(calls, calls[1:]))
    assert pt.multi_last_reduce_ms() >= 0.0
    with pytest.raises(pt.PtxError):  # RCCL refuses two ranks on one device: reported, not attempted
        pt.render_multi(devs, ph, total, reduce=pt.REDUCE_RCCL)
    with pytest.raises(pt.PtxError):
        pt.render_multi([devs[0], devs[0]], ph, total)
    for d in devs:
        d.close()


def test_cli_gpus_flag_shards_and_matches_the_library(tmp_path):
    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "pathtracer-cli")
    obj = os.path.join(SCENES, "cornell.obj")
    out = str(tmp_path / "multi.png")
    r = subprocess.run([exe, "-i", obj, "-o", out, "-s", "48x40", "-m", "6", "-t", "2", "-n", "3", "--seed", "5", "--gpus", "2",
                        "--share-device", "--reduce", "peer"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    assert "Sharding the iterations over 2 GPU(s)" in r.stdout and "Reduce over 2 GPU(s)" in r.stdout
    scene = pt.LoadedScene(obj, max_bounces=6)
    devs = [scene.upload(), scene.upload()]
    acc = pt.render_multi(devs, scene.pinhole(48, 40), 6, seed=5, reduce=pt.REDUCE_PEER)
    ref = str(tmp_path / "ref.png")
    pt.write_png(ref, pt.to_rgb8(acc, 6))
    assert open(out, "rb").read() == open(ref, "rb").read()
    for d in devs:
        d.close()
    # one GPU through the RCCL path in a process without PyTorch: librccl is found by the library itself
    one, plain = str(tmp_path / "one.png"), str(tmp_path / "plain.png")
    for extra, dst in ((["--gpus", "1", "--reduce", "rccl"], one), ([], plain)):
        r = subprocess.run([exe, "-i", obj, "-o", dst, "-s", "48x40", "-m", "6", "-t", "2", "-n", "3", "--seed", "5"] + extra,
                           capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
    assert open(one, "rb").read() == open(plain, "rb").read()
    # more GPUs than the machine has: a clear error, no crash
    n = pt.device_count()
    r = subprocess.run([exe, "-i", obj, "-o", out, "-s", "16x16", "--gpus", str(n + 1)], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "visible" in r.stderr


def test_two_handles_share_a_gpu_with_the_default_budget():
    """the automatic queue budget is half of what the scene could have when the call starts,
    so a second handle still finds memory; trim gives it all back."""
    import torch

    scene = pt.LoadedScene(os.path.join(SCENES, "cornell.obj"), max_bounces=5)
    a, b = scene.upload(), scene.upload()
    ph = scene.pinhole(512, 512)
    want = a.render(ph, 64, seed=2)
    held_a = a.memory()
    assert held_a > 0
    free_before, total = torch.cuda.mem_get_info()
    got = b.render(ph, 64, seed=2)  # while a still holds its queues
    assert got.tobytes() == want.tobytes()
    assert b.memory() <= 0.55 * (free_before + 0)  # never more than about half of what was free
    a.trim()
    b.trim()
    assert a.memory() == 0 and b.memory() == 0
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after >= free_before  # everything went back to the device
    again = a.render(ph, 64, seed=2)  # and the next render simply allocates again
    assert again.tobytes() == want.tobytes()
    a.set_queue_slots(1 << 16)  # fewer slots than held: released at once, rebuilt narrower
    assert a.memory() == 0
    narrow = a.render(ph, 64, seed=2)
    assert narrow.tobytes() == want.tobytes()
    a.set_queue_slots(100)  # under 64 slots per segment: fewer segments, not an ignored budget
    tiny = a.render(scene.pinhole(64, 64), 4, seed=2)
    assert tiny.tobytes() == b.render(scene.pinhole(64, 64), 4, seed=2).tobytes()
    assert a.memory() < (8 << 20) + 2 * 17 * 1024 * 512 * 8  # queues for ~128 slots + the stack spill areas
    a.close()
    b.close()


def test_overlapping_batch_calls_on_two_streams_are_serialised():
    """the *_batch_device calls of one handle share one cursor and one spill area; issued
    back to back on two streams they must still give the serial answers."""
    import torch

    path = os.path.join(SCENES, "cornell_teapot.obj")
    scene = pt.LoadedScene(path)
    dev = scene.upload()
    a = scene.arrays()
    lo, hi = a["triangles"].reshape(-1, 3).min(0), a["triangles"].reshape(-1, 3).max(0)
    rng = np.random.default_rng(7)
    n = 1 << 20
    sets = []
    for k in range(2):
        o = (lo + rng.random((n, 3)) * (hi - lo)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        sets.append(pt.make_rays(o, d))
    want = [dev.intersect(r) for r in sets]
    d_rays = [torch.from_numpy(r.view(np.uint8).copy()).cuda() for r in sets]
    d_hits = [torch.zeros(n * 16, dtype=torch.uint8, device="cuda") for _ in sets]
    d_occ = torch.zeros(n, dtype=torch.uint8, device="cuda")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for rep in range(3):
        for h in d_hits:
            h.zero_()
        torch.cuda.synchronize()
        dev.intersect_device(d_rays[0].data_ptr(), n, d_hits[0].data_ptr(), stream=streams[0].cuda_stream)
        dev.intersect_device(d_rays[1].data_ptr(), n, d_hits[1].data_ptr(), stream=streams[1].cuda_stream)
        dev.occluded_device(d_rays[0].data_ptr(), n, d_occ.data_ptr(), stream=streams[0].cuda_stream)
        torch.cuda.synchronize()
        for h, w in zip(d_hits, want):
            assert h.cpu().numpy().tobytes() == w.tobytes()
        assert (d_occ.cpu().numpy().astype(bool) == (want[0]["index"] != pt.MISS)).all()
    dev.close()


def test_distinct_gpus_rccl_equals_peer_fold_equals_one_gpu():
    """The paths a one-GPU box cannot reach: one handle per DISTINCT GPU, summed with RCCL (`ncclCommInitAll` over the
    devices, the grouped `ncclReduce`) and with the peer fold (`hipMemcpyPeerAsync` between devices).  Skipped unless
    two devices are visible - until it has run, multi-GPU results are unverified (README.md)."""
    n = min(pt.device_count(), 4)
    if n < 2:
        pytest.skip("needs at least two GPUs")
    scene = pt.LoadedScene(os.path.join(SCENES, "cornell_textured.obj"), max_bounces=6)
    ph = scene.pinhole(96, 64)
    devs = [scene.upload(device=g) for g in range(n)]
    total = 4 * n + 1
    peer = pt.render_multi(devs, ph, total, seed=5, first_sample=3, reduce=pt.REDUCE_PEER)
    # the peer fold is the left fold of the shards' own renders, in shard order: reproduce it on GPU 0 alone
    acc = None
    for g in range(n):
        first, count = pt.shard_samples(3, total, n, g)
        part = devs[0].render(ph, count, seed=5, first_sample=first)
        acc = part if acc is None else acc + part
    assert peer.tobytes() == acc.tobytes()
    rccl = pt.render_multi(devs, ph, total, seed=5, first_sample=3, reduce=pt.REDUCE_RCCL)
    assert np.allclose(rccl, peer, rtol=2e-6, atol=1e-6)  # same samples; the ring sums in another order
    auto = pt.render_multi(devs, ph, total, seed=5, first_sample=3)
    assert np.allclose(auto, peer, rtol=2e-6, atol=1e-6)
    one = devs[0].render(ph, total, seed=5, first_sample=3)
    assert np.allclose(one, peer, rtol=2e-6, atol=1e-6)
    for d in devs:
        d.close()
