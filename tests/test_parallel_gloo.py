"""world_size-2 gloo test of the multi-GPU path on CPU: sample sharding + one reduce of the accumulation
buffers (pathtracer_amd/parallel.py).  The per-rank renderer is replaced by the oracle (tests may use the
checker; the product itself has no CPU path), so the test shows that rank-sharded sample ranges with a
summed buffer equal the single-process render of the same samples."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, SEED = 24, 20, 6, 31


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    from pathtracer_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = orc.Scene(os.path.join(ROOT, "scenes", "cornell.obj"))
    accum = torch.zeros(H * W * 3, dtype=torch.float32)

    def render_fn(first, count, buf):  # stand-in for DeviceScene.render_device: ADDS samples into buf
        a = buf.numpy().reshape(H, W, 3)
        ref.render(W, H, threads=1, iterations_per_thread=count, mode=1, seed=SEED, first_sample=first, accum=a)

    first, count = parallel.render_sharded(render_fn, SPP, accum, first_sample=3)
    assert (first, count) == parallel.shard_samples(SPP, world, rank)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_samples_partitions_exactly():
    from pathtracer_amd.parallel import shard_samples

    for total in (0, 1, 5, 8, 1024, 2048, 4097):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_samples(total, world, r) for r in range(world)]
            assert sum(c for _, c in ranges) == total
            pos = 0
            for f, c in ranges:
                assert f == pos
                pos += c
            assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1
    with pytest.raises(ValueError):
        shard_samples(4, 2, 2)


def test_two_ranks_gloo_match_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc

    out = str(tmp_path / "sum.npy")
    mp.start_processes(_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out).reshape(H, W, 3)
    ref = orc.Scene(os.path.join(ROOT, "scenes", "cornell.obj"))
    want, _ = ref.render(W, H, threads=1, iterations_per_thread=SPP, mode=1, seed=SEED, first_sample=3)
    # same samples; only the order of the float additions differs (two partial sums vs one running sum)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    a, _ = ref.render(W, H, threads=1, iterations_per_thread=3, mode=1, seed=SEED, first_sample=3)
    b, _ = ref.render(W, H, threads=1, iterations_per_thread=3, mode=1, seed=SEED, first_sample=6)
    assert (got == a + b).all()  # bit-exact against the same two partial sums
