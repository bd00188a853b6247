"""Sample sharding over GPUs: one process per GPU, scene replicated, per-GPU accumulation buffers summed
with ONE collective — the multi-GPU form of `render_parallel_iterations` (tracing/src/worker.rs:84-119:
T private buffers, `buffers.reduce(Add::add)`).

The path shards by independent units (samples), so there is no data-path collective; the only exchange is
the final buffer sum (`torch.distributed.reduce`, RCCL over xGMI with backend "nccl", gloo in CPU tests).
Because the sampler is counter-based (include/ptx_sampling.h), the image does not depend on the number
of ranks except for the order in which the per-rank partial sums are added.
"""


def shard_samples(total_samples, world_size, rank):
    """Contiguous sample range [first, first+count) of `rank` (GPU g gets [g*S/N, (g+1)*S/N))."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    from . import shard_samples as _abi_shard  # one definition: ptx_shard_samples of the C ABI

    return _abi_shard(0, total_samples, world_size, rank)


def render_sharded(render_fn, total_samples, accum, first_sample=0, group=None, dst=0):
    """Renders this rank's share with `render_fn(first, count, accum)` (which ADDS into the torch tensor
    `accum`), then sums all ranks' buffers onto `dst`.  Returns (first, count) of this rank.

    `render_fn` is the GPU renderer in the product (DeviceScene.render_device on accum.data_ptr()); the CPU
    tests inject a stand-in so that the sharding and the reduce are exercised with gloo."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    first, count = shard_samples(total_samples, world, rank)
    if count:
        render_fn(first_sample + first, count, accum)
    if world > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return first, count


def render_image_sharded(scene, pinhole, total_samples, seed=0, first_sample=0, device=None):
    """Whole-job helper: every rank calls it; rank 0 gets the summed H x W x 3 float32 numpy image, the
    others None.  `scene` is a pathtracer_amd.DeviceScene living on this rank's GPU."""
    import torch
    import torch.distributed as dist

    dev = device or torch.device("cuda", torch.cuda.current_device())
    accum = torch.zeros(pinhole.height * pinhole.width * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def fn(first, count, buf):
        scene.render_device(pinhole, count, buf.data_ptr(), seed=seed, first_sample=first, stream=stream)

    render_sharded(fn, total_samples, accum, first_sample=first_sample)
    torch.cuda.synchronize()
    rank = dist.get_rank() if dist.is_initialized() else 0
    return accum.cpu().numpy().reshape(pinhole.height, pinhole.width, 3) if rank == 0 else None
