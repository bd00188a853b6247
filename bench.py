#!/usr/bin/env python3
"""bench.py — render hot path throughput on MI355X (contract: see the task statement).

A step = one pass of the wavefront path tracer (generate → [extend → shade → shadow] x max_bounces →
accumulate) over one batch of `--spp-per-step` samples of every pixel of the workload; inputs (scene,
accumulation buffer) are resident in HBM before the timed region.  Default workload = BASELINE.json
configs[2]: cornell_teapot.obj, 1080x1080, 4 steps x 640 spp = 2560 spp per GPU, max_bounces 10 (one step = one
batch of 640 samples of every pixel = 747 M paths in flight, ~130 GB of queues out of the 288 GB; wider batches
amortise the long-ray tail of every launch and the small late bounces; the library sizes its batches
from the free device memory).  The line also carries `config3_1024spp`: BASELINE.json configs[2] literally, one
ptx_render_device call of 1024 spp.

Multi-GPU (weak scaling: per-GPU work fixed), two forms of the same partition:
  * one process per GPU (default): the scene replicated, every rank renders its own sample range (global sample
    index = rank * steps * spp + ...; counter-based sampler, so ranks never overlap), and the per-GPU accumulation
    buffers are summed on rank 0 with ONE RCCL reduce inside the timed region (the analogue of
    `buffers.reduce(Add::add)`, tracing/src/worker.rs:106-107).  `python bench.py --gpus N` needs no external
    launcher: with N > 1 and no WORLD_SIZE in the environment this process - before it imports torch or touches a
    GPU - starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD, relays rank 0's ONE
    JSON line and exits with the child's status (the reference's driver, `render_parallel_iterations`
    worker.rs:84-119, spawns its own workers too).  Under torchrun (WORLD_SIZE set) it is a rank.
  * `--in-process`: ONE process, one host thread per GPU behind the C ABI (`ptx_render_multi`: one handle per GPU,
    RCCL reduce onto GPU 0 from one `ncclGroup`), timed as one call of steps x spp x N samples; reports `reduce_ms`.
PTX_BENCH_DRYRUN=1 (CPU rehearsal, tests/test_bench_launch.py): the ranks use gloo and a stand-in for the renderer,
so the launcher, the rendezvous, the reduce and the one-line contract run on a machine without a GPU; the line says
"dryrun": true and is no measurement.

value = whole-job pixel samples per second.  The line also carries extend Mrays/s, the roofline of the
dominant kernel (extend) from HIP-event timings taken inside the timed region, and a CPU baseline
(the oracle — a C++ restatement of the reference, kind "port" — on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# HBM bytes per extend ray come from the committed summary of the PMC passes (profiles/hbm_traffic.json, written by
# tools/pmc_summary.py --json from separate FETCH_SIZE / WRITE_SIZE runs of this bench; FETCH_SIZE doubled as
# MI355X_MICROARCH.md "HBM" prescribes for gfx950), keyed by scene@size: a workload nobody profiled gets null.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "hbm_traffic.json")


def measured_traffic(scene, size):
    try:
        with open(TRAFFIC_FILE) as f:
            return json.load(f).get("%s@%s" % (scene, size.lower()))
    except (OSError, ValueError):
        return None


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: native libraries that print banners on stdout
    cannot get between the driver and the ONE JSON line this script owes it."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell_teapot")
    ap.add_argument("--size", default="1080x1080")
    ap.add_argument("--spp-per-step", type=int, default=640)
    ap.add_argument("--max-bounces", type=int, default=10)
    ap.add_argument("--batch-samples", type=int, default=0, help="samples in flight per launch batch (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--save", default="", help="write the tonemapped image (rank 0) to this PNG")
    ap.add_argument("--force-dist", action="store_true", help="run the torch.distributed path even with one rank (rehearsal)")
    ap.add_argument("--no-config3", action="store_true", help="skip the extra one-call 1024-spp figure (BASELINE.json configs[2] literally)")
    ap.add_argument("--in-process", action="store_true", help="one process, one host thread per GPU: time ptx_render_multi over --gpus handles")
    ap.add_argument("--reduce", default="auto", choices=["auto", "rccl", "peer"], help="--in-process: how the per-GPU buffers are summed")
    ap.add_argument("--share-device", action="store_true", help="--in-process rehearsal on a one-GPU box: all handles on GPU 0 (peer fold)")
    return ap.parse_args(argv)


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start one fresh rank process per GPU
    (torch.distributed.run as a CHILD; this parent has not imported torch and never touches a GPU), relay rank 0's ONE
    JSON line on stdout - everything else the children print goes to stderr - and return the child's exit status."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")             # (torchrun would set it, with a warning on stderr)
    print("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        t = out.strip()
        if line is None and t.startswith("{") and '"metric"' in t:
            line = t
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("[bench] the ranks ended without a result line", file=sys.stderr)
        rc = 1
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.in_process:
        return run_in_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args)


def run_rank(args):
    dryrun = os.environ.get("PTX_BENCH_DRYRUN", "") not in ("", "0")
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    import pathtracer_amd as pt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start `python bench.py --gpus N` without a launcher, or torchrun with --nproc-per-node N)" % (args.gpus, world))
    where = "cpu" if dryrun else "cuda"
    if not dryrun:
        torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with stdout_to_stderr():  # RCCL prints a version banner on stdout when its communicator comes up
            if dryrun:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.all_reduce(torch.zeros(16, device=where))
            if not dryrun:
                torch.cuda.synchronize()

    W, H = (int(x) for x in args.size.lower().split("x"))
    obj = os.path.join(ROOT, "scenes", args.scene + ".obj")
    t0 = time.perf_counter()
    scene = pt.LoadedScene(obj, max_bounces=args.max_bounces)
    load_s = time.perf_counter() - t0
    flags = 0 if args.no_events else pt.FLAG_TIME_KERNELS
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=where)
    spp = args.spp_per_step
    base_sample = rank * (args.steps + args.warmup) * spp
    if dryrun:
        dev = DryrunDevice(accum, W * H)
        pinhole, stream = None, None
    else:
        dev = scene.upload(flags=flags, device=torch.cuda.current_device())  # this rank's GPU, named explicitly
        if args.batch_samples:
            dev.set_batch_samples(args.batch_samples)
        pinhole = scene.pinhole(W, H)
        stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        dev.render_device(pinhole, spp, accum.data_ptr(), seed=0, first_sample=base_sample + i * spp, stream=stream)

    def fence():
        if not dryrun:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            if not dryrun:
                torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if use_dist:  # warm the communicator too
        dist.reduce(torch.zeros(16, device=where), dst=0)
    fence()
    accum.zero_()
    dev.stats_reset()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if use_dist:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=where)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = dev.stats()
    counts = torch.tensor([st["extend_rays"], st["shadow_rays"], st["samples"], st.get("culled_camera_rays", 0)], dtype=torch.float64, device=where)
    if use_dist:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    extend_rays, shadow_rays, samples, culled_rays = (float(x) for x in counts.tolist())

    out = None
    if rank == 0:
        out = result_line(args, scene, world, elapsed, samples, extend_rays, shadow_rays, load_s)
        out["culled_camera_rays"] = culled_rays  # of extend_rays: ended at generation (they miss the scene's box), all ranks
        out["config"]["sharding"] = "samples per GPU, one process per GPU, one RCCL reduce"
        if dryrun:
            out["dryrun"] = True
            out["metric"] = "DRYRUN (no GPU, stand-in renderer, gloo): not a measurement; " + out["metric"]
            out["dryrun_sum_ok"] = bool((accum == float(args.steps * spp * world)).all().item())
        elif not args.no_events and st["extend_ms"] > 0:
            kernel_figures(out, args, scene, dev, st, pinhole, base_sample, stream, W, H)
        if not dryrun and world == 1 and not args.no_config3 and not args.batch_samples:
            # BASELINE.json configs[2] literally: ONE ptx_render_device call of 1024 spp (the library cuts it into
            # batches by itself), outside the timed region above
            tmp = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
            dev.stats_reset()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dev.render_device(pinhole, 1024, tmp.data_ptr(), seed=0, first_sample=0, stream=stream)
            torch.cuda.synchronize()
            secs = time.perf_counter() - t1
            c3 = dev.stats()
            out["config3_1024spp"] = {
                "workload": "%s.obj %s, one ptx_render_device call of 1024 spp" % (args.scene, args.size),
                "samples_per_s": 1024.0 * W * H / secs, "seconds": secs,
                "extend_mrays_per_s": c3["extend_rays"] / c3["extend_ms"] / 1e3 if c3["extend_ms"] > 0 else None,
                "extend_launches": c3["extend_launches"]}
            del tmp
        if not dryrun and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(obj, W, H, args.max_bounces)
        if args.save and not dryrun:
            img = pt.to_rgb8(accum.cpu().numpy().reshape(H, W, 3), int(samples / (W * H)))
            pt.write_png(args.save, img)
    dev.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)
    return 0


class DryrunDevice:
    """PTX_BENCH_DRYRUN=1: what a rank renders with when there is no GPU - every sample adds 1.0 to every value of the
    rank's buffer, so the reduced image is known.  It stands in for the renderer in a rehearsal of the launcher; the
    product has no CPU path (pathtracer_amd raises without its HIP library and a device)."""

    def __init__(self, accum, n_pixels):
        self.accum, self.n_pixels, self.samples = accum, n_pixels, 0

    def render_device(self, pinhole, spp, ptr, seed=0, first_sample=0, stream=None):
        self.accum.add_(float(spp))
        self.samples += spp * self.n_pixels

    def stats_reset(self):
        self.samples = 0

    def stats(self):
        return {"extend_rays": 0, "shadow_rays": 0, "samples": self.samples, "extend_ms": 0.0}

    def close(self):
        pass


def result_line(args, scene, n_gpus, elapsed, samples, extend_rays, shadow_rays, load_s):
    spp = args.spp_per_step
    return {
        "metric": "samples/s (pixel samples, all kernels + RCCL reduce) with extend Mrays/s, %s %s" % (args.scene, args.size),
        "value": samples / elapsed,
        "unit": "samples/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "scene files shipped in scenes/ (the reference's own resources); synthetic camera samples",
        "config": {"workload": "%s.obj %s, %d spp per GPU (%d steps x %d spp), max_bounces %d" % (args.scene, args.size, args.steps * spp, args.steps, spp, args.max_bounces),
                   "triangles": scene.n_triangles, "kd_nodes": int(scene.desc.n_nodes), "sharding": "samples per GPU, one RCCL reduce"},
        "rays_per_s": (extend_rays + shadow_rays) / elapsed,
        "extend_rays": extend_rays,
        "shadow_rays": shadow_rays,
        "scene_load_and_build_s": load_s,
    }


def kernel_figures(out, args, scene, dev, st, pinhole, base_sample, stream, W, H):
    """Per-GPU kernel figures from this rank's HIP events over the timed region, and the roofline of the dominant kernel."""
    import torch

    import pathtracer_amd as pt

    spp = args.spp_per_step
    ext_ms, sh_ms = st["extend_ms"], st["shadow_ms"]
    # camera rays that miss the scene's bounding box end where they are generated (k_start) and never reach the extend
    # kernel: they count as extend rays of the job (the reference traces them) but not in the kernel's rate
    culled = st.get("culled_camera_rays", 0)
    traversed = st["extend_rays"] - culled
    out["extend_mrays_per_s"] = traversed / ext_ms / 1e3
    out["shadow_mrays_per_s"] = st["shadow_rays"] / sh_ms / 1e3 if sh_ms > 0 else None
    if st["extend_launches"] == args.steps * args.max_bounces:
        # no path regeneration: round r is bounce r, and every path shoots exactly one camera ray
        r_ext, _ = dev.round_times(args.max_bounces)
        camera_rays = float(args.steps) * W * H * spp
        if r_ext[0] > 0 and ext_ms > r_ext[0]:
            out["extend_camera_mrays_per_s"] = (camera_rays - culled) / r_ext[0] / 1e3
            out["extend_bounced_mrays_per_s"] = (st["extend_rays"] - camera_rays) / (ext_ms - r_ext[0]) / 1e3
    out["kernel_ms"] = {"extend": ext_ms, "shadow": sh_ms, "shade": st["shade_ms"], "generate": st["generate_ms"], "resolve_accumulate": st["other_ms"]}
    # Algorithmic bytes per extend ray: 32 ray + 16 hit + 8 per inner node + 40 per triangle test, node and
    # triangle counts measured on this workload by counting passes outside the timed region:
    #   "visited"   what the product kernel itself walks (it narrows every ray to the bounding box of the triangles and
    #               finds the same hits with fewer visits): the figure `achieved` / `frac` are priced with;
    #   "reference" the REFERENCE algorithm's counts (FLAG_NO_BOX_CLIP: every ray walked over its whole range, equal to
    #               the CPU oracle's counters, tests/test_gpu_parity.py) - SURVEY 8d's B_ray literally, kept beside it.
    tmp = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    counted = {}
    for label, fl in (("reference", pt.FLAG_COUNT_WORK | pt.FLAG_NO_BOX_CLIP), ("visited", pt.FLAG_COUNT_WORK)):
        cdev = scene.upload(flags=fl, device=torch.cuda.current_device())
        cdev.render_device(pinhole, min(spp, 4), tmp.data_ptr(), seed=0, first_sample=base_sample, stream=stream)
        torch.cuda.synchronize()
        c = cdev.stats()
        cdev.close()
        c_rays = max(c["extend_rays"] - c.get("culled_camera_rays", 0), 1)  # rays the kernel traversed
        counted[label] = (c["extend_inner_nodes"] / c_rays, c["extend_tri_tests"] / c_rays)
    b_ray_ref = 48.0 + 8.0 * counted["reference"][0] + 40.0 * counted["reference"][1]
    b_ray = 48.0 + 8.0 * counted["visited"][0] + 40.0 * counted["visited"][1]
    launches = st["extend_launches"]
    rays_per_s = traversed / (ext_ms * 1e-3)
    achieved = rays_per_s * b_ray / 1e9
    # with several sample batches in flight (ptx_set_pipelines > 1; the default is 1) launches of different
    # batches share the CUs and the per-launch durations above are inflated: repeat on one batch alone
    pipelines = int(os.environ.get("PTX_PIPES", "1"))
    ex = None
    if pipelines > 1:
        dev.set_pipelines(1)
        dev.set_batch_samples(min(spp, 64))
        dev.stats_reset()
        tmp.zero_()
        dev.render_device(pinhole, min(spp, 64), tmp.data_ptr(), seed=0, first_sample=base_sample, stream=stream)
        torch.cuda.synchronize()
        ex = dev.stats()
        ex_achieved = (ex["extend_rays"] - ex.get("culled_camera_rays", 0)) * b_ray / (ex["extend_ms"] * 1e-3) / 1e9
    tr = measured_traffic(args.scene, args.size)
    hbm_per_ray = tr["extend_bytes_per_ray"] if tr else None
    out["roofline"] = {
        "kernel": "k_traverse<nearest> (extend)",
        # priced against HBM as SURVEY 8d defines it (algorithmic = cache-level bytes); what the counters
        # name as the limiter is in "limited_by", and the HBM the kernel really moves in "hbm_measured"
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": hbm_per_ray * traversed / max(launches, 1) if hbm_per_ray else None,
        "traffic_note": ("HBM bytes per launch = PMC bytes per ray (%s) x rays per launch" % tr["source"]) if tr
                        else "no PMC summary committed for this scene@size (profiles/hbm_traffic.json)",
        "limited_by": (tr or {}).get("limited_by", "VALU issue under divergence + dependent-load latency (see DESIGN.md 5); not HBM"),
        "hbm_measured": {"bytes_per_ray": hbm_per_ray, "GBps": hbm_per_ray * rays_per_s / 1e9,
                         "frac": hbm_per_ray * rays_per_s / 1e9 / HBM_PEAK_GBS} if hbm_per_ray else None,
        "bytes_per_ray": b_ray,
        "bytes_note": "algorithmic bytes of the nodes and triangles the kernel itself visits (48 + 8 x nodes + 40 x tests per ray)",
        "inner_nodes_per_ray": counted["visited"][0],
        "tri_tests_per_ray": counted["visited"][1],
        "reference_algorithm": {"note": "SURVEY 8d's B_ray literally: the reference's KdNode::intersect walks every ray over its whole range "
                                        "(no box clip) and visits more nodes for the same hits; priced with the kernel's time it overstates the kernel",
                                "inner_nodes_per_ray": counted["reference"][0], "tri_tests_per_ray": counted["reference"][1],
                                "bytes_per_ray": b_ray_ref, "achieved": rays_per_s * b_ray_ref / 1e9,
                                "frac": rays_per_s * b_ray_ref / 1e9 / HBM_PEAK_GBS},
        "launches": launches,
        "avg_launch_ms": ext_ms / max(launches, 1),
        "compulsory_hbm_frac": rays_per_s * 48.0 / 1e9 / HBM_PEAK_GBS,
        "concurrent_batches": pipelines,
    }
    if ex is not None:
        out["roofline"]["exclusive"] = {
            "achieved": ex_achieved, "frac": ex_achieved / HBM_PEAK_GBS,
            "extend_mrays_per_s": ex["extend_rays"] / ex["extend_ms"] / 1e3,
            "avg_launch_ms": ex["extend_ms"] / max(ex["extend_launches"], 1),
            "sample": "one batch of %d spp alone on the GPU (ptx_set_pipelines(1)), outside the timed region" % min(spp, 64)}


def run_in_process(args):
    """One process, one host thread per GPU behind the C ABI: `ptx_render_multi` over --gpus handles (GPU g renders the
    samples [g*S/N, (g+1)*S/N) of ONE call of S = steps x spp x N samples, then one RCCL reduce onto GPU 0 - the
    reference's `render_parallel_iterations` with GPUs for threads, tracing/src/worker.rs:84-119).  The call takes and
    returns a HOST buffer, as the reference's driver returns its image: one upload and one read-back of the W x H x 3
    sums (14 MB at 1080 x 1080) sit inside the timed region."""
    import numpy as np
    import torch  # noqa: F401  (first, so that its HIP runtime is the one mapped: pathtracer_amd.lib())

    import pathtracer_amd as pt

    n = args.gpus
    have = pt.device_count()
    if not args.share_device and n > have:
        raise SystemExit("bench.py --in-process: --gpus %d but %d device(s) visible (use --share-device to rehearse on one GPU)" % (n, have))
    W, H = (int(x) for x in args.size.lower().split("x"))
    obj = os.path.join(ROOT, "scenes", args.scene + ".obj")
    t0 = time.perf_counter()
    scene = pt.LoadedScene(obj, max_bounces=args.max_bounces)
    load_s = time.perf_counter() - t0
    flags = 0 if args.no_events else pt.FLAG_TIME_KERNELS
    devs = [scene.upload(flags=flags, device=0 if args.share_device else g) for g in range(n)]
    spp = args.spp_per_step
    for d in devs:
        d.set_batch_samples(args.batch_samples or spp)  # one batch = one step, as in the per-process form
    pinhole = scene.pinhole(W, H)
    how = {"auto": pt.REDUCE_AUTO, "rccl": pt.REDUCE_RCCL, "peer": pt.REDUCE_PEER}[args.reduce]
    accum = np.zeros((H, W, 3), np.float32)
    with stdout_to_stderr():  # RCCL's banner
        if args.warmup:
            pt.render_multi(devs, pinhole, args.warmup * spp * n, seed=0, first_sample=0, accum=accum, reduce=how)
    accum[:] = 0
    for d in devs:
        d.stats_reset()
    total = args.steps * spp * n
    with stdout_to_stderr():
        t0 = time.perf_counter()
        pt.render_multi(devs, pinhole, total, seed=0, first_sample=args.warmup * spp * n, accum=accum, reduce=how)
        elapsed = time.perf_counter() - t0
    reduce_ms = pt.multi_last_reduce_ms()
    stats = [d.stats() for d in devs]
    samples = float(sum(s["samples"] for s in stats))
    extend_rays = float(sum(s["extend_rays"] for s in stats))
    shadow_rays = float(sum(s["shadow_rays"] for s in stats))
    out = result_line(args, scene, n, elapsed, samples, extend_rays, shadow_rays, load_s)
    out["mode"] = "in-process: ptx_render_multi, one host thread per GPU"
    out["reduce_ms"] = reduce_ms
    out["reduce"] = args.reduce if not args.share_device else "peer (handles share GPU 0)"
    out["config"]["sharding"] = "samples per GPU (ptx_shard_samples), one process, one %s reduce onto GPU 0" % ("RCCL" if args.reduce != "peer" else "peer-copy")
    out["host_buffer_bytes_in_timed_region"] = 2 * W * H * 12
    if not args.no_events:
        out["per_gpu"] = [{"extend_mrays_per_s": s["extend_rays"] / s["extend_ms"] / 1e3 if s["extend_ms"] > 0 else None,
                           "shadow_mrays_per_s": s["shadow_rays"] / s["shadow_ms"] / 1e3 if s["shadow_ms"] > 0 else None,
                           "samples": s["samples"]} for s in stats]
    if args.save:
        pt.write_png(args.save, pt.to_rgb8(accum, int(samples / (W * H))))
    for d in devs:
        d.close()
    print(json.dumps(out), flush=True)
    return 0


def cpu_baseline(obj, W, H, max_bounces):
    """The oracle (CPU restatement of the reference renderer, its T-threads x N-iterations scheme) on a
    bounded sample of the same workload: full frame, one iteration per thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc

    cores = min(16, os.cpu_count() or 1)  # threads used (-t); the GPU box gives one job 16 of the host's cores
    ref = orc.Scene(obj, max_bounces=max_bounces)
    iters = 4  # ~10-15 s of CPU work at 1080^2 on 16 cores
    _, secs, st = ref.render(W, H, threads=cores, iterations_per_thread=iters, mode=0, seed=1, stats=True)
    return {
        "value": cores * iters * W * H / secs,
        "unit": "samples/s",
        "cores": cores,
        "threads": cores,
        "host_cores": os.cpu_count(),
        "kind": "port",
        "sample": "%dx%d full frame, -t %d -n %d (%d spp), sequential per-thread sampler; %.1f s" % (W, H, cores, iters, cores * iters, secs),
        "extend_mrays_per_s": st["extend_rays"] / secs / 1e6,
        "rays_per_s": (st["extend_rays"] + st["shadow_rays"]) / secs,
    }


if __name__ == "__main__":
    sys.exit(main())
