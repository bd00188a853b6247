#!/usr/bin/env python3
"""bench.py — render hot path throughput on MI355X (contract: see the task statement).

A step = one pass of the wavefront path tracer (generate → [extend → shade → shadow] x max_bounces →
accumulate) over one batch of `--spp-per-step` samples of every pixel of the workload; inputs (scene,
accumulation buffer) are resident in HBM before the timed region.  Default workload = BASELINE.json
configs[2]: cornell_teapot.obj, 1080x1080, 4 steps x 640 spp = 2560 spp per GPU, max_bounces 10 (one step = one
batch of 640 samples of every pixel = 747 M paths in flight, ~130 GB of queues out of the 288 GB; wider batches
amortise the long-ray tail of every launch and the small late bounces: 128 / 256 / 640 spp per batch =
1.26 / 1.44 / 1.55 G samples/s in round 1; the library sizes its batches from the free device memory).  The line also
carries `config3_1024spp`: BASELINE.json configs[2] literally, one ptx_render_device call of 1024 spp.

Multi-GPU: one process per GPU (torchrun), the scene replicated, every rank renders its own sample
range (global sample index = rank * steps * spp + ...; counter-based sampler, so ranks never overlap),
and the per-GPU accumulation buffers are summed on rank 0 with ONE RCCL reduce inside the timed region
(the analogue of `buffers.reduce(Add::add)`, tracing/src/worker.rs:106-107).  Weak scaling.

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


def main():
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
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import pathtracer_amd as pt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with stdout_to_stderr():  # RCCL prints a version banner on stdout when its communicator comes up
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.all_reduce(torch.zeros(16, device="cuda"))
            torch.cuda.synchronize()

    W, H = (int(x) for x in args.size.lower().split("x"))
    obj = os.path.join(ROOT, "scenes", args.scene + ".obj")
    t0 = time.perf_counter()
    scene = pt.LoadedScene(obj, max_bounces=args.max_bounces)
    load_s = time.perf_counter() - t0
    flags = 0 if args.no_events else pt.FLAG_TIME_KERNELS
    dev = scene.upload(flags=flags, device=torch.cuda.current_device())  # this rank's GPU, named explicitly
    if args.batch_samples:
        dev.set_batch_samples(args.batch_samples)
    pinhole = scene.pinhole(W, H)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    spp = args.spp_per_step
    base_sample = rank * (args.steps + args.warmup) * spp

    def step(i):
        dev.render_device(pinhole, spp, accum.data_ptr(), seed=0, first_sample=base_sample + i * spp, stream=stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if use_dist:  # warm the communicator too
        dist.reduce(torch.zeros(16, device="cuda"), dst=0)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = dev.stats()
    counts = torch.tensor([st["extend_rays"], st["shadow_rays"], st["samples"]], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    extend_rays, shadow_rays, samples = (float(x) for x in counts.tolist())

    out = None
    if rank == 0:
        value = samples / elapsed
        out = {
            "metric": "samples/s (pixel samples, all kernels + RCCL reduce) with extend Mrays/s, %s %s" % (args.scene, args.size),
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
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
        if not args.no_events and st["extend_ms"] > 0:
            # per-GPU kernel figures from rank 0's HIP events over the timed region
            ext_ms, sh_ms = st["extend_ms"], st["shadow_ms"]
            out["extend_mrays_per_s"] = st["extend_rays"] / ext_ms / 1e3
            out["shadow_mrays_per_s"] = st["shadow_rays"] / sh_ms / 1e3 if sh_ms > 0 else None
            if st["extend_launches"] == args.steps * args.max_bounces:
                # no path regeneration: round r is bounce r, and every path shoots exactly one camera ray
                r_ext, _ = dev.round_times(args.max_bounces)
                camera_rays = float(args.steps) * W * H * spp
                if r_ext[0] > 0 and ext_ms > r_ext[0]:
                    out["extend_camera_mrays_per_s"] = camera_rays / r_ext[0] / 1e3
                    out["extend_bounced_mrays_per_s"] = (st["extend_rays"] - camera_rays) / (ext_ms - r_ext[0]) / 1e3
            out["kernel_ms"] = {"extend": ext_ms, "shadow": sh_ms, "shade": st["shade_ms"], "generate": st["generate_ms"], "resolve_accumulate": st["other_ms"]}
            # algorithmic bytes per extend ray: 32 ray + 16 hit + 8 per inner node + 40 per triangle test,
            # node/triangle counts measured on this workload by a counting pass outside the timed region
            # The counts are those of the REFERENCE algorithm (FLAG_NO_BOX_CLIP: every ray walked over its whole range, equal
            # to the CPU oracle's counters, tests/test_gpu_parity.py), as SURVEY 8d defines B_ray; the product kernels narrow
            # every ray to the bounding box of the triangles and visit fewer nodes for the same hits: "visited" below.
            tmp = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
            counted = {}
            for label, fl in (("reference", pt.FLAG_COUNT_WORK | pt.FLAG_NO_BOX_CLIP), ("visited", pt.FLAG_COUNT_WORK)):
                cdev = scene.upload(flags=fl, device=torch.cuda.current_device())
                cdev.render_device(pinhole, min(spp, 4), tmp.data_ptr(), seed=0, first_sample=base_sample, stream=stream)
                torch.cuda.synchronize()
                c = cdev.stats()
                cdev.close()
                counted[label] = (c["extend_inner_nodes"] / max(c["extend_rays"], 1), c["extend_tri_tests"] / max(c["extend_rays"], 1))
            nodes_per_ray, tris_per_ray = counted["reference"]
            b_ray = 48.0 + 8.0 * nodes_per_ray + 40.0 * tris_per_ray
            b_ray_visited = 48.0 + 8.0 * counted["visited"][0] + 40.0 * counted["visited"][1]
            launches = st["extend_launches"]
            achieved = st["extend_rays"] * b_ray / (ext_ms * 1e-3) / 1e9
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
                ex_achieved = ex["extend_rays"] * b_ray / (ex["extend_ms"] * 1e-3) / 1e9
            tr = measured_traffic(args.scene, args.size)
            hbm_per_ray = tr["extend_bytes_per_ray"] if tr else None
            rays_per_s = st["extend_rays"] / (ext_ms * 1e-3)
            out["roofline"] = {
                "kernel": "k_traverse<nearest> (extend)",
                # priced against HBM as SURVEY 8d defines it (algorithmic = cache-level bytes); what the counters
                # name as the limiter is in "limited_by", and the HBM the kernel really moves in "hbm_measured"
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": hbm_per_ray * st["extend_rays"] / max(launches, 1) if hbm_per_ray else None,
                "traffic_note": ("HBM bytes per launch = PMC bytes per ray (%s) x rays per launch" % tr["source"]) if tr
                                else "no PMC summary committed for this scene@size (profiles/hbm_traffic.json)",
                "limited_by": (tr or {}).get("limited_by", "VALU issue under divergence + dependent-load latency (see DESIGN.md 5); not HBM"),
                "hbm_measured": {"bytes_per_ray": hbm_per_ray, "GBps": hbm_per_ray * rays_per_s / 1e9,
                                 "frac": hbm_per_ray * rays_per_s / 1e9 / HBM_PEAK_GBS} if hbm_per_ray else None,
                "bytes_per_ray": b_ray,
                "visited": {"note": "what the kernel itself walks after narrowing every ray to the bounding box of the triangles (same hits)",
                            "inner_nodes_per_ray": counted["visited"][0], "tri_tests_per_ray": counted["visited"][1],
                            "bytes_per_ray": b_ray_visited, "achieved": st["extend_rays"] * b_ray_visited / (ext_ms * 1e-3) / 1e9,
                            "frac": st["extend_rays"] * b_ray_visited / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "inner_nodes_per_ray": nodes_per_ray,
                "tri_tests_per_ray": tris_per_ray,
                "launches": launches,
                "avg_launch_ms": ext_ms / max(launches, 1),
                "compulsory_hbm_frac": st["extend_rays"] * 48.0 / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "concurrent_batches": pipelines,
            }
            if ex is not None:
                out["roofline"]["exclusive"] = {
                    "achieved": ex_achieved, "frac": ex_achieved / HBM_PEAK_GBS,
                    "extend_mrays_per_s": ex["extend_rays"] / ex["extend_ms"] / 1e3,
                    "avg_launch_ms": ex["extend_ms"] / max(ex["extend_launches"], 1),
                    "sample": "one batch of %d spp alone on the GPU (ptx_set_pipelines(1)), outside the timed region" % min(spp, 64)}
        if world == 1 and not args.no_config3 and not args.batch_samples:
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
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(obj, W, H, args.max_bounces)
        if args.save:
            img = pt.to_rgb8(accum.cpu().numpy().reshape(H, W, 3), int(samples / (W * H)))
            pt.write_png(args.save, img)
    dev.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


def cpu_baseline(obj, W, H, max_bounces):
    """The oracle (CPU restatement of the reference renderer, its T-threads x N-iterations scheme) on a
    bounded sample of the same workload: full frame, one iteration per thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc

    cores = min(16, os.cpu_count() or 1)
    ref = orc.Scene(obj, max_bounces=max_bounces)
    iters = 4  # ~10-15 s of CPU work at 1080^2 on 16 cores
    _, secs, st = ref.render(W, H, threads=cores, iterations_per_thread=iters, mode=0, seed=1, stats=True)
    return {
        "value": cores * iters * W * H / secs,
        "unit": "samples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%dx%d full frame, -t %d -n %d (%d spp), sequential per-thread sampler; %.1f s" % (W, H, cores, iters, cores * iters, secs),
        "extend_mrays_per_s": st["extend_rays"] / secs / 1e6,
        "rays_per_s": (st["extend_rays"] + st["shadow_rays"]) / secs,
    }


if __name__ == "__main__":
    main()
