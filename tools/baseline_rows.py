#!/usr/bin/env python3
"""BASELINE.md section 3 rows, measured on the machine this runs on (the GPU box): per scene the kdtree-cli
statistics of the tree the product builder makes, the CPU baseline (the oracle = C++ restatement of the
reference renderer, T threads x N iterations like tracing/src/worker.rs:84-119, sequential per-thread sampler)
and the GPU numbers beside it.  One JSON object per line on stdout.
usage: python tools/baseline_rows.py [--threads 16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

ROWS = [  # run, scene, W, H, CPU iterations per thread, GPU samples (configs of BASELINE.json)
    ("C1", "cornell", 256, 256, 1, 16),
    ("C3", "cornell_teapot", 1080, 1080, 2, 1024),
    ("C4", "cornell_textured", 1080, 1080, 2, 2048),
    ("C5a", "sr2", 3840, 2160, 1, 512),
    ("C5b", "cubesplosion", 3840, 2160, 1, 512),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="GPU numbers only (after a kernel change)")
    args = ap.parse_args()
    import pathtracer_amd as pt
    import pyoracle as orc

    if not args.no_gpu:
        import torch
    for run, name, w, h, n_iter, gpu_spp in ROWS:
        obj = os.path.join(ROOT, "scenes", name + ".obj")
        t0 = time.perf_counter()
        scene = pt.LoadedScene(obj)
        load_build_s = time.perf_counter() - t0
        st = scene.kdtree().stats()
        row = {"run": run, "scene": name, "size": "%dx%d" % (w, h), "triangles": scene.n_triangles, "load_and_build_s": load_build_s,
               "tree": {k: (float(v) if isinstance(v, float) else int(v)) for k, v in st.items()}}
        ref = None if args.no_cpu else orc.Scene(obj)
        T = args.threads
        if ref is not None:
            _, secs, rs = ref.render(w, h, threads=T, iterations_per_thread=n_iter, mode=0, seed=1, stats=True)
            row["cpu"] = {"kind": "port (oracle, C++ restatement, -O2 -ffp-contract=off)", "threads": T, "cores": os.cpu_count(),
                          "iterations_per_thread": n_iter, "spp": T * n_iter, "seconds": secs, "s_per_iteration": secs / n_iter,
                          "samples_per_s": T * n_iter * w * h / secs, "extend_mrays_per_s": rs["extend_rays"] / secs / 1e6,
                          "shadow_mrays_per_s": rs["shadow_rays"] / secs / 1e6}
        if not args.no_gpu:
            dev = scene.upload(flags=pt.FLAG_TIME_KERNELS)
            ph = scene.pinhole(w, h)
            buf = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
            stream = torch.cuda.current_stream().cuda_stream
            # warm-up with the call that is timed: the queues get their final size, and memory that an earlier handle of
            # this process gave back is handed out again (the driver scrubs it first: ~1.4 s per 127 GB, measured with
            # tools/call_sequence_probe.py) before the clock starts
            dev.render_device(ph, gpu_spp, buf.data_ptr(), seed=0, first_sample=0, stream=stream)
            torch.cuda.synchronize()
            dev.stats_reset()
            t0 = time.perf_counter()
            dev.render_device(ph, gpu_spp, buf.data_ptr(), seed=0, first_sample=64, stream=stream)
            torch.cuda.synchronize()
            secs = time.perf_counter() - t0
            g = dev.stats()
            row["gpu"] = {"spp": gpu_spp, "seconds": secs, "samples_per_s": gpu_spp * w * h / secs,
                          "extend_mrays_per_s": (g["extend_rays"] - g.get("culled_camera_rays", 0)) / g["extend_ms"] / 1e3, "culled_camera_rays_per_sample": g.get("culled_camera_rays", 0) / (gpu_spp * w * h), "shadow_mrays_per_s": g["shadow_rays"] / g["shadow_ms"] / 1e3 if g["shadow_ms"] else None,
                          "extend_rays_per_sample": g["extend_rays"] / (gpu_spp * w * h), "shadow_rays_per_sample": g["shadow_rays"] / (gpu_spp * w * h),
                          "queue_bytes": dev.memory()}
            if "cpu" in row:
                row["gpu_over_cpu"] = row["gpu"]["samples_per_s"] / row["cpu"]["samples_per_s"]
            dev.close()
            del buf
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
