#!/usr/bin/env python3
"""f-1 (per-lobe shade queues): what could lobe-sorted queues buy?  Renders a scene with the library named by
PTX_LIB and prints k_shade's time per shaded path and its share of the frame.  Run it with the production library
and with the probe builds -DPTX_FORCE_LOBE=1/3 (every path takes the specular / the diffuse lobe: k_shade without
any lobe divergence).  usage: PTX_LIB=... python tools/lobe_probe.py [scene] [spp] [--material-queues]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pathtracer_amd as pt

args = [a for a in sys.argv[1:] if not a.startswith("--")]
keyed = "--material-queues" in sys.argv  # PTX_FLAG_MATERIAL_QUEUES: shade grouped by the material of the hit
name = args[0] if len(args) > 0 else "cornell_textured"
spp = int(args[1]) if len(args) > 1 else 256
scene = pt.LoadedScene(os.path.join(ROOT, "scenes", name + ".obj"))
dev = scene.upload(flags=pt.FLAG_TIME_KERNELS | (pt.FLAG_MATERIAL_QUEUES if keyed else 0))
ph = scene.pinhole(1080, 1080)
buf = torch.zeros(1080 * 1080 * 3, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
dev.render_device(ph, spp, buf.data_ptr(), seed=0, first_sample=0, stream=stream)
torch.cuda.synchronize()
dev.stats_reset()
t0 = time.perf_counter()
dev.render_device(ph, spp, buf.data_ptr(), seed=0, first_sample=spp, stream=stream)
torch.cuda.synchronize()
secs = time.perf_counter() - t0
st = dev.stats()
tot = st["extend_ms"] + st["shadow_ms"] + st["shade_ms"] + st["generate_ms"] + st["other_ms"]
print("%-40s %-18s shade %.1f ms = %.1f %% of the kernels, %.3f ns per shaded path (%.0f M paths, %.2f shadow rays per path); extend %.0f shadow %.0f Mrays/s; %.0f Msamples/s"
      % (os.path.basename(os.environ.get("PTX_LIB", "libptx_mi355x.so")) + (" +material queues" if keyed else ""), name, st["shade_ms"], 100 * st["shade_ms"] / tot, st["shade_ms"] * 1e6 / st["extend_rays"],
         st["extend_rays"] / 1e6, st["shadow_rays"] / st["extend_rays"], st["extend_rays"] / st["extend_ms"] / 1e3, st["shadow_rays"] / max(st["shadow_ms"], 1e-9) / 1e3,
         spp * 1080 * 1080 / secs / 1e6))
