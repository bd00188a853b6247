"""What a render call costs as a function of what ran before it in the process (queue reuse, growth, rebuilds,
memory handed back by a closed handle).  usage: python tools/call_sequence_probe.py scene:W:H:spp,spp,... [scene:...]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pathtracer_amd as pt

for spec in sys.argv[1:]:
    name, w, h, spps = spec.split(":")
    w, h = int(w), int(h)
    scene = pt.LoadedScene(os.path.join(ROOT, "scenes", name + ".obj"))
    free0 = torch.cuda.mem_get_info()[0]
    dev = scene.upload(flags=pt.FLAG_TIME_KERNELS)
    ph = scene.pinhole(w, h)
    buf = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for spp in [int(x) for x in spps.split(",")]:
        dev.stats_reset()
        free = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter()
        dev.render_device(ph, spp, buf.data_ptr(), seed=0, first_sample=0, stream=stream)
        torch.cuda.synchronize()
        s = time.perf_counter() - t0
        g = dev.stats()
        kern = sum(g[k] for k in g if k.endswith("_ms"))
        print(json.dumps({"scene": name, "spp": spp, "seconds": round(s, 4), "kernel_ms": round(kern, 1), "G_samples_per_s": round(spp * w * h / s / 1e9, 3),
                          "free_before_GB": round(free / 1e9, 1), "held_after_GB": round(dev.memory() / 1e9, 1), "extend_launches": g.get("extend_launches")}), flush=True)
    dev.close()
    del buf
    print(json.dumps({"scene": name, "free_at_upload_GB": round(free0 / 1e9, 1), "free_after_close_GB": round(torch.cuda.mem_get_info()[0] / 1e9, 1)}), flush=True)
