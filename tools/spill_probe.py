import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pathtracer_amd as pt
for name, W, H, spp in (("cornell_teapot", 1080, 1080, 32), ("sr2", 1920, 1080, 16), ("cubesplosion", 1920, 1080, 16)):
    scene = pt.LoadedScene("scenes/%s.obj" % name, max_bounces=10)
    dev = scene.upload(flags=pt.FLAG_COUNT_WORK)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    dev.render_device(scene.pinhole(W, H), spp, acc.data_ptr(), seed=0, first_sample=0, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = dev.stats(); pf = dev.loop_profile()
    for k, rays in (("extend", st["extend_rays"]), ("shadow", st["shadow_rays"])):
        p = pf[k]
        print("%s %s: rays %d  wave iterations %d  spill pushes/ray %.3f  spill pops/ray %.3f  pops per wave iteration %.3f" % (name, k, rays, p["iters"], p["refills"] / rays, p["live_lanes"] / rays, p["live_lanes"] / max(p["iters"], 1)), flush=True)
    dev.close()
