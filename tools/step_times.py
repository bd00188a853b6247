import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, pathtracer_amd as pt
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
scene = pt.LoadedScene(os.path.join(ROOT, "scenes", "cornell_teapot.obj"), max_bounces=10)
dev = scene.upload(flags=pt.FLAG_TIME_KERNELS)
W = H = 1080
ph = scene.pinhole(W, H)
acc = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for i in range(8):
    torch.cuda.synchronize(); t = time.perf_counter()
    dev.render_device(ph, 640, acc.data_ptr(), seed=0, first_sample=i * 640, stream=s)
    torch.cuda.synchronize(); print("step %d: %.1f ms" % (i, (time.perf_counter() - t) * 1e3), flush=True)
