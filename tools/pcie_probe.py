#!/usr/bin/env python3
"""What the host-buffer entry point could reach: H2D / D2H rates of pageable and pinned memory and the cost of pinning
a caller's buffer in place (hipHostRegister), 512 MB in / 256 MB out = 2^24 rays."""
import ctypes as C, time
import numpy as np, torch

rt = torch.cuda.cudart()  # torch's HIP runtime
n_in, n_out = 512 << 20, 256 << 20
host_in = np.ones(n_in, np.uint8); host_out = np.zeros(n_out, np.uint8)
d_in = torch.empty(n_in, dtype=torch.uint8, device="cuda"); d_out = torch.empty(n_out, dtype=torch.uint8, device="cuda")
t_in = torch.from_numpy(host_in); t_out = torch.from_numpy(host_out)
def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
a = timed(lambda: d_in.copy_(t_in)); b = timed(lambda: t_out.copy_(d_out))
print("pageable: H2D 512 MB %.1f ms (%.1f GB/s), D2H 256 MB %.1f ms (%.1f GB/s)" % (a * 1e3, n_in / a / 1e9, b * 1e3, n_out / b / 1e9))
t0 = time.perf_counter(); r1 = rt.cudaHostRegister(host_in.ctypes.data, n_in, 0); r2 = rt.cudaHostRegister(host_out.ctypes.data, n_out, 0); t1 = time.perf_counter()
print("hipHostRegister 768 MB: rc %s %s, %.1f ms" % (r1, r2, (t1 - t0) * 1e3))
a = timed(lambda: d_in.copy_(t_in, non_blocking=True)); b = timed(lambda: t_out.copy_(d_out, non_blocking=True))
print("registered in place: H2D %.1f ms (%.1f GB/s), D2H %.1f ms (%.1f GB/s)" % (a * 1e3, n_in / a / 1e9, b * 1e3, n_out / b / 1e9))
s2 = torch.cuda.Stream()
def both():
    d_in.copy_(t_in, non_blocking=True)
    with torch.cuda.stream(s2):
        t_out.copy_(d_out, non_blocking=True)
c = timed(both)
print("registered, both directions at once: %.1f ms" % (c * 1e3))
t0 = time.perf_counter(); rt.cudaHostUnregister(host_in.ctypes.data); rt.cudaHostUnregister(host_out.ctypes.data); t1 = time.perf_counter()
print("hipHostUnregister: %.1f ms" % ((t1 - t0) * 1e3))
p_in = torch.empty(n_in, dtype=torch.uint8, pin_memory=True)
t0 = time.perf_counter(); p_in.numpy()[:] = host_in; t1 = time.perf_counter()
print("CPU copy 512 MB pageable -> pinned staging (one thread): %.1f ms (%.1f GB/s)" % ((t1 - t0) * 1e3, n_in / (t1 - t0) / 1e9))
