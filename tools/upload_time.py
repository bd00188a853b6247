import os, sys, time
sys.path.insert(0, os.getcwd())
import pathtracer_amd as pt
for name in ("cornell_teapot", "sr2", "cubesplosion", "cornell_teapot"):
    sc = pt.LoadedScene("scenes/%s.obj" % name)
    t0 = time.perf_counter(); dev = sc.upload(); t1 = time.perf_counter()
    print(name, "upload %.3f s" % (t1 - t0), flush=True)
    dev.close()
