"""(GPU box) where does an advance() frame go?  upload_aos / step / download_aos timed separately at 1 M particles."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
pkg = bench.load_package()
from pbf_sph_amd import capi
sc, side = pkg.scene_dambreak(1 << 20, False)
n = len(sc["id"])
dt = np.dtype([("id", "<u8"), ("type", "u1"), ("_pad", "u1", 3), ("mass", "<f4"), ("pos", "<f4", 3), ("vel", "<f4", 3), ("colour", "<f4", 4)])
a = np.zeros(n, dt)
for k in ("id", "type", "mass", "pos", "vel", "colour"):
    a[k] = sc[k]
lay = capi.AosLayout(56, 0, 8, 12, 16, 28, 40)
s = pkg.Solver(h=0.1)
p = pkg.default_params(4, side)
L = s.L
ptr = a.ctypes.data_as(C.c_void_p)
for rep in range(60):
    t0 = time.perf_counter(); L.pbf_upload_aos(s.ctx, n, ptr, C.byref(lay)); t1 = time.perf_counter()
    L.pbf_step(s.ctx, C.byref(p)); L.pbf_sync(s.ctx); t2 = time.perf_counter()
    L.pbf_download_aos(s.ctx, ptr, C.byref(lay)); t3 = time.perf_counter()
    if rep in (0, 1, 20, 40, 59):
        print(f"frame {rep}: upload {1e3*(t1-t0):.3f} ms  step {1e3*(t2-t1):.3f} ms  download {1e3*(t3-t2):.3f} ms  ({n*56/1e6:.1f} MB each way)", flush=True)
