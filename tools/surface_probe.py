"""(GPU box) where does pbf_surface + the mesh hand-over spend its time at 1 M particles, settled (frame 200)?"""
import ctypes as C, importlib.util, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(ROOT, "pbf-sph_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "pbf-sph_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["pbf_sph_amd"] = pkg; spec.loader.exec_module(pkg)
from pbf_sph_amd import capi
nominal = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
sc, side = pkg.scene_dambreak(nominal, False)
s = pkg.Solver(h=0.1); s.upload(**sc); p = pkg.default_params(4, side)
s.steps(p, 200); s.sync()
mc = capi.McParams()
out = {}
for rep in range(3):
    s.step(p); s.sync()
    nt = C.c_uint64()
    t0 = time.perf_counter(); s._chk(s.L.pbf_surface(s.ctx, C.byref(p), C.byref(mc), C.byref(nt)), "surface"); s.sync(); t1 = time.perf_counter()
    a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
    s._chk(s.L.pbf_map_mesh(s.ctx, C.byref(a), C.byref(b), C.byref(c)), "map"); t2 = time.perf_counter()
    n = nt.value
    vs = np.empty((3 * n, 3), np.float32)
    t3 = time.perf_counter(); C.memmove(vs.ctypes.data, a.value, vs.nbytes); t4 = time.perf_counter()
    out = dict(triangles=n, surface_kernels_ms=1e3 * (t1 - t0), map_mesh_dma_ms=1e3 * (t2 - t1), memmove_vs_ms=1e3 * (t4 - t3),
               memmove_GBs=vs.nbytes / (t4 - t3) / 1e9, first_touch_included=True)
    t5 = time.perf_counter(); C.memmove(vs.ctypes.data, a.value, vs.nbytes); t6 = time.perf_counter()
    out["memmove_vs_warm_ms"] = 1e3 * (t6 - t5)
print(json.dumps(out))
