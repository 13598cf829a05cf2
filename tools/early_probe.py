"""(GPU box) list lengths / overflow share frame by frame from the start lattice (why the first frames are slow)."""
import importlib.util, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(ROOT, "pbf-sph_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "pbf-sph_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["pbf_sph_amd"] = pkg; spec.loader.exec_module(pkg)
nominal = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
sc, side = pkg.scene_dambreak(nominal, False)
s = pkg.Solver(h=0.1); s.upload(**sc); p = pkg.default_params(4, side)
for f in range(0, 41):
    s.step(p)
    if f in (0, 1, 2, 4, 8, 12, 16, 20, 30, 40):
        c = s.nbr_counts().astype(np.int64); ov = c == 0xFFFFFFFF; c = np.where(ov, 97, c)
        print(json.dumps({"frame": f, "mean": round(float(c[~ov].mean()), 1), "p50": float(np.percentile(c, 50)), "p90": float(np.percentile(c, 90)),
                          "gt40": round(float((c > 40).mean()), 4), "overflow": round(float(ov.mean()), 5)}))
