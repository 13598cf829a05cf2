"""(GPU box) neighbour-list length distribution of the settled dam-break (frame 200+): what a two-tier nbrList must hold.
  python tools/list_hist.py [nominal=1048576] [frames=210]"""
import importlib.util, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(ROOT, "pbf-sph_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "pbf-sph_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["pbf_sph_amd"] = pkg; spec.loader.exec_module(pkg)
nominal = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 210
sc, side = pkg.scene_dambreak(nominal, False)
s = pkg.Solver(h=0.1); s.upload(**sc); p = pkg.default_params(4, side)
out = []
for upto in (30, frames):
    s.steps(p, upto - (out[-1]["frame"] if out else 0))
    c = s.nbr_counts().astype(np.int64); ov = c == 0xFFFFFFFF; c = np.where(ov, 65, c)
    blockmax = c[: len(c) // 256 * 256].reshape(-1, 256).max(1)
    out.append({"frame": upto, "n": len(c), "mean": float(c.mean()), "overflow64_frac": float(ov.mean()),
                "pct": {str(q): float(np.percentile(c, q)) for q in (50, 90, 99, 99.9)},
                "frac_gt": {str(t): float((c > t).mean()) for t in (32, 36, 40, 44, 48, 56)},
                "block_max_pct": {str(q): float(np.percentile(blockmax, q)) for q in (10, 50, 90)},
                "hist": np.bincount(np.minimum(c, 65), minlength=66).tolist()})
print(json.dumps(out))
