"""(GPU box) neighbour-list length distribution after the k-th solver iteration's build of a settled step (k = 1..4): is one
iteration's list population different (the second delta-p launch of a step takes 1.7 x the others')?
  python tools/iter_probe.py [nominal=1048576] [frames=205]"""
import importlib.util, json, os, sys, copy
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(ROOT, "pbf-sph_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "pbf-sph_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["pbf_sph_amd"] = pkg; spec.loader.exec_module(pkg)
nominal = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 205
sc, side = pkg.scene_dambreak(nominal, False)
out = []
for k in (1, 2, 3, 4):
    s = pkg.Solver(h=0.1); s.upload(**sc); p = pkg.default_params(4, side)
    s.steps(p, frames)
    s.stage("predict", p).stage("sort", p).stage("diffuse", p)
    for it in range(k):
        s.stage("lambda", p)
        if it < k - 1:
            s.stage("delta", p)
    c = s.nbr_counts().astype(np.int64); ov = c == 0xFFFFFFFF; c = np.where(ov, 65, c)
    lam = s.pstar()[:, 3]
    out.append({"iteration": k, "mean": float(c.mean()), "p50": float(np.percentile(c, 50)), "p99": float(np.percentile(c, 99)),
                "max": int(c.max()), "gt40": float((c > 40).mean()), "gt56": float((c > 56).mean()), "overflow": float(ov.mean()),
                "wave_max_mean": float(c[: len(c) // 64 * 64].reshape(-1, 64).max(1).mean()),
                "lambda_zero": float((lam == 0).mean()), "lambda_abs_p50": float(np.percentile(np.abs(lam), 50)),
                "lambda_tiny": float((np.abs(lam) < 1e-30).mean())})
print(json.dumps(out, indent=1))
