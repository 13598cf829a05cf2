"""(GPU box) How far does pStar move between the solver iterations of one step?  Decides whether a neighbour list built
once per step with a skin could serve all K iterations.  Runs the 1 M dam-break to the settled regime, then walks
single frames stage by stage and prints, per iteration k, the distribution of |pStar_k - pStar_0| / h over particles.
  python tools/skin_probe.py [nominal] [settle] [frames]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

import bench  # noqa: E402

nominal = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 200
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pkg = bench.load_package()
scene, side = pkg.scene_dambreak(nominal, False)
K = 4
p = pkg.default_params(K, side)
s = pkg.Solver(h=0.1)
s.upload(**scene)
done = 0
for target in [settle + 40 * j for j in range(frames)]:
    s.steps(p, target - done)
    s.sync()
    done = target
    s.stage("predict", p).stage("sort", p).stage("diffuse", p)
    p0 = s.pstar()[:, :3].astype(np.float64)
    ext = s.extent()
    row = {"frame": done, "iters": []}
    prev = p0
    for k in range(K):
        s.stage("lambda", p).stage("delta", p)
        pk = s.pstar()[:, :3].astype(np.float64)
        d0 = np.sqrt(((pk - p0) ** 2).sum(1))
        dk = np.sqrt(((pk - prev) ** 2).sum(1))
        prev = pk
        row["iters"].append({
            "k": k + 1,
            "since_build": {"mean": float(d0.mean()), "p99": float(np.quantile(d0, 0.99)), "p9999": float(np.quantile(d0, 0.9999)),
                            "max": float(d0.max())},
            "frac_over": {str(t): float((d0 > t).mean()) for t in (0.0025, 0.005, 0.01, 0.025, 0.05, 0.1)},
            "this_iter_max": float(dk.max()),
        })
    s.stage("finalise", p)
    done += 1
    row["extent"] = [int(x) for x in ext[0]]
    print(json.dumps(row), flush=True)
print(json.dumps({"note": "distances in solver units (position / scale), h = 0.1: 0.005 = 0.05 h"}))
