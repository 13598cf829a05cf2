"""Per-iteration displacement statistics |delta p| / h in the bench scene (diagnostic)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package
pkg = load_package()
sc, side = pkg.scene_dambreak(1 << 20, False)
s = pkg.Solver(h=0.1)
s.upload(**sc)
p = pkg.default_params(4, side)
done = 0
for f in (60, 120, 200, 250):
    s.steps(p, f - done); done = f
    s.stage("predict", p).stage("sort", p).stage("diffuse", p)
    p0 = s.pstar()[:, :3].copy()
    prev = p0
    out = []
    for it in range(4):
        s.stage("lambda", p).stage("delta", p)
        cur = s.pstar()[:, :3]
        d = np.linalg.norm(cur - prev, axis=1) / 0.1
        dc = np.linalg.norm(cur - p0, axis=1) / 0.1
        out.append(dict(it=it, step_max=float(d.max()), step_p999=float(np.percentile(d, 99.9)), step_p50=float(np.median(d)),
                        cum_max=float(dc.max()), cum_p999=float(np.percentile(dc, 99.9))))
        prev = cur.copy()
    s.stage("finalise", p); done += 1
    print(json.dumps(dict(frame=f, iters=out)))
