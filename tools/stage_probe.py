"""Time one gather stage repeatedly on a FROZEN state (frame F of the bench scene) under different
kernel variants (config = kind:0[:list_max]).  lambda only writes pstar.w, so re-running it is idempotent."""
import argparse, os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package
ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1 << 20)
ap.add_argument("--frame", type=int, default=150)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--stage", default="lambda")
ap.add_argument("--configs", default="global:0,lists:0:16,bricks:0")
args = ap.parse_args()
pkg = load_package()
sc, side = pkg.scene_dambreak(args.particles, False)
p = pkg.default_params(4, side)
res = {}
for fast in (0, 1):
    base = pkg.Solver(h=0.1, flags=(pkg.FLAG_FAST_MATH if fast else 0) | pkg.FLAG_NO_LDS)
    base.upload(**sc)
    base.steps(p, args.frame)
    st = base.download()
    for cfg in args.configs.split(","):
        kind, probe, *rest = cfg.split(":")
        flags = (pkg.FLAG_FAST_MATH if fast else 0) | (pkg.FLAG_NO_LDS if kind == "global" else 0)
        s = pkg.Solver(h=0.1, flags=flags)
        s.set_option("gather", {"global": 0, "lists": 1, "bricks": 2}[kind])
        if rest and kind == "global":
            s.set_option("pad_lds", int(rest[0]))
        elif rest:
            s.set_option("list_max", int(rest[0]))
        s.upload(**st)
        s.stage("predict", p).stage("sort", p)
        s.set_option("diag", int(probe))
        for _ in range(3):
            s.stage(args.stage, p)
        s.sync()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            s.stage(args.stage, p)
        s.sync()
        res[f"fast{fast}:{cfg}"] = round((time.perf_counter() - t0) / args.reps * 1e3, 4)
print(json.dumps(res))
