"""(GPU box) Where does a short bench run spend its time?  Runs the 1 M dam-break from the start lattice in chunks
and prints, per chunk of frames: wall ms/step (no events), then the same chunk size again with every stage
bracketed (HIP events), the stage split, and the neighbour-list statistics.  Separates cold start (first chunk),
regime (list length vs frame) and event overhead.   python tools/regime_probe.py [nominal] [chunk] [chunks]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

import bench  # noqa: E402

nominal = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 5
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 12
pkg = bench.load_package()
scene, side = pkg.scene_dambreak(nominal, False)
n = len(scene["id"])
p = pkg.default_params(4, side)
s = pkg.Solver(h=0.1, flags=pkg.FLAG_STAGE_TIMING)
s.upload(**scene)
frame = 0
for c in range(chunks):
    timed = c % 2 == 0  # alternate: plain wall clock / all stages bracketed
    s.set_option("timing_mask", 0 if timed else 0xFFFFFFFF)
    s.reset_stage_times()
    s.sync()
    t0 = time.perf_counter()
    s.steps(p, chunk)
    s.sync()
    ms = 1e3 * (time.perf_counter() - t0) / chunk
    cnt = s.nbr_counts()
    row = {"frames": [frame, frame + chunk], "ms_per_step": round(ms, 4), "events": not timed,
           "mean_list": round(float(np.minimum(cnt, 64).mean()), 2), "max_list": int(np.minimum(cnt, 64).max()),
           "overflow": float((cnt == 0xFFFFFFFF).mean())}
    if not timed:
        st = s.stage_times()
        row["stage_ms_per_launch"] = {k: round(v[0], 4) for k, v in st.items()}
    print(json.dumps(row), flush=True)
    frame += chunk
# long tail: jump to frame 200 and measure there
s.set_option("timing_mask", 0)
s.steps(p, max(0, 200 - frame))
s.sync()
for rep in range(3):
    t0 = time.perf_counter()
    s.steps(p, 20)
    s.sync()
    print(json.dumps({"frames": "200+", "ms_per_step": round(1e3 * (time.perf_counter() - t0) / 20, 4)}), flush=True)
