"""How many candidate slots a 64-lane wave executes in the list build vs. how many candidates its lanes
have: per (dy,dz) row / per z plane / per particle flattening.  Diagnostic tool (GPU needed)."""
import argparse, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1 << 20)
ap.add_argument("--frame", type=int, default=150)
args = ap.parse_args()
pkg = load_package()
sc, side = pkg.scene_dambreak(args.particles, False)
s = pkg.Solver(h=0.1)
s.upload(**sc)
p = pkg.default_params(4, side)
s.steps(p, args.frame)
s.stage("predict", p).stage("sort", p)
keys = s.keys().astype(np.int64); table = s.table().astype(np.int64)
tn = len(table); n = len(keys)
cnt = np.diff(np.concatenate([table, [np.searchsorted(keys, tn)]]))
cnt[-1] = 0  # the table's last cell is empty by the reference's rule
MX, MY, MZ = 0x09249249, 0x12492492, 0x24924924
def nb(k, m, unit):
    a = k & m
    return [(a - unit) & m, a, ((a | (~m & 0x3FFFFFFF)) + unit) & m]
xs, ys, zs = nb(keys, MX, 1), nb(keys, MY, 2), nb(keys, MZ, 4)
L = np.zeros((n, 9), np.int64)
for dz in range(3):
    for dy in range(3):
        for dx in range(3):
            code = xs[dx] | ys[dy] | zs[dz]
            ok = code < tn
            L[:, dz * 3 + dy] += np.where(ok, cnt[np.minimum(code, tn - 1)], 0)
nw = n // 64
Lw = L[:nw * 64].reshape(nw, 64, 9)
tot = Lw.sum(2)
r4 = lambda a: (a + 3) // 4 * 4
out = {
    "mean_candidates_per_particle": float(tot.mean()),
    "slots_per_lane_row_flatten": float(r4(Lw.max(1) + 1).sum(1).mean()),
    "slots_per_lane_plane_flatten": float(r4(Lw.reshape(nw, 64, 3, 3).sum(3).max(1) + 3).sum(1).mean()),
    "slots_per_lane_full_flatten": float(r4(tot.max(1) + 9).mean()),
    "distinct_cells_per_wave": float(np.mean([len(np.unique(keys[w * 64:(w + 1) * 64])) for w in range(0, nw, 37)])),
    "row_L_mean": float(Lw.mean()), "row_L_wavemax_mean": float(Lw.max(1).mean()),
}
print(json.dumps(out, indent=1))
