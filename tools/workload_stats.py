"""Workload statistics of the bench scene at a given frame: particles per cell, 27-cell candidates
per particle, neighbours within h, records per brick halo.  Diagnostic tool (GPU needed)."""
import argparse, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1 << 20)
ap.add_argument("--frames", type=str, default="0,20,60,150,250")
args = ap.parse_args()
pkg = load_package()
sc, side = pkg.scene_dambreak(args.particles, False)
s = pkg.Solver(h=0.1)
s.upload(**sc)
p = pkg.default_params(4, side)
done = 0
for f in [int(x) for x in args.frames.split(",")]:
    if f > done:
        s.steps(p, f - done); done = f
    s.stage("predict", p).stage("sort", p)
    keys = s.keys().astype(np.int64); table = s.table().astype(np.int64)
    ps = s.pstar()[:, :3]
    tn = len(table)
    cnt = np.diff(np.concatenate([table, [np.searchsorted(keys, tn)]]))
    occ = cnt[cnt > 0]
    # decode
    def compact(v):
        v = v & 0x09249249; v = (v | (v >> 2)) & 0x030C30C3; v = (v | (v >> 4)) & 0x0300F00F
        v = (v | (v >> 8)) & 0x030000FF; v = (v | (v >> 16)) & 0x3FF; return v
    ext = int(s.extent()[0][0])
    grid = np.zeros((ext + 2, ext + 2, ext + 2), np.int64)
    codes = np.flatnonzero(cnt > 0)
    x, y, z = compact(codes), compact(codes >> 1), compact(codes >> 2)
    ok = (x <= ext) & (y <= ext) & (z <= ext)
    grid[x[ok], y[ok], z[ok]] = cnt[codes[ok]]
    # 27-cell sums
    pad = np.pad(grid, 1)
    s27 = sum(pad[1 + dx:pad.shape[0] - 1 + dx, 1 + dy:pad.shape[1] - 1 + dy, 1 + dz:pad.shape[2] - 1 + dz]
              for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1))
    w = grid.ravel(); c27 = s27.ravel()
    mean_c = (w * c27).sum() / w.sum()
    # candidates per particle distribution
    per_cell = c27[w > 0]; weights = w[w > 0]
    order = np.argsort(per_cell); cw = np.cumsum(weights[order]) / weights.sum()
    q = lambda f_: per_cell[order][np.searchsorted(cw, f_)]
    # brick halos for several Morton-contiguous brick shapes
    shapes = {}
    for (BX, BY, BZ) in ((2, 2, 2), (4, 2, 2), (4, 4, 2), (4, 4, 4)):
        nbx, nby, nbz = (ext + 4) // BX + 1, (ext + 4) // BY + 1, (ext + 4) // BZ + 1
        g2 = np.zeros((nbx * BX + 2, nby * BY + 2, nbz * BZ + 2), np.int64)
        g2[1:1 + grid.shape[0], 1:1 + grid.shape[1], 1:1 + grid.shape[2]] = grid
        cs = np.pad(g2.cumsum(0).cumsum(1).cumsum(2), ((1, 0), (1, 0), (1, 0)))
        X, Y, Z = np.meshgrid(np.arange(nbx), np.arange(nby), np.arange(nbz), indexing="ij")
        def box(x0, x1, y0, y1, z0, z1):
            return (cs[x1, y1, z1] - cs[x0, y1, z1] - cs[x1, y0, z1] - cs[x1, y1, z0] + cs[x0, y0, z1] + cs[x0, y1, z0] + cs[x1, y0, z0] - cs[x0, y0, z0])
        hm = box(X * BX + 1, X * BX + 1 + BX, Y * BY + 1, Y * BY + 1 + BY, Z * BZ + 1, Z * BZ + 1 + BZ)
        hl = box(X * BX, X * BX + BX + 2, Y * BY, Y * BY + BY + 2, Z * BZ, Z * BZ + BZ + 2)
        m = hm > 0
        homes, halos = hm[m], hl[m]
        shapes[f"{BX}x{BY}x{BZ}"] = dict(bricks=int(m.sum()), home_mean=round(float(homes.mean()), 1), home_max=int(homes.max()),
            halo_mean=round(float(halos.mean()), 1), halo_p90=int(np.percentile(halos, 90)), halo_p99=int(np.percentile(halos, 99)),
            halo_max=int(halos.max()), staged_over_n=round(float(halos.sum()) / float(w.sum()), 2),
            lanes64=round(float(homes.sum()) / float((np.ceil(homes / 64) * 64).sum()), 3))
    # neighbours within h (sample)
    from scipy.spatial import cKDTree
    t = cKDTree(ps)
    idx = np.random.default_rng(0).choice(len(ps), 20000, replace=False)
    nn = np.array([len(v) for v in t.query_ball_point(ps[idx], 0.1)])
    print(json.dumps(dict(frame=f, occupied_cells=int(len(occ)), per_cell_mean=float(occ.mean()), per_cell_max=int(occ.max()),
          cand_mean=float(mean_c), cand_p50=int(q(.5)), cand_p99=int(q(.99)), cand_max=int(per_cell.max()),
          within_h_mean=float(nn.mean()), within_h_p99=int(np.percentile(nn, 99)), within_h_max=int(nn.max()),
          shapes=shapes)))
