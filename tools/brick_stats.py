"""(GPU box) Occupancy statistics of Morton-aligned bricks in the settled 1 M dam-break: how many bricks of 8 / 64 codes
are non-empty, how many particles they hold and how many records their +-1 halo holds (what an LDS tile must take).
  python tools/brick_stats.py [nominal] [frames]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

import bench  # noqa: E402

nominal = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pkg = bench.load_package()
scene, side = pkg.scene_dambreak(nominal, False)
p = pkg.default_params(4, side)
s = pkg.Solver(h=0.1)
s.upload(**scene)
s.steps(p, frames)
s.sync()
s.stage("predict", p).stage("sort", p)
keys = s.keys().astype(np.int64)
table = s.table().astype(np.int64)
tableN = len(table) - 2 if len(table) > 2 else len(table)
ext, _ = s.extent()


def compact(v):
    v = v & 0x09249249
    v = (v | (v >> 2)) & 0x030C30C3
    v = (v | (v >> 4)) & 0x0300F00F
    v = (v | (v >> 8)) & 0x030000FF
    v = (v | (v >> 16)) & 0x000003FF
    return v


def spread(v):
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


keys = keys[keys < tableN]
cx, cy, cz = compact(keys), compact(keys >> 1), compact(keys >> 2)
dim = 1024
grid = np.zeros((dim // 8, dim // 8, dim // 8), np.int64) if False else None
# dense count grid over the occupied bounding box
mx, my, mz = int(cx.max()) + 3, int(cy.max()) + 3, int(cz.max()) + 3
cnt = np.zeros((mz + 2, my + 2, mx + 2), np.int64)
np.add.at(cnt, (cz + 1, cy + 1, cx + 1), 1)   # shifted by 1: index 0 is the -1 layer
out = {"particles": int(len(keys)), "extent": [int(e) for e in ext], "cells_occupied": int((cnt > 0).sum()),
       "mean_per_occupied_cell": float(len(keys) / (cnt > 0).sum())}
for name, (bx, by, bz) in {"2x2x2": (2, 2, 2), "4x4x2": (4, 4, 2), "4x4x4": (4, 4, 4)}.items():
    nx, ny, nz = (mx + bx) // bx, (my + by) // by, (mz + bz) // bz
    home = np.zeros((nz, ny, nx), np.int64)
    halo = np.zeros((nz, ny, nx), np.int64)
    pad = np.zeros((nz * bz + 2, ny * by + 2, nx * bx + 2), np.int64)
    pad[:cnt.shape[0], :cnt.shape[1], :cnt.shape[2]] = cnt[:pad.shape[0], :pad.shape[1], :pad.shape[2]]
    csum = pad.cumsum(0).cumsum(1).cumsum(2)
    csum = np.pad(csum, ((1, 0), (1, 0), (1, 0)))

    def box(z0, z1, y0, y1, x0, x1):
        return (csum[z1, y1, x1] - csum[z0, y1, x1] - csum[z1, y0, x1] - csum[z1, y1, x0]
                + csum[z0, y0, x1] + csum[z0, y1, x0] + csum[z1, y0, x0] - csum[z0, y0, x0])
    zs, ys, xs = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    # pad index = cell + 1; home cells [b*bs, b*bs+bs) -> pad [b*bs+1, b*bs+bs+1); halo -> pad [b*bs, b*bs+bs+2)
    home = box(zs * bz + 1, zs * bz + bz + 1, ys * by + 1, ys * by + by + 1, xs * bx + 1, xs * bx + bx + 1)
    halo = box(zs * bz, zs * bz + bz + 2, ys * by, ys * by + by + 2, xs * bx, xs * bx + bx + 2)
    act = home > 0
    h, t = home[act], halo[act]
    q = lambda a, f: float(np.quantile(a, f))
    out[name] = {"active_bricks": int(act.sum()), "particles_per_brick": {"mean": float(h.mean()), "p10": q(h, .1), "p50": q(h, .5), "p90": q(h, .9), "max": int(h.max())},
                 "halo_records": {"mean": float(t.mean()), "p50": q(t, .5), "p90": q(t, .9), "p99": q(t, .99), "p999": q(t, .999), "max": int(t.max())},
                 "share_of_particles_in_bricks_with_halo_over": {str(c): float(h[t > c].sum() / h.sum()) for c in (448, 512, 576, 640, 768, 1536, 2048)},
                 "lane_fill_64": float(h.sum() / (np.ceil(h / 64) * 64).sum()), "lane_fill_512": float(h.sum() / (np.ceil(h / 512) * 512).sum())}
print(json.dumps(out, indent=1))
