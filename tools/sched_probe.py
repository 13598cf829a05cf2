"""(GPU box) how unevenly the 64 lanes of a wave are loaded in the list build / the list-driven delta-p on the settled
dam-break, and what a per-workgroup schedule (lanes take the particles of their 256 / 1024-block in order of list
length) would recover.  Row-slot order (the default working set).
  python tools/sched_probe.py [nominal=1048576] [frames=205]"""
import importlib.util, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(ROOT, "pbf-sph_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "pbf-sph_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["pbf_sph_amd"] = pkg; spec.loader.exec_module(pkg)
nominal = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 205
sc, side = pkg.scene_dambreak(nominal, False)
s = pkg.Solver(h=0.1); s.upload(**sc); p = pkg.default_params(4, side)
s.steps(p, frames)
cnt = s.nbr_counts().astype(np.int64); cnt = np.where(cnt == 0xFFFFFFFF, 65, cnt)   # row-slot order
keys = s.keys().astype(np.int64)                                                        # Morton order
n = len(cnt)

def compact(k):  # every third bit
    k = k & 0x09249249
    k = (k | k >> 2) & 0x030C30C3; k = (k | k >> 4) & 0x0300F00F; k = (k | k >> 8) & 0x030000FF; k = (k | k >> 16) & 0x3FF
    return k
x, y, z = compact(keys), compact(keys >> 1), compact(keys >> 2)
P = 1 << int(np.ceil(np.log2(max(x.max(), y.max(), z.max()) + 2)))
lin = (z * P + y) * P + x
order = np.argsort(lin, kind="stable"); lin = lin[order]; x, y, z = x[order], y[order], z[order]
occ = np.bincount(lin, minlength=P ** 3 + 2 * P * P).astype(np.int64)
cs = np.concatenate([[0], np.cumsum(occ)])
L = np.zeros((n, 9), np.int64)
for r in range(9):
    yy, zz = y + r % 3 - 1, z + r // 3 - 1
    ok = (yy >= 0) & (yy < P) & (zz >= 0) & (zz < P)
    base = (np.clip(zz, 0, P - 1) * P + np.clip(yy, 0, P - 1)) * P
    L[:, r] = np.where(ok, cs[base + np.minimum(x + 2, P)] - cs[base + np.maximum(x - 1, 0)], 0)

def cost(perm_block):
    """perm_block: None (identity) or the block size inside which lanes take particles sorted by list length"""
    idx = np.arange(n)
    if perm_block:
        nb = n // perm_block * perm_block
        blk = cnt[:nb].reshape(-1, perm_block)
        o = np.argsort(blk, axis=1, kind="stable") + (np.arange(nb // perm_block) * perm_block)[:, None]
        idx = np.concatenate([o.ravel(), np.arange(nb, n)])
    nw = n // 64
    c = cnt[idx][: nw * 64].reshape(nw, 64); l = L[idx][: nw * 64].reshape(nw, 64, 9)
    trips = ((l.max(1) + 7) // 8).sum(1)            # test trips of 8 candidates: per row, the longest lane's
    flush = (c.max(1) + 3) // 4 * 4                 # drained slots per lane: the longest lane's
    return {"test_slots_per_lane": float(trips.mean() * 8), "candidates_per_lane": float(l.sum(2).mean()),
            "drain_slots_per_lane": float(flush.mean()), "hits_per_lane": float(c.mean()),
            "test_efficiency": float(l.sum(2).mean() / (trips.mean() * 8)), "drain_efficiency": float(c.mean() / flush.mean())}

out = {"n": n, "frame": frames, "identity": cost(None), "sorted_in_256": cost(256), "sorted_in_1024": cost(1024),
       "sorted_in_4096": cost(4096)}
# a schedule made from the PREVIOUS step's lengths (what the build of iteration 1 would have): lengths one step later
s.steps(p, 1)
cnt2 = s.nbr_counts().astype(np.int64)
out["note"] = "sorted_* use this iteration's own lengths (an upper bound on what a schedule from the previous iteration gives)"
print(json.dumps(out, indent=1))
