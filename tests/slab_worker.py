"""Worker of the slab tests: one rank of an N-rank run of pbf-sph_amd/slab.py.

  engine "oracle": CPU engine (tests/slab_engines.py), gloo, CPU tensors        — runs anywhere
  engine "hip"   : the product kernels on cuda:0 driven by slab.py's Python protocol, gloo with host-staged buffers
  engine "hipc"  : the product path: pbf_slab_step inside libpbf_hip.so (C ABI) with the host-callback transport
                   over gloo                                                      — several ranks share ONE GPU
Writes the rank's final owned particles to <out>/rank<r>.npz."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from conftest import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", default="oracle")
    ap.add_argument("--scene", default="cubes2048")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--iteration", type=int, default=4)
    ap.add_argument("--cuts", default="even")
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--rebalance", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0, help="records in the first message of an assembly round (hipc)")
    ap.add_argument("--xsph", type=int, default=0)
    ap.add_argument("--vorticity", type=int, default=0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_package()
    from pbf_sph_amd import slab

    if a.scene.startswith("cubes"):
        sc, side = pkg.scene_cubes(int(a.scene[5:]), a.fp64), 1000.0
    else:
        sc, side = pkg.scene_dambreak(int(a.scene[3:]), a.fp64)
    p = pkg.default_params(a.iteration, side)
    p.xsph, p.vorticity = a.xsph, a.vorticity
    if a.cuts == "even":
        cuts = slab.even_cuts(world, side)
    elif a.cuts.startswith("x:"):  # explicit world-space cut positions
        cuts = [0] + [slab.column_of(float(v)) for v in a.cuts[2:].split(",")] + [1024]
    else:
        cuts = slab.balanced_cuts(world, sc["pos"][:, 0], side)
    col = ((sc["pos"][:, 0].astype(np.float64) / 500.0 + 0.2) / 0.1).astype(np.int64)
    mine = (col >= cuts[rank]) & (col < cuts[rank + 1])
    part = {k: v[mine] for k, v in sc.items()}
    cap = len(sc["id"])
    if a.engine == "oracle":
        from slab_engines import OracleEngine
        eng = OracleEngine(a.fp64, device_pow=True)
        eng.upload(**part)
        get = eng.download
        stage = False
    else:
        s = pkg.Solver(h=0.1, fp64=a.fp64, device=0)
        s._chk(s.L.pbf_reserve(s.ctx, cap), "pbf_reserve")
        s.upload(**part)
        eng = slab.HipEngine(s, torch, torch.device("cuda", 0))
        get = s.download
        stage = True
    if a.engine == "hipc":
        drv = slab.CSlabSolver(s, dist, torch, rank, world, cuts, a.chunk or cap, a.chunk or cap, transport="gloo-host",
                               rebalance_every=a.rebalance)
        drv.steps(p, a.steps)
        stats = dict(migrated=-1, ghosts=-1, exchanges=drv.rounds, recuts=drv.stats["recuts"])
        cuts = drv.cuts
    else:
        drv = slab.SlabSolver(eng, dist, rank, world, cuts, cap, stage_via_host=stage, rebalance_every=a.rebalance)
        drv.steps(p, a.steps)
        stats = dict(migrated=drv.stats["migrated"], ghosts=drv.stats["ghosts"], exchanges=drv.stats["exchanges"],
                     recuts=drv.stats["recuts"])
        cuts = drv.cuts
    out = get()
    os.makedirs(a.out, exist_ok=True)
    np.savez(os.path.join(a.out, f"rank{rank}.npz"), cuts=np.array(cuts), **stats, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
