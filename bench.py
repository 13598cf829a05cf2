#!/usr/bin/env python3
"""bench.py — particle-steps/s of the PBF-SPH hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P] [--fp64] [--fast-math]

A "step" is one advance() (predict+key, counting sort + cell table, diffuse, K_s = 4 x (lambda,
delta-p), finalise) over the device-resident particle state of the dam-break scene
(N = 1: 1 M nominal = 1 024 000 particles, BASELINE.json configs[2]).  Inputs are
resident in HBM when the timed region starts; the PCIe-inclusive advance() rate is reported
separately by the C++ benchmark CLI and in DESIGN.md, never as `value`.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — the dominant kernel's algorithmic bytes / its mean HIP-event duration
  cpu_baseline — the CPU oracle (kind "port", OpenMP) timed on a bounded sample of the same workload.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

# algorithmic bytes per particle for one launch of each stage (fp32; fp64 doubles
# the float fields, 4-byte keys/indices stay).
STAGE_BYTES_F32 = {"advect+zindex": 52, "sortz+gridtable": 132, "sph-diffuse": 32, "sph-lambda": 16, "sph-delta": 28,
                   "sph-finalise": 60}
STAGE_BYTES_F64 = {"advect+zindex": 100, "sortz+gridtable": 256, "sph-diffuse": 64, "sph-lambda": 32,
                   "sph-delta": 56, "sph-finalise": 120}
# the kernel (rocprofv3 name) behind each timing entry of the default configuration
KERNEL_OF = {"sph-lambda/list-build": "k_build_lists_q", "sph-delta": "k_gather_from_lists<DeltaOp>",
             "sph-finalise": "k_finalise", "advect+zindex": "k_predict"}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def candidates_per_particle(keys, table):
    """Mean number of candidates a particle's 27-cell walk visits (SURVEY.md §8d "C"), from the sorted Morton keys
    and the grid table of the last step: for every particle the populations of its 27 cells, with the reference's
    rules — a cell outside the table, and the table's last cell, are empty (sph.hpp:206-208)."""
    keys = np.asarray(keys, np.int64)
    table = np.asarray(table, np.int64)
    tn = len(table)
    inside = keys[keys < tn]
    cnt = np.bincount(inside, minlength=tn)
    cnt[-1] = 0
    mx, my, mz = 0x09249249, 0x12492492, 0x24924924

    def nb(m, unit):
        a = keys & m
        return [(a - unit) & m, a, ((a | (~m & 0x3FFFFFFF)) + unit) & m]

    xs, ys, zs = nb(mx, 1), nb(my, 2), nb(mz, 4)
    total = np.zeros(len(keys), np.int64)
    for x in xs:
        for y in ys:
            for z in zs:
                code = x | y | z
                total += np.where(code < tn, cnt[np.minimum(code, tn - 1)], 0)
    return float(total.mean()) if len(keys) else 0.0


def load_package():
    pkg_dir = os.path.join(ROOT, "pbf-sph_amd")
    spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["pbf_sph_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(state, side, fp64, iteration, budget_s=20.0, max_steps=8):
    """Time the oracle (the checker, kind 'port') on the host cores: a bounded sample of the SAME
    workload, started from the GPU's post-warm-up state.  Reported, never optimised against."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    threads = len(os.sched_getaffinity(0))
    o = O.Oracle(fp64)
    o.set_particles(**state)
    q = O.make_params(iteration=iteration, max_bound=(side,) * 3, mode=O.JACOBI, sort=O.SORT_STABLE, threads=threads)
    n = len(state["id"])
    t0 = time.perf_counter()
    o.step(q)  # untimed: first touch / OpenMP pool start
    first = time.perf_counter() - t0
    steps, t = 0, 0.0
    while steps < max_steps and (steps == 0 or t + t / steps < budget_s - first):
        a = time.perf_counter()
        o.step(q)
        t += time.perf_counter() - a
        steps += 1
    return {"value": n * steps / t, "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "sample": f"{steps} steps of the same {n}-particle dam-break state after warm-up "
                      f"(oracle Jacobi mode, OpenMP, -O2 no fast-math), {t:.1f} s",
            "ms_per_step": 1e3 * t / steps}, o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--particles", type=int, default=1 << 20, help="nominal particle count per GPU (dam-break)")
    ap.add_argument("--solver-iter", type=int, default=4)
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--fast-math", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lds", action="store_true", help="A/B: per-particle global-memory gather kernels")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dist = None
    backend = os.environ.get("PBF_BENCH_BACKEND", "nccl")  # "gloo": rehearsal with several ranks on ONE GPU
    if world == 1 or backend == "nccl":
        torch.cuda.set_device(local_rank)
    force_slab = os.environ.get("PBF_BENCH_FORCE_SLAB") == "1"  # rehearsal: slab driver + RCCL set-up with ONE rank
    if world > 1 or force_slab:
        import torch.distributed as dist
        if force_slab and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend)

    pkg = load_package()
    flags = (0 if os.environ.get("PBF_BENCH_NO_EVENTS") else pkg.FLAG_STAGE_TIMING) | (pkg.FLAG_FAST_MATH if args.fast_math else 0) | (pkg.FLAG_NO_LDS if args.no_lds else 0)
    scene, side = pkg.scene_dambreak(args.particles, args.fp64)
    n = len(scene["id"])
    p = pkg.default_params(args.solver_iter, side)
    drv = None
    if world == 1 and not force_slab:
        solver = pkg.Solver(h=0.1, fp64=args.fp64, device=local_rank, flags=flags)
        solver.upload(**scene)
        run = lambda k: solver.steps(p, k)  # noqa: E731
    else:
        # Weak scaling: `world` dam-break columns side by side along x, one per rank, in ONE box of
        # world*side x side x side; slabs of equal width, ghost-layer exchange over RCCL (slab.py).
        from pbf_sph_amd import slab
        # Odd ranks hold the MIRROR image of the column (x -> side - x inside their sub-box): every cut plane is
        # then a mirror plane of the whole set-up, i.e. dynamically a wall — each rank's problem is the N = 1
        # dam-break, the net flux through a cut is zero and the load stays balanced (up to chaotic symmetry breaking).
        x = scene["pos"][:, 0]
        scene["pos"][:, 0] = (x.dtype.type(side) - x if rank % 2 else x) + x.dtype.type(rank * side)
        scene["id"] += np.uint64(rank * n)
        p.max_bound[0] = world * side
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)  # RCCL ops and the solver's kernels are ordered on this one stream
        solver = pkg.Solver(h=0.1, fp64=args.fp64, device=local_rank, flags=flags, stream=stream.cuda_stream)
        cap = max(n // 2, 1 << 16)  # wire records per neighbour and phase
        solver._chk(solver.L.pbf_reserve(solver.ctx, 3 * n + 2 * cap), "pbf_reserve")  # head-room if the load drifts
        solver.upload(**scene)
        eng = slab.HipEngine(solver, torch, torch.device("cuda", local_rank))
        drv = slab.SlabSolver(eng, dist, rank, world, slab.even_cuts(world, world * side), cap,
                              stage_via_host=(backend != "nccl"))
        run = lambda k: drv.steps(p, k)  # noqa: E731

    def barrier():
        solver.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Warm-up; its last steps are bracketed stage by stage (HIP events on the solver's stream) to find the
    # dominant stage and the per-stage split.  The timed region then brackets ONLY the dominant stage:
    # 8 event records per step instead of 44 (the full set costs ~4 % at 1 M particles, ~10 % at 256 K).
    split_steps = min(10, args.warmup)
    solver.set_option("timing_mask", 0)
    run(args.warmup - split_steps)
    barrier()
    solver.set_option("timing_mask", 0xFFFFFFFF)
    solver.reset_stage_times()
    run(split_steps)
    barrier()
    split = solver.stage_times()
    names = list(split)
    # an entry "stage/part" is one kernel inside "stage": the composite stage is then not a kernel of its own
    composite = {k for k in split if any(o.startswith(k + "/") and split[o][1] > 0 for o in split)}
    kernels = [k for k in split if k not in composite]
    dom_name = max(kernels, key=lambda k: split[k][0] * split[k][1]) if split_steps else "sph-lambda"
    solver.set_option("timing_mask", 1 << names.index(dom_name) if dom_name in names else 0xFFFFFFFF)
    solver.reset_stage_times()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    n_final = solver.n
    if dist is not None:
        t = torch.tensor([elapsed, float(n_final)], device="cuda", dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        assert int(round(float(t[1].item()))) == n * world, "particles were lost or duplicated across slabs"
        imbalance = float(tmax[1].item()) / (n * 1.0)
    stage = solver.stage_times()

    if rank == 0:
        total_particles = n * world
        value = total_particles * args.steps / elapsed
        sb = STAGE_BYTES_F64 if args.fp64 else STAGE_BYTES_F32
        # dominant kernel = the kernel-level entry with the largest total time per step
        per_step = {k: ms * calls / max(split_steps, 1) for k, (ms, calls) in split.items()}  # last warm-up steps
        dom = dom_name
        dom_ms, dom_calls = stage[dom]  # the dominant stage: every launch of the timed region
        if dom == "sph-lambda/list-build":
            # k_build_lists_q: own pStar + quantised position of every particle once + key in,
            # list length + S list entries out per particle; S = mean length of the lists the last launch wrote
            cnt = solver.nbr_counts()
            mean_list = float(np.minimum(cnt, 64).mean())
            dom_bytes = (32 if args.fp64 else 16) + 8 + 4 + 4 + 4 * mean_list
        else:
            mean_list = None
            dom_bytes = sb[dom]
        achieved = dom_bytes * n / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        bytes_step = (276 + 44 * args.solver_iter) if not args.fp64 else (528 + 88 * args.solver_iter)
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
        # WRITE_SIZE collected separately, 2 x FETCH gfx950 correction; profiles/r01_pmc_traffic.json) — only
        # valid for the configuration those passes ran: 1 GPU, fp32, default particle count and math mode
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if world == 1 and not args.fp64 and args.particles == (1 << 20) and not args.fast_math and os.path.exists(tfile):
            tag = {"sph-lambda/list-build": "k_build_lists", "sph-lambda": "LambdaOp", "sph-delta": "DeltaOp",
                   "sph-diffuse": "DiffuseOp"}.get(dom)
            for name, t in json.load(open(tfile)).items():
                if tag and tag in name:
                    traffic = t["hbm_bytes_gfx950_corrected"]
        out = {
            "metric": "particle-steps/sec (1 M particles, 4 iters) + achieved HBM GB/s, 1/2/4/8 MI355X",
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",  # per-GPU work is fixed: one 1 M-particle column per rank
            "vs_baseline": None,
            "dtype": "f64" if args.fp64 else "f32",
            "data": "synthetic",
            "config": {"workload": f"dam-break, {n} particles/GPU (nominal {args.particles}), box {side:.0f}^3, "
                                   f"h=0.1, K={args.solver_iter}, dt=0.01245, scale=500",
                       "particles": total_particles, "solver_iter": args.solver_iter,
                       "math": "fast (v_rsq, fma)" if args.fast_math else "precise (IEEE div/sqrt, no contraction)",
                       "parallelism": "1 GPU, device-resident" if world == 1 else
                                      f"{world} x-slabs, one rank per GPU, 1-cell ghost layer refreshed after every "
                                      f"lambda/delta launch over RCCL ({backend}); {world} columns side by side, odd ones mirrored; "
                                      f"max rank load {imbalance:.2f}x mean"},
            "roofline": {"bound": "hbm", "kernel": KERNEL_OF.get(dom, dom), "stopwatch_entry": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_particle": dom_bytes, "mean_list_length": mean_list,
                         "mean_launch_ms": dom_ms, "launches_timed": dom_calls,
                         "whole_step_GBs": value * bytes_step / 1e9 / world,
                         "note": "the neighbour kernels are instruction-issue / LDS / latency bound, not HBM-bound (SURVEY.md §8d)"},
            # the step's pure streams for contrast ("where >= 50 % of 8 TB/s is physically meaningful")
            "roofline_streaming": {k: {"achieved": sb[k] * n / (split[k][0] * 1e-3) / 1e9, "unit": "GB/s",
                                       "frac": sb[k] * n / (split[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "algorithmic_bytes_per_particle": sb[k], "mean_launch_ms": split[k][0]}
                                   for k in ("sph-finalise", "advect+zindex") if k in split and split[k][0] > 0},
            # the whole lambda stage (quantise + list build + list-driven lambda) against SURVEY.md §8d's 16 B/particle
            "roofline_stage": {k: {"achieved": sb[k] * n / (split[k][0] * 1e-3) / 1e9, "unit": "GB/s",
                                   "frac": sb[k] * n / (split[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "algorithmic_bytes_per_particle": sb[k], "mean_stage_ms": split[k][0]}
                               for k in ("sph-lambda", "sph-delta") if k in split and split[k][0] > 0},
            "stage_ms_per_step": per_step,
            "stage_split_from": f"last {split_steps} warm-up steps (all stages bracketed); roofline: timed region",
        }
        if world == 1:
            # SURVEY.md §8(d)'s secondary figure: C candidates per particle are distance-tested once per solver
            # iteration (the list build); the S that survive get exact pair terms twice (lambda, delta-p)
            try:
                C = candidates_per_particle(solver.keys(), solver.table())
                rate = value * args.solver_iter
                out["pairs"] = {"candidates_per_particle": C, "candidate_tests_per_s": rate * C,
                                "pair_terms_per_s": (rate * 2 * mean_list) if mean_list is not None else None}
            except Exception as e:  # diagnostic only
                out["pairs"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:
            state = solver.download()
            cb, o = cpu_baseline(state, side, args.fp64, args.solver_iter)
            # the oracle is also the checker: one more GPU step from the same state must agree
            solver.step(p)
            g = solver.download()
            o2_state = None
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib as O
                # device_pow: the oracle evaluates pow(q,4) as (q*q)*(q*q) like the kernels => bit-exact expected
                oo = O.Oracle(args.fp64, device_pow=not args.fast_math)
                oo.set_particles(**state)
                oo.step(O.make_params(iteration=args.solver_iter, max_bound=(side,) * 3, mode=O.JACOBI,
                                      sort=O.SORT_STABLE, threads=cb["cores"]))
                o2_state = oo.get_particles()
            except Exception as e:  # the check is informative here; tests/ are the gate
                out["parity_check_error"] = str(e)
            if o2_state is not None:
                gi, wi = np.argsort(g["id"]), np.argsort(o2_state["id"])
                d = np.linalg.norm(g["pos"][gi].astype(np.float64) - o2_state["pos"][wi], axis=1)
                out["parity_check_max_dx_world_units"] = float(d.max())
                out["parity_check_bit_exact"] = bool(np.array_equal(g["pos"][gi], o2_state["pos"][wi]))
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = value / cb["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
