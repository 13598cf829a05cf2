#!/usr/bin/env python3
"""bench.py — particle-steps/s of the PBF-SPH hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P] [--fp64] [--fast-math]
                  [--scaling weak|strong] [--settle-to F]

A "step" is one advance() (predict+key, counting sort + cell table, diffuse, K_s = 4 x (lambda,
delta-p), finalise) over the device-resident particle state of the dam-break scene
(N = 1: 1 M nominal = 1 024 000 particles, BASELINE.json configs[2]).  Inputs are
resident in HBM when the timed region starts; the PCIe-inclusive advance() rate is reported
separately by the C++ benchmark CLI and in DESIGN.md, never as `value`.

Frames of one run (nothing is measured in the first frames of the process — code-object load, cold clocks and
the 1.84x over-dense start lattice's blow-up are not the workload):
    settle   max(0, F - W) untimed frames, F = --settle-to, default 200 = the reference CLI's own warm-up
             (args.hpp:36, benchmark.cpp:31-54: warm-up loop, then timed loop) — flag-independent regime
    warm-up  W untimed frames (their last ones are bracketed stage by stage to find the dominant kernel)
    timed    EXACTLY K frames = simulation frames [max(F, W), max(F, W) + K), barrier + synchronize on both sides
    split    10 more untimed frames with every stage bracketed by HIP events: the per-stage split

With --gpus N > 1 and no WORLD_SIZE in the environment this script starts its own N ranks (fresh processes,
one per GPU, before anything touches the GPU); under torchrun it uses the ranks it is given.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — the dominant kernel's algorithmic bytes / its mean HIP-event duration (solver stream)
  cpu_baseline — the CPU oracle (kind "port", OpenMP) timed on a bounded sample of the same workload.
"""
import argparse
import importlib.util
import json
import os
import socket
import subprocess
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
# (sph-lambda: the list build with lambda riding on it, on the row-major copy — the default since round 3; with option
# row_major = 0 the same launch is k_build_lists_op)
KERNEL_OF = {"sph-lambda/list-build": "k_build_lists_q", "sph-lambda": "k_build_rows_op", "sph-delta": "k_gather_from_lists<DeltaOp>",
             "sph-lambda/gather": "k_gather_from_lists<LambdaOp>",
             "sph-finalise": "k_finalise", "advect+zindex": "k_predict"}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
SPLIT_STEPS = 10
PROBE_STEPS = 10
# committed counter summaries of THIS command line (tools/r03_profiles.sh); the newest round present is quoted, by name
TRAFFIC_FILES = [os.path.join("profiles", f"r0{r}_pmc_traffic.json") for r in (3, 2)]
LIMITER_FILES = [os.path.join("profiles", f"r0{r}_limiter.json") for r in (3,)]


def first_existing(paths):
    for f in paths:
        if os.path.exists(os.path.join(ROOT, f)):
            return f
    return None


def match_kernel(table, kname):
    """entry of a {rocprofv3 kernel name: ...} table for a KERNEL_OF name like 'k_gather_from_lists<DeltaOp>'"""
    base, toks = kname.split("<")[0], kname.replace(">", "").split("<")[1:]
    for name, v in table.items():
        if base in name and all(t in name for t in toks):
            return v
    return None


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


def frame_plan(steps, warmup, settle_to):
    """(settle, first timed frame): the timed region covers simulation frames [first, first + steps)."""
    settle = max(0, settle_to - warmup)
    return settle, settle + warmup


def load_package():
    pkg_dir = os.path.join(ROOT, "pbf-sph_amd")
    spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["pbf_sph_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def _time_oracle(O, so_path, state, side, fp64, iteration, threads, budget_s, max_steps):
    o = O.Oracle(fp64, so_path=so_path)
    o.set_particles(**state)
    q = O.make_params(iteration=iteration, max_bound=(side,) * 3, mode=O.JACOBI, sort=O.SORT_STABLE, threads=threads)
    t0 = time.perf_counter()
    o.step(q)  # untimed: first touch / OpenMP pool start
    first = time.perf_counter() - t0
    steps, t = 0, 0.0
    while steps < max_steps and (steps == 0 or t + t / steps < budget_s - first):
        a = time.perf_counter()
        o.step(q)
        t += time.perf_counter() - a
        steps += 1
    return steps, t


def host_cores():
    """(physical cores, hardware threads) this process may run on: distinct (physical id, core id) pairs of
    /proc/cpuinfo among the CPUs of the affinity mask.  north_star asks for the CORE count; OpenMP runs one thread per
    hardware thread, reported beside it."""
    cpus = os.sched_getaffinity(0)
    cores, cur = set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = (t.strip() for t in line.split(":", 1))
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in cpus:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in cpus:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except OSError:
        pass
    return (len(cores) or len(cpus)), len(cpus)


def cpu_baseline(state, side, fp64, iteration, budget_s=9.0, max_steps=6):
    """Time the oracle (the checker's source, kind 'port') on the host cores: a bounded sample of the SAME workload,
    started from the GPU's post-run state.  BASELINE.md §4: -O3 -march=native without fast-math is `value`; the
    reference's Release flags (-Ofast -march=native, CMakeLists.txt:136) are reported beside it.  Both are compiled
    on this box (oracle/Makefile `native`).  Reported, never optimised against."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores, threads = host_cores()
    n = len(state["id"])
    try:
        native = O.build_native()
        builds = [("-O3 -march=native -fno-fast-math", native["O3"]), ("-Ofast -march=native", native["Ofast"])]
    except Exception as e:  # no compiler on the box: fall back to the checker build that travelled with the repo
        builds = [(f"-O2 -fno-fast-math (native build failed: {type(e).__name__})", None)]
    res = []
    for flags, path in builds:
        steps, t = _time_oracle(O, path, state, side, fp64, iteration, threads, budget_s, max_steps)
        res.append({"flags": flags, "value": n * steps / t, "ms_per_step": 1e3 * t / steps, "steps": steps, "seconds": t})
    main = res[0]
    out = {"value": main["value"], "unit": "particle-steps/s", "cores": cores, "threads": threads, "kind": "port",
           "sample": f"{main['steps']} steps of the same {n}-particle dam-break state after the timed region "
                     f"(oracle Jacobi mode, OpenMP on {threads} hardware threads = {cores} physical cores, "
                     f"{main['flags']}), {main['seconds']:.1f} s",
           "ms_per_step": main["ms_per_step"], "flags": main["flags"]}
    if len(res) > 1:
        out["reference_release_flags"] = {"flags": res[1]["flags"], "value": res[1]["value"],
                                          "ms_per_step": res[1]["ms_per_step"], "steps": res[1]["steps"]}
    return out, O


def launch_ranks(args):
    """--gpus N without torchrun: start N fresh rank processes of this script (one per GPU) BEFORE this process
    touches the GPU; rank 0 prints the JSON line.  Exit code = the first failing rank's."""
    import torch  # counting devices does not initialise the GPU

    have = torch.cuda.device_count()
    backend = os.environ.get("PBF_BENCH_BACKEND", "nccl")
    if have < args.gpus and backend == "nccl":
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PBF_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("PBF_BENCH_TIMEOUT", "1500"))
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in procs:  # one rank died: the others would wait in a collective for ever
                    q.terminate()
        if time.time() > deadline:
            for q in procs:
                q.kill()
            return rc or 124
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--particles", type=int, default=0,
                    help="nominal particle count: per GPU (weak) or of the whole column (strong); "
                         "default 1 M (N = 1, weak) / 4 M (strong, BASELINE.json configs[3])")
    ap.add_argument("--solver-iter", type=int, default=4)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="strong (default for N > 1) = ONE 4 M dam-break column cut into N load-balanced x-slabs — with "
                         "--gpus 1 the whole 4 M column on one GPU, the first point of a like-for-like sweep; weak "
                         "(default for N = 1) = 1 M per GPU, N mirrored columns side by side")
    ap.add_argument("--settle-to", type=int, default=200,
                    help="simulation frame at which the timed region starts when warmup is smaller (reference CLI "
                         "warm-up default, args.hpp:36)")
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--fast-math", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lds", action="store_true", help="A/B: per-particle global-memory gather kernels")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    # stdout carries the ONE JSON line and nothing else: whatever libraries print to fd 1 while the run lasts
    # (RCCL's version banner, gloo's connection notes) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"communicator has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    n_gpus = dist.get_world_size() if dist is not None else 1

    pkg = load_package()
    flags = (0 if os.environ.get("PBF_BENCH_NO_EVENTS") else pkg.FLAG_STAGE_TIMING) | \
        (pkg.FLAG_FAST_MATH if args.fast_math else 0) | (pkg.FLAG_NO_LDS if args.no_lds else 0)
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    # --scaling strong = BASELINE.json configs[3]'s ONE 4 M column whatever N is (N = 1: the whole column on one GPU, the
    # like-for-like first point of a strong-scaling sweep); default: N = 1 -> the 1 M headline config, N > 1 -> strong
    nominal = args.particles or ((1 << 22) if scaling == "strong" else (1 << 20))
    scene, side = pkg.scene_dambreak(nominal, args.fp64)
    n = len(scene["id"])
    p = pkg.default_params(args.solver_iter, side)
    drv = None
    if world == 1 and not force_slab:
        solver = pkg.Solver(h=0.1, fp64=args.fp64, device=local_rank, flags=flags)
        solver.upload(**scene)
        run = lambda k: solver.steps(p, k)  # noqa: E731
        total_particles = n
    else:
        from pbf_sph_amd import slab
        # the whole slab step — kernels AND exchanges — runs inside libpbf_hip.so (pbf_slab_step): RCCL send/recv over
        # xGMI on the solver's stream.  PBF_BENCH_BACKEND=gloo: rehearsal with several ranks sharing one GPU
        # (host-callback transport).
        solver = pkg.Solver(h=0.1, fp64=args.fp64, device=local_rank, flags=flags)
        if scaling == "weak":
            # `world` dam-break columns side by side along x, one per rank, in ONE box of world*side x side x side;
            # slabs of equal width.  Odd ranks hold the MIRROR image of the column (x -> side - x inside their
            # sub-box): every cut plane is then a mirror plane of the whole set-up, i.e. dynamically a wall — each
            # rank's problem is the N = 1 dam-break and the load stays balanced (up to chaotic symmetry breaking).
            x = scene["pos"][:, 0]
            scene["pos"][:, 0] = (x.dtype.type(side) - x if rank % 2 else x) + x.dtype.type(rank * side)
            scene["id"] += np.uint64(rank * n)
            p.max_bound[0] = world * side
            cuts = slab.even_cuts(world, world * side)
            mine = scene
            total_particles = n * world
            per_rank = n
            rebalance = 0
        else:
            # BASELINE.json configs[3]: ONE column (4 M nominal at N = 8) cut into `world` x-slabs holding equal
            # particle counts (slab.balanced_cuts), re-cut every 4 steps as the column collapses (slab.recut).
            cuts = slab.balanced_cuts(world, scene["pos"][:, 0], side)
            col = slab.columns_of(scene["pos"][:, 0])
            sel = (col >= cuts[rank]) & (col < cuts[rank + 1])
            mine = {k: v[sel] for k, v in scene.items()}
            total_particles = n
            per_rank = max(1, n // world)
            rebalance = int(os.environ.get("PBF_BENCH_REBALANCE", "4"))
        # records in the FIRST message of the two assembly rounds (what does not fit follows in a second, exactly
        # sized exchange): about one boundary column of copies / an ordinary step's migrants
        cap_ghost = max(per_rank // 8, 1 << 14)
        cap_mig = max(per_rank // 64, 1 << 12)
        solver._chk(solver.L.pbf_reserve(solver.ctx, 3 * per_rank + 2 * cap_ghost), "pbf_reserve")  # head-room if the load drifts
        solver.upload(**mine)
        drv = slab.CSlabSolver(solver, dist, torch, rank, world, cuts, cap_mig, cap_ghost,
                               transport="rccl" if backend == "nccl" else "gloo-host", rebalance_every=rebalance,
                               device=local_rank)
        run = lambda k: drv.steps(p, k)  # noqa: E731

    def barrier():
        solver.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- settle + warm-up (untimed); a probe near their end finds the dominant kernel -----------------------
    settle, first_timed = frame_plan(args.steps, args.warmup, args.settle_to)
    pre = settle + args.warmup
    probe = PROBE_STEPS if pre >= 4 * PROBE_STEPS else 0  # never probe in the first frames of the process
    solver.set_option("timing_mask", 0)
    run(pre - probe)
    barrier()
    names = list(solver.stage_times())
    dom_name = None
    if probe:
        solver.set_option("timing_mask", 0xFFFFFFFF)
        solver.reset_stage_times()
        run(probe)
        barrier()
        pr = solver.stage_times()
        composite = {k for k in pr if any(o.startswith(k + "/") and pr[o][1] > 0 for o in pr)}
        dom_name = max((k for k in pr if k not in composite), key=lambda k: pr[k][0] * pr[k][1])
    # timed region: only the dominant kernel is bracketed (HIP events on the solver's stream, 8 records per step
    # instead of ~50); without a probe every stage is (costs ~4 % at 1 M)
    solver.set_option("timing_mask", (1 << names.index(dom_name)) if dom_name else 0xFFFFFFFF)
    solver.reset_stage_times()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    stage = solver.stage_times()
    n_final = solver.n if drv is None else drv.n_owned
    imbalance = 1.0
    if dist is not None:
        t = torch.tensor([elapsed, float(n_final)], device="cuda", dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        assert int(round(float(t[1].item()))) == total_particles, "particles were lost or duplicated across slabs"
        imbalance = float(tmax[1].item()) * world / total_particles
    # ---- per-stage split: SPLIT_STEPS more frames, every stage bracketed (untimed) ---------------------------
    solver.set_option("timing_mask", 0xFFFFFFFF)
    solver.reset_stage_times()
    run(SPLIT_STEPS)
    barrier()
    split = solver.stage_times()
    if dom_name is None:
        composite = {k for k in split if any(o.startswith(k + "/") and split[o][1] > 0 for o in split)}
        dom_name = max((k for k in split if k not in composite), key=lambda k: split[k][0] * split[k][1])

    if rank == 0:
        value = total_particles * args.steps / elapsed
        n_rank = n if drv is None else total_particles / world
        sb = STAGE_BYTES_F64 if args.fp64 else STAGE_BYTES_F32
        per_step = {k: ms * calls / SPLIT_STEPS for k, (ms, calls) in split.items()}
        dom = dom_name
        dom_ms, dom_calls = stage[dom]  # the dominant kernel: every launch of the timed region
        # neighbour-list statistics of the run's last build (always reported)
        cnt = solver.nbr_counts()
        overflow = float((cnt == 0xFFFFFFFF).mean()) if len(cnt) else 0.0
        mean_list = float(np.minimum(cnt, 64).mean()) if len(cnt) else 0.0
        S = 8 if args.fp64 else 4
        if dom == "sph-lambda/list-build":
            # the list build: own pStar + quantised position of every particle once + key in,
            # list length + mean_list entries out per particle (the lists are this kernel's product)
            dom_bytes = 4 * S + 8 + 4 + 4 + 4 * mean_list
            dom_bytes_note = "16 (own pStar) + 8 (quantised positions, once) + 4 (key) in; 4 + 4*mean_list out"
        elif dom == "sph-lambda" and not any(k.startswith("sph-lambda/") and split[k][1] > 0 for k in split):
            # default configuration: ONE kernel builds the lists and computes lambda on the way (k_build_lists_op)
            dom_bytes = 4 * S + 8 + 4 + 4 + 4 * mean_list + 4 + S
            dom_bytes_note = ("list build + lambda in one kernel: 16 (own pStar) + 8 (quantised positions, once) + 4 (cell) + 4 (mass) "
                              "in; 4 (list length) + 4*mean_list (lists) + 4 (lambda) out")
        elif dom == "sph-lambda/gather":
            dom_bytes = sb["sph-lambda"]
            dom_bytes_note = "SURVEY §8(d): lambda 12 R + 4 W per particle"
        else:
            dom_bytes = sb[dom]
            dom_bytes_note = "SURVEY §8(d) per-stage figure"
        achieved = dom_bytes * n_rank / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        bytes_step = (276 + 44 * args.solver_iter) if not args.fp64 else (528 + 88 * args.solver_iter)
        # HBM bytes per launch of the dominant kernel: NOT measured in this run (PMC counters need rocprofv3) — read
        # from the committed separate --pmc passes of this same command line (FETCH_SIZE and WRITE_SIZE collected
        # separately, 2 x FETCH gfx950 correction) and only for the configuration those passes ran
        traffic, traffic_src, limiter = None, None, None
        headline = world == 1 and not args.fp64 and nominal == (1 << 20) and not args.fast_math
        tfile, lfile = first_existing(TRAFFIC_FILES), first_existing(LIMITER_FILES)
        if headline and tfile:
            t = match_kernel(json.load(open(os.path.join(ROOT, tfile))), KERNEL_OF.get(dom, dom))
            if t:
                traffic = t["hbm_bytes_gfx950_corrected"]
                traffic_src = f"file: {tfile} (rocprofv3 --pmc passes of this command, not this run)"
        if headline and lfile:
            # what binds the kernel, as MEASURED by the SQ / TA counter passes of this command (not this run): nothing here
            # is an HBM-bound kernel, the HBM fractions above are reported because the contract asks for them
            lim = match_kernel(json.load(open(os.path.join(ROOT, lfile))), KERNEL_OF.get(dom, dom))
            if lim:
                limiter = dict(lim, source=f"file: {lfile} (rocprofv3 --pmc SQ / TA passes of this command, not this run)")
        # SURVEY.md §8(d)-STRICT fraction: only the stage's algorithmic bytes (lambda: 12 R + 4 W; the neighbour list is an
        # implementation artefact §8(d) does not count); the own-accounting figure above also prices the list the kernel writes
        strict_bytes = sb.get(dom.split("/")[0], dom_bytes)
        achieved_strict = strict_bytes * n_rank / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        out = {
            "metric": "particle-steps/sec (1 M particles, 4 iters) + achieved HBM GB/s, 1/2/4/8 MI355X",
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64" if args.fp64 else "f32",
            "data": "synthetic",
            "config": {"workload": f"dam-break, {total_particles} particles (nominal {nominal}"
                                   f"{' per GPU' if (world > 1 and scaling == 'weak') else ''}), box {side:.0f}^3, "
                                   f"h=0.1, K={args.solver_iter}, dt=0.01245, scale=500; timed = simulation frames "
                                   f"[{first_timed}, {first_timed + args.steps}) after {settle} settle + "
                                   f"{args.warmup} warm-up frames",
                       "particles": total_particles, "solver_iter": args.solver_iter,
                       "settle_frames": settle, "first_timed_frame": first_timed,
                       "math": "fast (v_rsq, fma)" if args.fast_math else "precise (IEEE div/sqrt, no contraction)",
                       "parallelism": "1 GPU, device-resident" if world == 1 else
                                      (f"{world} x-slabs, one rank per GPU, 1-cell ghost layer refreshed after every "
                                       f"lambda/delta launch: pbf_slab_step inside the library, " +
                                       ("ncclSend/ncclRecv over xGMI" if backend == "nccl" else "host-callback transport (gloo rehearsal)") +
                                       f", {drv.rounds // max(1, drv.frame)} exchange rounds per step; " +
                                       (f"{world} columns side by side, odd ones mirrored" if scaling == "weak" else
                                        f"ONE column, particle-balanced cuts, re-cut every {drv.rebalance_every} steps ({drv.stats['recuts']} re-cuts so far)") +
                                       f"; max rank load {imbalance:.2f}x mean")},
            "roofline": {"bound": "hbm", "kernel": KERNEL_OF.get(dom, dom), "stopwatch_entry": dom,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "achieved_strict": achieved_strict, "frac_strict": achieved_strict / HBM_PEAK_GBS,
                         "strict_bytes_per_particle": strict_bytes,
                         "strict_note": "SURVEY.md §8(d) bytes of the stage only (the neighbour list the kernel writes is not counted)",
                         "limiter": limiter,
                         "algorithmic_bytes_per_particle": dom_bytes, "algorithmic_bytes_note": dom_bytes_note,
                         "mean_launch_ms": dom_ms, "launches_timed": dom_calls,
                         "whole_step_GBs": value * bytes_step / 1e9 / world,
                         "whole_step_frac": value * bytes_step / 1e9 / world / HBM_PEAK_GBS,
                         "note": "`bound` names the roofline `peak` is taken from (the contract's HBM figure); the kernel itself "
                                 "is NOT HBM-bound — `limiter` holds the measured issue / texture-path / occupancy figures "
                                 "(SURVEY.md §8d: the neighbour kernels are instruction-issue / latency bound)"},
            "lists": {"mean_list_length": mean_list, "overflow_fraction": overflow},
            # the step's pure streams for contrast ("where >= 50 % of 8 TB/s is physically meaningful")
            "roofline_streaming": {k: {"achieved": sb[k] * n_rank / (split[k][0] * 1e-3) / 1e9, "unit": "GB/s",
                                       "frac": sb[k] * n_rank / (split[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "algorithmic_bytes_per_particle": sb[k], "mean_launch_ms": split[k][0]}
                                   for k in ("sph-finalise", "advect+zindex") if k in split and split[k][0] > 0},
            # the whole lambda / delta stages against SURVEY.md §8d's 16 / 28 B per particle
            "roofline_stage": {k: {"achieved": sb[k] * n_rank / (split[k][0] * 1e-3) / 1e9, "unit": "GB/s",
                                   "frac": sb[k] * n_rank / (split[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "algorithmic_bytes_per_particle": sb[k], "mean_stage_ms": split[k][0]}
                               for k in ("sph-lambda", "sph-delta") if k in split and split[k][0] > 0},
            "stage_ms_per_step": per_step,
            "stage_split_from": f"{SPLIT_STEPS} untimed frames after the timed region, every stage bracketed "
                                f"(sum {sum(v for k, v in per_step.items() if '/' not in k):.3f} ms); roofline: every "
                                f"launch of the timed region",
        }
        if world == 1:
            # SURVEY.md §8(d)'s secondary figure: C candidates per particle are distance-tested once per solver
            # iteration (the list build); the mean_list that survive get exact pair terms twice (lambda, delta-p)
            try:
                C = candidates_per_particle(solver.keys(), solver.table())
                rate = value * args.solver_iter
                out["pairs"] = {"candidates_per_particle": C, "candidate_tests_per_s": rate * C,
                                "pair_terms_per_s": rate * 2 * mean_list}
            except Exception as e:  # diagnostic only
                out["pairs"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:
            state = solver.download()
            cb, O = cpu_baseline(state, side, args.fp64, args.solver_iter)
            # the oracle is also the checker: one more GPU step from the same state must agree
            solver.step(p)
            g = solver.download()
            o2_state = None
            try:
                # device_pow: the oracle evaluates pow(q,4) as (q*q)*(q*q) like the kernels => bit-exact expected
                oo = O.Oracle(args.fp64, device_pow=not args.fast_math)
                oo.set_particles(**state)
                oo.step(O.make_params(iteration=args.solver_iter, max_bound=(side,) * 3, mode=O.JACOBI,
                                      sort=O.SORT_STABLE, threads=cb["threads"]))
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
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if drv is not None:
        solver.sync()
        drv.close()  # ncclCommDestroy before the process group goes away
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
