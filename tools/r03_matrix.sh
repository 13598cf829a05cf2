# (GPU box) round 3: (1) the bench line of EVERY BASELINE.json GPU config with cpu_baseline, driver-style (--steps 20 --warmup 5)
# -> gpurun_out/<tag>/bench_<config>.json (copied to profiles/r03_bench_<config>.json); (2) option A/Bs without the CPU leg;
# (3) the C++ CLI lines.  usage: bash tools/r03_matrix.sh <tag> [configs|ab|cli ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r3matrix}; mkdir -p $O; shift
WHAT="${*:-configs ab cli}"
show() { python3 - <<PY
import json
try:
    d=json.load(open("$O/$1.json"))
    cb=d.get("cpu_baseline") or {}
    print("$1", "%.4g" % d["value"], "particle-steps/s", "%.4f" % d["ms_per_step"], "ms/step", d["dtype"], d["roofline"]["kernel"], "%.4f" % d["roofline"]["mean_launch_ms"], "cpu %.3g (%s cores)" % (cb.get("value", 0), cb.get("cores")), "bit-exact" if d.get("parity_check_bit_exact") else "")
except Exception as e: print("$1 ERR", e)
PY
}
run() { n=$1; shift; "$@" > $O/$n.json 2> $O/$n.err || echo "$n failed"; show $n; }
for w in $WHAT; do case $w in
configs)
  run bench_256k python3 $R/bench.py --steps 20 --warmup 5 --particles 262144
  run bench_1m python3 $R/bench.py --steps 20 --warmup 5
  run bench_1m_fp64 python3 $R/bench.py --steps 20 --warmup 5 --fp64
  run bench_4m_1gpu python3 $R/bench.py --steps 20 --warmup 5 --gpus 1 --scaling strong
  ;;
ab)
  run ab_1m_default python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5
  run ab_1m_fast python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --fast-math
  PBF_BENCH_FORCE_SLAB=1 run ab_1m_slab_1rank python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5
  PBF_PIPELINE=0 run ab_fp64_pipeline0 python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --fp64
  PBF_PIPELINE=1 run ab_fp64_pipeline1 python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --fp64
  PBF_OVERLAP_DIFFUSE=0 run ab_256k_diffuse_serial python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --particles 262144
  PBF_OVERLAP_DIFFUSE=1 run ab_256k_diffuse_overlapped python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --particles 262144
  run ab_256k_fp64 python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --fp64 --particles 262144
  ;;
cli)
  B=$R/pbf-sph_amd/benchmark
  $B --scene dam-break --particles 1048576 --solver-iter 4 -n 100 -w 20 --no-surface -o "" 2>&1 | grep "Particle-steps\|Frame-time mean" | sed 's/^/cli advance 1M: /'
  $B --scene dam-break --particles 1048576 --solver-iter 4 -n 100 -w 20 --no-surface --resident -o "" 2>&1 | grep "Particle-steps\|Frame-time mean" | sed 's/^/cli resident 1M: /'
  $B --scene dam-break --particles 1048576 --solver-iter 4 -n 100 -w 20 -o "" 2>&1 | grep "Particle-steps\|Frame-time mean\|Vertex" | sed 's/^/cli advance 1M +surface: /'
  $B -n 200 -w 200 -o "" 2>&1 | grep "Particle-steps\|Frame-time mean\|Vertex\|Particle count" | sed 's/^/cli stock: /'
  $B --scene dam-break --particles 1048576 --solver-iter 4 -n 50 -w 20 --slabs 2 --no-surface -o "" 2>&1 | grep "Particle-steps\|Frame-time mean" | sed 's/^/cli 1M 2 slabs one GPU (host-staged exchange), no surface: /'
  $B --scene dam-break --particles 1048576 --solver-iter 4 -n 50 -w 20 --slabs 2 -o "" 2>&1 | grep "Particle-steps\|Frame-time mean\|Vertex" | sed 's/^/cli 1M 2 slabs one GPU, surface across the slabs: /'
  ;;
esac; done
