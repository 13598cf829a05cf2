# (GPU box) the numbers DESIGN.md §6 quotes: bench.py over the BASELINE.json configs + options, and the C++ CLI.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r2matrix}; mkdir -p $O
run() { n=$1; shift; "$@" > $O/$n.json 2> $O/$n.err || echo "$n failed"; python3 - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("$n", "%.4g" % d["value"], "particle-steps/s", "%.4f" % d["ms_per_step"], "ms/step", d["config"]["math"][:8], d["roofline"]["kernel"], "%.4f" % d["roofline"]["mean_launch_ms"])
except Exception as e: print("$n ERR", e)
PY
}
run 1m_precise python3 $R/bench.py --no-cpu-baseline
run 1m_fast python3 $R/bench.py --no-cpu-baseline --fast-math
for c in 2 4 8; do PBF_COOP=$c run 1m_coop$c python3 $R/bench.py --no-cpu-baseline; done
PBF_COOP=4 run 1m_fast_coop4 python3 $R/bench.py --no-cpu-baseline --fast-math
run 256k python3 $R/bench.py --no-cpu-baseline --particles 262144
run 1m_fp64 python3 $R/bench.py --no-cpu-baseline --fp64
run 256k_fp64 python3 $R/bench.py --no-cpu-baseline --fp64 --particles 262144
run 4m python3 $R/bench.py --no-cpu-baseline --particles 4194304 --steps 100
PBF_BENCH_FORCE_SLAB=1 run 1m_slab_1rank python3 $R/bench.py --no-cpu-baseline
B=$R/pbf-sph_amd/benchmark
$B --scene dam-break --particles 1048576 --solver-iter 4 -n 100 -w 20 --no-surface -o "" 2>&1 | grep "Particle-steps\|Frame-time mean" | sed 's/^/cli advance 1M: /'
$B --scene dam-break --particles 1048576 --solver-iter 4 -n 100 -w 20 --no-surface --resident -o "" 2>&1 | grep "Particle-steps\|Frame-time mean" | sed 's/^/cli resident 1M: /'
$B --scene dam-break --particles 1048576 --solver-iter 4 -n 100 -w 20 -o "" 2>&1 | grep "Particle-steps\|Frame-time mean\|Vertex" | sed 's/^/cli advance 1M +surface: /'
$B -n 200 -w 200 -o "" 2>&1 | grep "Particle-steps\|Frame-time mean\|Vertex\|Particle count" | sed 's/^/cli stock: /'
$B --scene dam-break --particles 1048576 --solver-iter 4 -n 50 -w 20 --slabs 2 -o "" 2>&1 | grep "Particle-steps\|Frame-time mean" | sed 's/^/cli 1M 2 slabs one GPU: /'
