# (GPU box) the measurement matrix quoted in DESIGN.md §6
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/matrix.txt; : > $O
run() { tag=$1; shift; timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', d['roofline']['kernel'], round(d['roofline']['mean_launch_ms'],4))" >> $O || echo "$tag FAILED" >> $O; }
run fast1M --fast-math
run f32_256K --particles 262144
run f64_1M --fp64
run f64_256K --fp64 --particles 262144
run f32_4M --particles 4194304 --steps 60 --warmup 40
B=$R/pbf-sph_amd/benchmark
timeout -k 10 200 $B --scene dam-break --particles 1048576 --solver-iter 4 -n 100 -w 20 --resident --json --no-surface 2>&1 | grep particle_steps >> $O
timeout -k 10 200 $B --scene dam-break --particles 1048576 --solver-iter 4 -n 40 -w 10 --json --no-surface 2>&1 | grep particle_steps >> $O
timeout -k 10 200 $B --json 2>&1 | grep -E "particle_steps|Vertex" >> $O
cat $O
