# (GPU box) hipGraph replay vs eager launches of the resident step: bench.py without HIP events (events force the eager path)
R=$GRAFT_REPO_ROOT
for n in 16384 262144 1048576; do
  for g in 0 1 0 1; do
    PBF_GRAPH=$g PBF_BENCH_NO_EVENTS=1 python3 $R/bench.py --no-cpu-baseline --particles $n --steps 200 --warmup 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('n=$n graph=$g  ms/step', round(d['ms_per_step'],4))"
  done
done
