# (GPU box) the round's committed profile artefacts: kernel stats + FETCH_SIZE / WRITE_SIZE passes of the default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ks -o run -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_ks.json 2> $R/gpurun_out/prof_ks.err || exit 1
echo kernel-stats done
for cset in FETCH_SIZE WRITE_SIZE; do
  PBF_BENCH_NO_EVENTS=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d $R/gpurun_out/prof_pmc_$cset -o run -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 150 > $R/gpurun_out/prof_pmc_$cset.log 2>&1 || exit 1
  echo pmc $cset done
done
