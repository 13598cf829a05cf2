# (GPU box) the round's committed profile artefacts, taken with the DRIVER'S bench command line
# (bench.py --steps 20 --warmup 5): rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r2prof}; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/ks_bench.json 2> $O/ks.err || { echo kernel-stats failed; tail -3 $O/ks.err; exit 1; }
echo kernel-stats done
for cset in FETCH_SIZE WRITE_SIZE; do
  PBF_BENCH_NO_EVENTS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d $O/pmc_$cset -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc_$cset.log 2>&1 || { echo pmc $cset failed; exit 1; }
  echo pmc $cset done
done
python3 $R/tools/profile_summary.py kernel $O/ks $O/kernel_stats.md 230 200 20 > /dev/null && python3 $R/tools/profile_summary.py pmc $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.md > /dev/null && echo summaries done
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err && echo bench done
