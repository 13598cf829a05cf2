# (GPU box) round 3's committed profile set, every pass with the DRIVER'S bench command line (bench.py --steps 20 --warmup 5):
#   ks/            rocprofv3 --kernel-trace --stats                      -> kernel_stats.md, bench_under_rocprof.json
#   pmc_FETCH/WRITE separate --pmc passes (HBM bytes)                     -> pmc_traffic.{md,json}
#   sq/pass1..5    SQ / TCP / TCC / TA counter passes                     -> pmc_sq.md, limiter.json
#   bench_driver_cmd.json  the plain driver command (with cpu_baseline)
# usage: bash tools/r03_profiles.sh <out-subdir-of-gpurun_out> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r3prof}; mkdir -p $O; shift
B="--steps 20 --warmup 5 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o run -- python3 $R/bench.py $B --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/ks.err || { echo kernel-stats failed; tail -3 $O/ks.err; exit 1; }
echo kernel-stats done
for cset in FETCH_SIZE WRITE_SIZE; do
  PBF_BENCH_NO_EVENTS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d $O/pmc_$cset -o run -- python3 $R/bench.py $B --no-cpu-baseline > $O/pmc_$cset.log 2>&1 || { echo pmc $cset failed; exit 1; }
  echo pmc $cset done
done
i=0
while read -r cset; do
  i=$((i+1))
  PBF_BENCH_NO_EVENTS=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d $O/sq/pass$i -o run -- python3 $R/bench.py $B --no-cpu-baseline > $O/sq_pass$i.log 2>&1 || { echo "pass $i ($cset) failed"; tail -3 $O/sq_pass$i.log; }
  echo "sq pass $i done"
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
TA_BUSY_avr GRBM_GUI_ACTIVE
SETS
python3 $R/tools/profile_summary.py kernel $O/ks $O/kernel_stats.md 230 200 20 > /dev/null && python3 $R/tools/profile_summary.py pmc $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.md > /dev/null && echo summaries done
python3 $R/tools/pmc_sq_summary.py $O/sq $O/pmc_sq.md $O/limiter.json > /dev/null && echo limiter done
python3 $R/bench.py $B > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err && echo bench done
