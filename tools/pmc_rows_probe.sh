cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3rows; mkdir -p $O
i=0
while read -r cset; do
  i=$((i+1))
  PBF_ROW_MAJOR=1 PBF_BENCH_NO_EVENTS=1 timeout -k 10 150 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d $O/pass$i -o run -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 5 > $O/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/pass$i.log; }
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
SETS
python3 $R/tools/pmc_sq_summary.py $O $O/pmc_sq.md 2>&1 | cut -c1-250 | head -8
