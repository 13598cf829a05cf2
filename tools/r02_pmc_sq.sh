# (GPU box) SQ / TCP counter passes over the default bench (short run, no HIP events): where do the iteration
# kernels' cycles go — issue, wait, lanes, L1/L2 requests.  One counter set per pass (gfx950: 8 SQ slots).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r2pmc}
mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
i=0
while read -r cset; do
  i=$((i+1))
  PBF_BENCH_NO_EVENTS=1 timeout -k 10 150 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d $O/pass$i -o run -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 190 > $O/pass$i.log 2>&1 || { echo "pass $i ($cset) failed"; tail -3 $O/pass$i.log; }
  echo "pass $i done: $cset"
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
SETS
python3 $R/tools/profile_summary.py pmc $O/pass* $O/pmc_sq.md > /dev/null 2>&1 || echo summary failed
