# (GPU box) quick SQ counter passes of the driver's bench command (two passes: wave time shares, instruction mix)
# usage: bash tools/quick_pmc.sh <out-subdir-of-gpurun_out> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-qp}; mkdir -p $O; shift
B="--steps 10 --warmup 3 $*"
i=0
while read -r cset; do
  i=$((i+1))
  PBF_BENCH_NO_EVENTS=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d $O/sq/pass$i -o run -- python3 $R/bench.py $B --no-cpu-baseline > $O/sq_pass$i.log 2>&1 || { echo "pass $i ($cset) failed"; tail -3 $O/sq_pass$i.log; }
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
SETS
python3 $R/tools/pmc_table.py $O
