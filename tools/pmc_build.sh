# usage (GPU box): PBF_SPLIT_BUILD=4 bash tools/pmc_build.sh <tag> — rocprofv3 --pmc passes (one counter set per run) of a short bench
cd /tmp && export TMPDIR=/tmp && export PBF_BENCH_NO_EVENTS=1
R=$GRAFT_REPO_ROOT; tag=${1:-build}
i=0
for set in ${PMC_SETS:-"SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY"}; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -o run -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 100 > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || exit 1
done
echo done
