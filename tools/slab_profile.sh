# (GPU box) kernel-level profile of ONE rank running through the slab machinery (pbf_slab_step, RCCL communicator, no neighbours)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-slabprof}; mkdir -p $O
PBF_BENCH_FORCE_SLAB=1 PBF_BENCH_NO_EVENTS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/err.log || tail -3 $O/err.log
python3 - <<PY
import pandas as pd
d=pd.read_csv("$O/ks/run_kernel_stats.csv")
d["us_per_step"]=d.TotalDurationNs/1e3/235
print(d[["Name","Calls","AverageNs","us_per_step"]].sort_values("us_per_step",ascending=False).head(40).to_string())
import json; print(json.load(open("$O/bench.json"))["ms_per_step"])
PY
