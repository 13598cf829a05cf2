# (GPU box) A/B of environment knobs on the default libpbf_hip.so: same bench command, interleaved twice.
#   bash tools/ab_env.sh <outdir> "PBF_PIPELINE=1" "PBF_COOP=4" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-abenv}; mkdir -p $O; shift
for rep in 1 2; do
  for setting in "NONE=1" "$@"; do
    n=$(echo $setting | tr '= ' '__')
    env $setting python3 $R/bench.py --no-cpu-baseline --steps 100 --warmup 100 > $O/${n}_$rep.json 2> $O/${n}_$rep.err || echo "$n failed"
    python3 - <<PY
import json
d=json.load(open("$O/${n}_$rep.json"))
print("$n", $rep, round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stage_ms_per_step"].items()})
PY
  done
done
