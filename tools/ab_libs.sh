# (GPU box) A/B builds of libpbf_hip.so (pbf-sph_amd/ab/*.so vs the default): same bench command, interleaved twice.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ab}; mkdir -p $O
for rep in 1 2; do
  for lib in default $(ls $R/pbf-sph_amd/ab/*.so 2>/dev/null); do
    n=$(basename $lib .so)
    if [ "$lib" = default ]; then unset PBF_HIP_LIB; else export PBF_HIP_LIB=$lib; fi
    python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 > $O/${n}_$rep.json 2> $O/${n}_$rep.err || echo "$n failed"
    python3 - <<PY
import json
d=json.load(open("$O/${n}_$rep.json"))
print("$n", $rep, round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stage_ms_per_step"].items()})
PY
  done
done
