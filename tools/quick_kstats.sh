# (GPU box) quick iteration loop: the driver's bench command under rocprofv3 kernel stats + one plain run
# usage: bash tools/quick_kstats.sh <out-subdir-of-gpurun_out> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-qk}; mkdir -p $O; shift
B="--steps 20 --warmup 5 $*"
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o run -- python3 $R/bench.py $B --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/ks.err || { echo kernel-stats failed; tail -3 $O/ks.err; exit 1; }
python3 $R/tools/profile_summary.py kernel $O/ks $O/kernel_stats.md 230 200 20 > /dev/null
head -24 $O/kernel_stats.md | cut -c1-200
for i in 1 2; do
python3 $R/bench.py $B --no-cpu-baseline > $O/bench$i.json 2> $O/bench$i.err
python3 -c "
import json;j=json.load(open('$O/bench$i.json'));print(j['value'],j['ms_per_step'],j.get('post_check'))"
done
