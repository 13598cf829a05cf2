# usage: bash tools/kstats.sh <tag> [bench args...]   (on the GPU box) — rocprofv3 kernel stats of one bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -o run -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/ks_$tag.log 2>&1 || exit 1
python3 $R/tools/profile_summary.py kernel $R/gpurun_out/ks_$tag $R/gpurun_out/ks_$tag.md | cut -c1-150 | head -12
