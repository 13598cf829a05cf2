# (GPU box) one counter pass: how busy is the texture-address / L1 path per kernel?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ta}; mkdir -p $O
PBF_BENCH_NO_EVENTS=1 timeout -k 10 90 rocprofv3 --kernel-trace --pmc TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $O/pass1 -o run -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 190 > $O/pass1.log 2>&1 || tail -3 $O/pass1.log
PBF_BENCH_NO_EVENTS=1 timeout -k 10 90 rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $O/pass2 -o run -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 190 > $O/pass2.log 2>&1 || tail -3 $O/pass2.log
python3 - <<PY
import pandas as pd, glob
for f in sorted(glob.glob("$O/pass*/run_counter_collection.csv")):
    df = pd.read_csv(f)
    df["K"] = df.Kernel_Name.str.replace("pbf::", "").str.replace("void ", "").str.split("(").str[0].str.slice(0, 50)
    df["dur_us"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
    p = df.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    p["us"] = df.groupby("K").dur_us.mean()
    p = p[p.us > 10].sort_values("us", ascending=False)
    print(p.round(0).to_string())
PY
