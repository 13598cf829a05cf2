"""Condense rocprofv3 CSV output into the per-round summaries kept under profiles/.

  python tools/profile_summary.py kernel <dir with *_kernel_stats.csv> <out.md> [total_steps first_timed n_timed]
  python tools/profile_summary.py pmc <dir with *_counter_collection.csv> ... <out.md>
"""
import glob
import os
import sys

import pandas as pd


def short(name):
    name = name.replace("pbf::", "")
    for a, b in (("StepConsts<float>", "C"), ("StepConsts<double>", "C"), ("vec4_of<float>::type", "f4"),
                 ("vec4_of<double>::type", "d4"), ("unsigned int", "u32"), ("unsigned char", "u8")):
        name = name.replace(a, b)
    return name.split("(")[0][:90]


def kernel(d, out, total_steps=None, first_timed=None, n_timed=None):
    """total_steps / first_timed / n_timed (bench.py's frame plan): adds, from the kernel trace, each kernel's mean
    duration over the dispatches of the TIMED frames only — the figure bench.py's HIP-event mean must agree with."""
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    df = pd.read_csv(f)
    df["Name"] = df["Name"].map(short)
    df["AverageUs"] = (df["AverageNs"] / 1e3).round(2)
    df["TotalMs"] = (df["TotalDurationNs"] / 1e6).round(3)
    cols = ["Name", "Calls", "TotalMs", "AverageUs", "Percentage", "MinNs", "MaxNs"]
    note = ""
    if total_steps:
        t = pd.read_csv(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]).sort_values("Start_Timestamp")
        t["Name"] = t["Kernel_Name"].map(short)
        t["us"] = (t["End_Timestamp"] - t["Start_Timestamp"]) / 1e3
        timed = {}
        for name, g in t.groupby("Name"):
            if len(g) % total_steps == 0 and len(g) >= total_steps:
                per = len(g) // total_steps
                timed[name] = round(float(g["us"].iloc[first_timed * per:(first_timed + n_timed) * per].mean()), 2)
        df["TimedFramesUs"] = df["Name"].map(timed)
        cols.insert(4, "TimedFramesUs")
        note = (f"TimedFramesUs = mean over the dispatches of simulation frames [{first_timed}, {first_timed + n_timed}) "
                f"(the timed region of `bench.py --steps {n_timed}`; {total_steps} frames per run); AverageUs covers all frames "
                f"incl. the settle phase, where the lists are shorter.\n\n")
    with open(out, "w") as fh:
        fh.write(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(f)})\n\n{note}")
        fh.write(df[cols].to_markdown(index=False))
        fh.write("\n")
    print(df[cols].to_string(index=False))


def pmc(dirs, out):
    rows = []
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            df = pd.read_csv(f)
            df["Name"] = df["Kernel_Name"].map(short)
            df["dur_us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
            g = df.groupby(["Name", "Counter_Name"]).agg(dispatches=("Counter_Value", "size"), mean=("Counter_Value", "mean"),
                                                         dur_us=("dur_us", "mean")).reset_index()
            rows.append(g)
    allr = pd.concat(rows)
    with open(out, "w") as fh:
        fh.write("# rocprofv3 --pmc per-kernel means (one counter set per pass)\n\n")
        fh.write("FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on gfx950 FETCH_SIZE reads 1/2 of the bytes of a wide\n"
                 "coalesced stream (MI355X_MICROARCH.md §HBM), so HBM read bytes ~= 2 x FETCH_SIZE x 1024 for streaming kernels\n"
                 "(uncalibrated for the gather kernels' 16-byte per-lane accesses).\n\n")
        fh.write(allr.round(2).to_markdown(index=False))
        fh.write("\n")
    print(allr.round(2).to_string(index=False))
    # machine-readable per-kernel HBM traffic (bench.py quotes it in roofline.traffic)
    import json
    piv = allr.pivot_table(index="Name", columns="Counter_Name", values="mean")
    traffic = {}
    for name, row in piv.iterrows():
        f, w = row.get("FETCH_SIZE"), row.get("WRITE_SIZE")
        if f == f and w == w:  # both present
            traffic[name] = {"fetch_kib": float(f), "write_kib": float(w),
                             "hbm_bytes_gfx950_corrected": float((2 * f + w) * 1024)}
    with open(os.path.splitext(out)[0] + ".json", "w") as fh:
        json.dump(traffic, fh, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel(sys.argv[2], sys.argv[3], *[int(v) for v in sys.argv[4:7]])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
