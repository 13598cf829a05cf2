"""Condense the SQ / TCP / TCC counter passes of tools/r02_pmc_sq.sh into one table per kernel with the derived
figures DESIGN.md quotes: wave-time shares (issue / issue-stall / memory-wait), VALU instructions per wave, lane
utilisation, the clock the kernel held (SQ_BUSY_CYCLES / 32 shader engines / duration), L1 / L2 requests.
  python tools/pmc_sq_summary.py <dir with pass*/run_counter_collection.csv> <out.md>"""
import glob
import sys

import pandas as pd

d, out = sys.argv[1], sys.argv[2]
rows = []
for f in sorted(glob.glob(d + "/pass*/run_counter_collection.csv")):
    df = pd.read_csv(f)
    df["dur_us"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
    df["K"] = df.Kernel_Name.str.replace("pbf::", "").str.replace("void ", "").str.split("(").str[0].str.slice(0, 58)
    rows.append(df.groupby(["K", "Counter_Name"]).agg(mean=("Counter_Value", "mean"), dur=("dur_us", "mean")).reset_index())
a = pd.concat(rows)
p = a.pivot_table(index="K", columns="Counter_Name", values="mean")
p["dur_us"] = a.groupby("K").dur.mean()
p = p[p.index.str.contains("k_")]
r = pd.DataFrame(index=p.index)
r["us"] = p.dur_us.round(1)
r["GHz"] = (p.SQ_BUSY_CYCLES / 32 / p.dur_us / 1e3).round(2)
r["waves"] = p.SQ_WAVES.astype(int)
r["VALU/wave"] = (p.SQ_INSTS_VALU / p.SQ_WAVES).round(0)
r["SALU/wave"] = (p.SQ_INSTS_SALU / p.SQ_WAVES).round(0)
r["LDS/wave"] = (p.SQ_INSTS_LDS / p.SQ_WAVES).round(0)
r["VMEM/wave"] = ((p.SQ_INSTS_VMEM_RD + p.SQ_INSTS_VMEM_WR) / p.SQ_WAVES).round(0)
r["lanes"] = (p.SQ_THREAD_CYCLES_VALU / (p.SQ_ACTIVE_INST_VALU * 64)).round(2)
r["issue%"] = (100 * p.SQ_ACTIVE_INST_ANY / p.SQ_WAVE_CYCLES).round(0)
r["stall%"] = (100 * p.SQ_WAIT_INST_ANY / p.SQ_WAVE_CYCLES).round(0)
r["wait%"] = (100 * p.SQ_WAIT_ANY / p.SQ_WAVE_CYCLES).round(0)
# VALU pipe share: wave-instructions x 2 cycles (wave64 on a SIMD-32) over the SIMD-cycles the kernel had
r["VALUpipe%"] = (100 * p.SQ_INSTS_VALU * 2 / (1024 * p.SQ_BUSY_CYCLES / 32)).round(0)
r["L1acc/wave"] = (p.TCP_TOTAL_CACHE_ACCESSES_sum / p.SQ_WAVES).round(0)
r["L1->L2 rd/wave"] = (p.TCP_TCC_READ_REQ_sum / p.SQ_WAVES).round(0)
r["L2 hit%"] = (100 * p.TCC_HIT_sum / (p.TCC_HIT_sum + p.TCC_MISS_sum)).round(0)
r = r.sort_values("us", ascending=False)
with open(out, "w") as fh:
    fh.write("# rocprofv3 --pmc: SQ / TCP / TCC per-kernel means (tools/r02_pmc_sq.sh; bench.py --steps 10 --warmup 190)\n\n"
             "issue% / stall% / wait% = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES (the share of a\n"
             "wave's life spent issuing, ready-but-not-issued, parked on s_waitcnt); lanes = SQ_THREAD_CYCLES_VALU /\n"
             "(64 x SQ_ACTIVE_INST_VALU); GHz = SQ_BUSY_CYCLES / 32 shader engines / duration (the clock the kernel held);\n"
             "VALUpipe% = VALU wave-instructions x 2 cycles / SIMD-cycles available.\n\n")
    fh.write(r.to_markdown())
    fh.write("\n")
print(r.to_string())
