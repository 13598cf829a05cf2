"""Condense the SQ / TCP / TCC counter passes of tools/r02_pmc_sq.sh into one table per kernel with the derived
figures DESIGN.md quotes: wave-time shares (issue / issue-stall / memory-wait), VALU instructions per wave, lane
utilisation, the clock the kernel held (SQ_BUSY_CYCLES / 32 shader engines / duration), L1 / L2 requests.
  python tools/pmc_sq_summary.py <dir with pass*/run_counter_collection.csv> <out.md> [limiter.json]
The optional third argument also writes the machine-readable per-kernel "what binds it" record bench.py quotes in
roofline.limiter (wave-time shares, VALU pipe share, texture-address busy share when a TA pass is present, clock held)."""
import glob
import sys

import pandas as pd

d, out = sys.argv[1], sys.argv[2]
limiter_out = sys.argv[3] if len(sys.argv) > 3 else None
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
if limiter_out:
    import json
    rec = {}
    for k, row in r.iterrows():
        e = {"mean_launch_us": float(row["us"]), "clock_ghz_held": float(row["GHz"]), "clock_ghz_peak": 2.4,
             "valu_per_wave": float(row["VALU/wave"]), "vmem_per_wave": float(row["VMEM/wave"]), "lds_per_wave": float(row["LDS/wave"]),
             "lanes_active_frac": float(row["lanes"]), "wave_issue_frac": float(row["issue%"]) / 100,
             "wave_ready_not_issued_frac": float(row["stall%"]) / 100, "wave_waiting_frac": float(row["wait%"]) / 100,
             "valu_pipe_busy_frac": float(row["VALUpipe%"]) / 100, "l2_hit_frac": float(row["L2 hit%"]) / 100}
        if "TA_BUSY_avr" in p.columns and "GRBM_GUI_ACTIVE" in p.columns and p.loc[k, "GRBM_GUI_ACTIVE"] == p.loc[k, "GRBM_GUI_ACTIVE"]:
            e["texture_address_busy_frac"] = round(float(p.loc[k, "TA_BUSY_avr"] / (p.loc[k, "GRBM_GUI_ACTIVE"] / 8)), 3)
        # the verdict a reader should take away: which unit is closest to saturation, and that none is saturated
        shares = {"valu pipe": e["valu_pipe_busy_frac"], "texture-address path": e.get("texture_address_busy_frac", 0.0)}
        top = max(shares, key=shares.get)
        msg = (f"{top} {shares[top]:.0%} busy by instruction COUNT, waves waiting {e['wave_waiting_frac']:.0%} / "
               f"ready-not-issued {e['wave_ready_not_issued_frac']:.0%} of their life: no unit saturated by count")
        if "k_build" in k:
            msg += ("; priced by instruction class (fma / add / mul 2 cycles per wave64 instruction on a SIMD, compares / selects / "
                    "v_dot2 / v_pk_sub ~3.5: profiles/r02_valu_rates.md) the list build is ~75 % VALU-issue bound (DESIGN.md section 5)")
        elif e["wave_waiting_frac"] >= 0.6:
            msg += "; latency bound (waves wait on memory / LDS most of their life)"
        e["measured_limiter"] = msg
        rec[k] = e
    json.dump(rec, open(limiter_out, "w"), indent=1)
