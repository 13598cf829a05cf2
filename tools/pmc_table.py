"""per-kernel table of the SQ counters of tools/quick_pmc.sh (two passes): python tools/pmc_table.py <dir with sq/pass*/>"""
import pandas as pd, glob, sys
fs = glob.glob(sys.argv[1] + '/sq/pass*/run_counter_collection.csv')
df = pd.concat([pd.read_csv(f) for f in fs])
df['k'] = df.Kernel_Name.str.replace('void pbf::', '').str.replace('pbf::', '').str.slice(0, 34)
p = df.pivot_table(index='k', columns='Counter_Name', values='Counter_Value', aggfunc='mean')
p = p[p.SQ_WAVES > 0]
r = pd.DataFrame({'waves': p.SQ_WAVES, 'kcyc/wave': p.SQ_WAVE_CYCLES / p.SQ_WAVES * 4 / 1000, 'wait%': p.SQ_WAIT_ANY / p.SQ_WAVE_CYCLES * 100,
                  'waitinst%': p.SQ_WAIT_INST_ANY / p.SQ_WAVE_CYCLES * 100, 'issue%': p.SQ_ACTIVE_INST_ANY / p.SQ_WAVE_CYCLES * 100,
                  'valu/w': p.SQ_INSTS_VALU / p.SQ_WAVES, 'salu/w': p.SQ_INSTS_SALU / p.SQ_WAVES, 'vmrd/w': p.SQ_INSTS_VMEM_RD / p.SQ_WAVES,
                  'vmwr/w': p.SQ_INSTS_VMEM_WR / p.SQ_WAVES, 'lds/w': p.SQ_INSTS_LDS / p.SQ_WAVES,
                  'lanes': p.SQ_THREAD_CYCLES_VALU / p.SQ_ACTIVE_INST_VALU / 64 if 'SQ_THREAD_CYCLES_VALU' in p else 0})
pd.set_option('display.width', 250)
print(r.round(1).to_string())
