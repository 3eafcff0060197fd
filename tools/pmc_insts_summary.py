"""Per-kernel instruction counters of two `rocprofv3 --pmc` passes (tools/pmc_insts.sh):  python tools/pmc_insts_summary.py DIR_A DIR_B WORKLOAD
Counters are summed over all waves of all dispatches of a kernel in ONE execute; gpu-cycles = GRBM_GUI_ACTIVE / 8 XCDs."""
import csv, glob, os, sys
from collections import defaultdict

da, db, wl = sys.argv[1], sys.argv[2], sys.argv[3]
acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
for d in (da, db):
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', row.get('Kernel Name', '?'))
                acc[name][row.get('Counter_Name', '?')] += float(row.get('Counter_Value', 0))
                launches[(d, name)].add(row.get('Dispatch_Id', row.get('Dispatch Id', '')))
T = {'cfg3': 200000, 'cfg3_sqrt': 200000, 'cfg2': 84010, 'cfg4': 88200}.get(wl)
print('rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM | --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (two passes)')
print('-- python3 bench.py --workload %s --steps 1 --warmup 0 --no-cpu-baseline --extras none; one execute (three sweeps); counters summed over all waves' % wl)
rows = []
for name, c in acc.items():
    cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    rows.append((cyc, name, c))
for cyc, name, c in sorted(rows, key=lambda r: -r[0])[:8]:
    short = name.replace('void nagp::', '').split('(')[0]
    waves = c.get('SQ_WAVES', 0.0)
    ln = '%-46s gpu-cycles %.3e  waves %d  VALU %.3e  SALU %.3e  LDS %.3e  SMEM %.3e' % (short[:46], cyc, waves, c.get('SQ_INSTS_VALU', 0), c.get('SQ_INSTS_SALU', 0), c.get('SQ_INSTS_LDS', 0), c.get('SQ_INSTS_SMEM', 0))
    if T and waves and ('adf8' in name or 'ihgp_adf' in name or ('gf_filter_kernel' in name and cyc > 1e8)):
        per = lambda k: c.get(k, 0.0) / waves / T
        ln += '\n    per wave and time step (T = %d): VALU %.0f  SALU %.0f  LDS %.0f  -> %.0f instructions per wave-step; gpu-cycles per step %.0f' % (
            T, per('SQ_INSTS_VALU'), per('SQ_INSTS_SALU'), per('SQ_INSTS_LDS'), per('SQ_INSTS_VALU') + per('SQ_INSTS_SALU') + per('SQ_INSTS_LDS'), cyc / T)
    print(ln)
