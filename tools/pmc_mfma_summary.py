"""Per-kernel matrix-core summary of a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE` run
(tools/pmc_mfma.sh):  python tools/pmc_mfma_summary.py DIR WORKLOAD
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); FP64 MFMA flops = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512
(the conventions of profiles/r01_pmc_mfma_batch64_cfg2.txt); TFLOP/s at the 2.4 GHz the peak of 78.6 TFLOP/s is quoted on."""
import csv, glob, os, sys
from collections import defaultdict

d, wl = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get('Kernel_Name', row.get('Kernel Name', '?'))
            acc[name][row.get('Counter_Name', '?')] += float(row.get('Counter_Value', 0))
            launches[name].add(row.get('Dispatch_Id', row.get('Dispatch Id', '')))
print('rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -- python3 bench.py --workload %s --steps 1 --warmup 0 --no-cpu-baseline --extras none' % wl)
print('one execute (three sweeps).  GRBM_GUI_ACTIVE is summed over the 8 XCDs (divided by 8 below).  MfmaUtil = MFMA_BUSY / (gpu-cycles * 1024 SIMDs);')
print('FP64 MFMA flops = MOPS_F64 * 512; TFLOP/s = flops / (gpu-cycles / 2.4 GHz); peak 78.6 TFLOP/s')
rows = []
for name, c in acc.items():
    cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    if cyc <= 0:
        continue
    busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0); mops = c.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0.0)
    fl = mops * 512.0
    rows.append((cyc, name, len(launches[name]), busy / (cyc * 1024.0), fl, fl / (cyc / 2.4e9) / 1e12))
for cyc, name, n, util, fl, tf in sorted(rows, reverse=True):
    if cyc < 1e5:
        continue
    short = name.replace('void nagp::', '').split('(')[0]
    print('%-52s launches %3d  gpu-cycles %.3e  MfmaUtil %5.1f%%  FP64-MFMA flops %.3e  (%.1f TFLOP/s = %.1f%% of peak)' % (short[:52], n, cyc, 100 * util, fl, tf, 100 * tf / 78.6))
