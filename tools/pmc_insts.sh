#!/bin/bash
export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
# tools/pmc_insts.sh -- instruction counters of ONE execute per workload: what the waves of the dominant kernel issue per step (the floor of a
# latency-bound sequential kernel is its dependent instructions x the issue interval of a lone wave).  Run on the GPU box from the repository root:
#      bash tools/pmc_insts.sh r05 cfg3
# rocprofv3 --pmc only (two passes, no trace flags; the program directly behind --).  Summary -> profiles/<tag>_pmc_insts_<workload>.txt
set -u
tag=$1; shift
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
for wl in "$@"; do
  for pass in a b; do
    d=gpurun_out/${tag}_pmci_${wl}_$pass
    rm -rf "$d"
    if [ $pass = a ]; then ctr="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"; else ctr="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; fi
    rocprofv3 --pmc $ctr --output-format csv -d "$d" -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --extras none > "$d.log" 2>&1 || { echo "rocprofv3 --pmc failed for $wl ($ctr)"; tail -5 "$d.log"; exit 1; }
  done
  python3 tools/pmc_insts_summary.py gpurun_out/${tag}_pmci_${wl}_a gpurun_out/${tag}_pmci_${wl}_b "$wl" > profiles/${tag}_pmc_insts_${wl}.txt && cat profiles/${tag}_pmc_insts_${wl}.txt
done
