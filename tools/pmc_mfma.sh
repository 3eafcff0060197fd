#!/bin/bash
export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
# tools/pmc_mfma.sh -- matrix-core counters of ONE execute per workload (north_star: "MFMA-busy counters against peak"); run on the GPU box
# from the repository root:      bash tools/pmc_mfma.sh r04 cfg5_fill cfg2_batch cfg3_batch cfg3_sqrt
# rocprofv3 --pmc only (no trace flags; the program directly behind --).  Summaries -> profiles/<tag>_pmc_mfma_<workload>.txt
set -u
tag=$1; shift
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
for wl in "$@"; do
  d=gpurun_out/${tag}_pmcm_${wl}
  rm -rf "$d"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d "$d" -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --extras none > "$d.log" 2>&1 || { echo "rocprofv3 --pmc failed for $wl"; tail -5 "$d.log"; exit 1; }
  python3 tools/pmc_mfma_summary.py "$d" "$wl" > profiles/${tag}_pmc_mfma_${wl}.txt && cat profiles/${tag}_pmc_mfma_${wl}.txt
done
