#!/bin/bash
export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
# tools/pmc_run.sh -- HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, no trace flags) and the kernel
# statistics (rocprofv3 --kernel-trace --stats) of ONE execute of the benched build, per workload; run on the GPU box from the
# repository root:      bash tools/pmc_run.sh r02 cfg3 cfg2
# Writes gpurun_out/<tag>_pmc_<workload>_{fetch,write}/, gpurun_out/<tag>_stats_<workload>/ and the summaries
# profiles/<tag>_pmc_traffic_<workload>.txt, profiles/<tag>_kernel_stats_<workload>.csv, and refreshes profiles/pmc_traffic.json
# (keyed by the source hash of the library, which bench.py checks before it reports roofline.traffic).
set -u
tag=$1; shift
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
for wl in "$@"; do
  args="bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --extras none"
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/${tag}_pmc_${wl}_$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)
    rm -rf "$d"
    rocprofv3 --pmc $c --output-format csv -d "$d" -- python3 $args > "$d.log" 2>&1 || { echo "rocprofv3 --pmc $c failed for $wl"; tail -5 "$d.log"; exit 1; }
  done
  d=gpurun_out/${tag}_stats_${wl}
  rm -rf "$d"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 $args > "$d.log" 2>&1 || { echo "rocprofv3 --stats failed for $wl"; tail -5 "$d.log"; exit 1; }
  python3 tools/pmc_summary.py --tag "$tag" --workload "$wl" gpurun_out/${tag}_pmc_${wl}_fetch gpurun_out/${tag}_pmc_${wl}_write --stats "$d"
done
