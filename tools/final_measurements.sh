#!/bin/bash
export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
# tools/final_measurements.sh <tag> <part> -- everything DESIGN.md section 5 cites (run from the repository root on the GPU box; two GPU calls:
# a call is limited to twenty minutes):
#   part 1: PMC traffic + kernel statistics (tools/pmc_run.sh) of the single-GPU configurations, instruction counters of the headline kernel
#   part 2: PMC traffic of the fill-the-chip workloads, matrix-core counters, kernel traces -> pipeline timelines, gain kernel phase stamps, the default bench line
# Outputs under gpurun_out/ (tools/collect_final.sh copies the summaries into profiles/).
set -o pipefail
tag=${1:-r05}; part=${2:-1}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$part" = 1 ]; then
  bash tools/pmc_run.sh $tag cfg3 cfg2 cfg4 cfg3_sqrt > gpurun_out/${tag}_pmc.log 2>&1 && tail -3 gpurun_out/${tag}_pmc.log &&
  bash tools/pmc_insts.sh $tag cfg3 > gpurun_out/${tag}_pmci.log 2>&1 && tail -4 gpurun_out/${tag}_pmci.log
  for wl in cfg3 cfg2 cfg4 cfg3_sqrt; do cp profiles/${tag}_pmc_traffic_$wl.txt profiles/${tag}_kernel_stats_$wl.csv gpurun_out/ 2>/dev/null; done
  cp profiles/${tag}_pmc_insts_cfg3.txt gpurun_out/ 2>/dev/null; cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_part1.json
  echo FINAL_PART1_DONE
else
  # (gpurun_out/ does not travel to the GPU box: between the two calls, copy the merged gpurun_out/pmc_traffic_part1.json over profiles/pmc_traffic.json HERE,
  #  so that part 2 adds its workloads to part 1's records)
  bash tools/pmc_run.sh $tag cfg5_fill cfg2_batch > gpurun_out/${tag}_pmc2.log 2>&1 && tail -3 gpurun_out/${tag}_pmc2.log &&
  bash tools/pmc_mfma.sh $tag cfg5_fill cfg2_batch > gpurun_out/${tag}_pmcm.log 2>&1 && tail -3 gpurun_out/${tag}_pmcm.log &&
  for wl in cfg2 cfg4 cfg2_batch; do
    d=gpurun_out/${tag}_trace_$wl; rm -rf $d
    rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --extras none > $d.log 2>&1 &&
    python3 tools/trace_timeline.py $d > gpurun_out/${tag}_pipeline_timeline_$wl.txt || exit 1
  done &&
  d=gpurun_out/${tag}_trace_cfg5x8; rm -rf $d; rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload cfg5 --steps 1 --warmup 0 --no-cpu-baseline --extras none > $d.log 2>&1 && python3 tools/trace_timeline.py $d > gpurun_out/${tag}_pipeline_timeline_cfg5x8.txt &&
  find gpurun_out/${tag}_trace_cfg5x8 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats_cfg5x8.csv \; ; find gpurun_out/${tag}_trace_* -name "*kernel_trace.csv" -size +20M -delete
  for wl in cfg5_fill cfg2_batch; do cp profiles/${tag}_pmc_traffic_$wl.txt profiles/${tag}_kernel_stats_$wl.csv profiles/${tag}_pmc_mfma_$wl.txt gpurun_out/ 2>/dev/null; done; cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
  { for a in "19 4096 0 1" "19 4096 0 0" "38 4096 1 1" "38 4096 1 0"; do tools/ubench/gain_check $a; done; tools/ubench/chol16; } > gpurun_out/${tag}_gain_check.txt 2>&1
  ( time python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err ) 2> gpurun_out/${tag}_bench_default.time && tail -3 gpurun_out/${tag}_bench_default.time ; echo FINAL_PART2_DONE
fi
