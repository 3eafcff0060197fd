#!/bin/bash
export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
# tools/final_measurements.sh -- everything DESIGN.md section 5 cites, in one GPU call (run from the repository root on the GPU box):
# PMC traffic + kernel statistics (tools/pmc_run.sh), rocprofv3 kernel traces -> pipeline timelines, then the default bench line.
# Outputs under gpurun_out/ (copy the summaries into profiles/).
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/pmc_run.sh r04 cfg3 cfg2 cfg4 cfg3_sqrt > gpurun_out/r04_pmc.log 2>&1 && tail -3 gpurun_out/r04_pmc.log &&
for wl in cfg2 cfg4; do
  d=gpurun_out/r04_trace_$wl; rm -rf $d
  rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --extras none > $d.log 2>&1 &&
  python3 tools/trace_timeline.py $d > gpurun_out/r04_pipeline_timeline_$wl.txt || exit 1
done &&
d=gpurun_out/r04_trace_cfg5x1; rm -rf $d; rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload cfg5 --segments 1 --steps 1 --warmup 0 --no-cpu-baseline --extras none > $d.log 2>&1 && python3 tools/trace_timeline.py $d > gpurun_out/r04_pipeline_timeline_cfg5x1.txt &&
d=gpurun_out/r04_trace_cfg5x8; rm -rf $d; rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload cfg5 --steps 1 --warmup 0 --no-cpu-baseline --extras none > $d.log 2>&1 && python3 tools/trace_timeline.py $d > gpurun_out/r04_pipeline_timeline_cfg5x8.txt &&
find gpurun_out/r04_trace_cfg5x8 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_kernel_stats_cfg5x8.csv \; ; find gpurun_out/r04_trace_cfg5x1 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_kernel_stats_cfg5x1.csv \; ; find gpurun_out/r04_trace_* -name "*kernel_trace.csv" -size +20M -delete
for wl in cfg3 cfg2 cfg4 cfg3_sqrt; do cp profiles/r04_pmc_traffic_$wl.txt profiles/r04_kernel_stats_$wl.csv gpurun_out/ 2>/dev/null; done; cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
( time python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err ) 2> gpurun_out/r04_bench_default.time && tail -3 gpurun_out/r04_bench_default.time ; echo FINAL_DONE
