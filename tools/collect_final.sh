#!/bin/bash
export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
# tools/collect_final.sh -- after `gpurun -- bash tools/final_measurements.sh`: copy the merged summaries from gpurun_out/ into profiles/
cd "$(dirname "$0")/.." || exit 1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/r03_pmc_traffic_*.txt gpurun_out/r03_kernel_stats_*.csv gpurun_out/r03_pipeline_timeline_*.txt profiles/
cp gpurun_out/r03_bench_default.json profiles/r03_bench_default.json
rm -rf gpurun_out/r03_pmc_cfg* gpurun_out/r03_stats_cfg* gpurun_out/r03_trace_*
python3 - <<'PY'
import json, sys
sys.path.insert(0, 'nonstationary-audio-gp_amd')
from nagp import _lib
d = json.loads(open('profiles/r03_bench_default.json').read().strip().splitlines()[-1])
pm = json.load(open('profiles/pmc_traffic.json'))
print('library source hash %s; PMC hash %s' % (_lib.source_hash()[:12], pm['cfg3']['source_hash'][:12]))
def brief(n, x):
    r = x.get('roofline', {})
    print('%-11s value %10.0f  ms/execute %8.1f  traffic %s' % (n, x['value'], x['ms_per_step'], r.get('traffic')))
brief('cfg3', d)
for k in ('cfg5_strong', 'cfg5_fill', 'cfg3_batch', 'cfg2_batch', 'cfg2', 'cfg4'):
    brief(k, d[k])
print({k: round(v, 1) for k, v in d.items() if k.startswith('speedup')})
PY
for f in cfg5x1 cfg5x8 cfg2 cfg4; do echo "== $f: $(sed -n 1p profiles/r03_pipeline_timeline_$f.txt)"; grep "wall clock" profiles/r03_pipeline_timeline_$f.txt; done
