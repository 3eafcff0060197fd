#!/bin/bash
export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
# A/B of the IHGP ADF kernel forms on one box: NAGP_STAMP_WORKER = 0 (default), 64 (tables / q0 / s0 between two barriers on wave 1 and workers 3, 4)
for v in ${FORMS:-0 64 0 64}; do
  echo "dbg $v: $(NAGP_STAMP_WORKER=$v python bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline --extras none 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])')"
done
for v in ${FORMS_STAMPS:-0 64}; do
  echo "== NAGP_STAMP_WORKER=$v"
  NAGP_STAMPS=1 NAGP_STAMP_WORKER=$v timeout -k 10 120 python tools/gpu_perf_probe.py cfg3 20000 2>&1 | grep -a "worker wave" | head -2 | cut -c1-330
done
