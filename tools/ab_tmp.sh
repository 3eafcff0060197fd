for i in 1 2 3; do
  for w in 16 0; do
    NAGP_STAMP_WORKER=$w python bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline --extras none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NAGP_STAMP_WORKER=$w  value %.0f  filter %.1f ms' % (d['value'], d['kernel_ms_per_step']['filter']))"
  done
done
