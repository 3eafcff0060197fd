mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8 > gpurun_out/r5_gputests_e.log; cat gpurun_out/r5_gputests_e.log
for wl in cfg2_batch cfg2 cfg2_sqrt; do
  timeout -k 10 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --extras none > gpurun_out/r5_${wl}_e.json 2> gpurun_out/r5_${wl}_e.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5_${wl}_e.json').read().strip().splitlines()[-1])
print('$wl ms_per_step', d['ms_per_step'], d['kernel_ms_per_step'])
PY
done
