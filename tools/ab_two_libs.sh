#!/bin/bash
# same-box A/B of two builds of libnagp.so: tools/ab_two_libs.sh old.so new.so "bench args" [rounds]
L=nonstationary-audio-gp_amd/libnagp.so
cp $L /tmp/libnagp_keep.so
for r in $(seq 1 ${4:-2}); do for which in "$1" "$2"; do
  cp "$which" $L
  echo "$which: $(python bench.py $3 --no-cpu-baseline --extras none 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])')"
done; done
cp /tmp/libnagp_keep.so $L
