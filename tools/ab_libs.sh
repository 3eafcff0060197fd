#!/bin/bash
# same-box A/B of several builds of libnagp.so: tools/ab_libs.sh "bench args" rounds lib1.so lib2.so ...
L=nonstationary-audio-gp_amd/libnagp.so
cp $L /tmp/libnagp_keep.so
args="$1"; rounds=$2; shift 2
for r in $(seq 1 $rounds); do for which in "$@"; do
  cp "$which" $L
  echo "$(basename $which) [$args]: $(python bench.py $args --no-cpu-baseline --extras none 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],1))')"
done; done
cp /tmp/libnagp_keep.so $L
