#!/bin/bash
export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
# same-box A/B of several builds of libnagp.so: tools/ab_libs.sh "bench args" rounds lib1.so lib2.so ...
# The candidates are selected through NAGP_LIB (nagp/_lib.py); the in-tree library is never touched.
cd "$(dirname "$0")/.." || exit 1
args="$1"; rounds=$2; shift 2
for r in $(seq 1 $rounds); do for which in "$@"; do
  echo "$(basename $which) [$args]: $(NAGP_LIB="$(realpath "$which")" python bench.py $args --no-cpu-baseline --extras none 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],1))')"
done; done
