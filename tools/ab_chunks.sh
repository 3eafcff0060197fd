export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
for n in ${CHUNK_LIST:-12 24 36}; do for wl in "cfg2" "cfg4" "cfg5 --segments 1" "cfg5"; do
  echo "chunks $n $wl: $(NAGP_CHUNKS=$n python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --extras none 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])')"
done; done
