# world-size-1 forced exchange (DCLIP_FORCE_DIST=1) against the plain step, same box: bash tools/diag/dist1_variants.sh
for v in "" "DCLIP_DP_SPARSE_EMBED=0" "DCLIP_DP_MODE=allreduce" "DCLIP_DP_TOWER_GROUPS=1"; do
  echo "== FORCE_DIST $v"
  env DCLIP_FORCE_DIST=1 $v python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
echo "== plain"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
