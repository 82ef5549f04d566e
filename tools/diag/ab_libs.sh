# A/B of two builds of the library on one box (tools/diag/bin/lib_a.so, lib_b.so: built by hand, not tracked): bash tools/diag/ab_libs.sh [bench args]
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['others']; print(d['value'], d['ms_per_step'], 'gemm_nt', r['gemm_nt_kernel']['ms_per_step'])"; }
for rep in 1 2; do
  for v in a b; do cp tools/diag/bin/lib_$v.so distillclip_amd/libdistillclip_hip.so; echo "== lib_$v"; run "$@"; done
done
