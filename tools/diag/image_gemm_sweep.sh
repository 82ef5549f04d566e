for v in "" "DCLIP_GEMM256=1" "DCLIP_GEMM_SPLITM=1" "DCLIP_GEMM320=0" "DCLIP_GEMM256=3"; do
  echo "== $v"
  env $v python bench.py --config image --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['others']['gemm_nt_kernel']['ms_per_step'])"
done
