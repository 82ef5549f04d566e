# python bench.py under single knobs, same box: bash tools/diag/knob_sweep.sh "VAR=VAL" "VAR2=VAL2" ...
run() { env "$@" python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['others']; print(d['value'], d['ms_per_step'], 'gemm_nt', r['gemm_nt_kernel']['ms_per_step'], 'attn', r['attention']['ms_per_step'])"; }
echo "== default"; run DCLIP_NOP=1
for v in "$@"; do echo "== $v"; run $v; done
echo "== default again"; run DCLIP_NOP=1
