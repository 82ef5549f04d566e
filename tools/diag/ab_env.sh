# alternate environments on one box, two rounds: bash tools/diag/ab_env.sh "VAR=VAL [VAR2=VAL2]" "..."   (first entry "-" = plain)
run() { env "$@" python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['others']; print(d['value'], d['ms_per_step'], 'gemm_nt', r['gemm_nt_kernel']['ms_per_step'], 'attn', r['attention']['ms_per_step'])"; }
for rep in 1 2; do for v in "$@"; do echo "== $v"; if [ "$v" = "-" ]; then run DCLIP_NOP=1; else run $v; fi; done; done
