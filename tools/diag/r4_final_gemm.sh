#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_final_gemm.log; : > $L
timeout -k 10 500 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
grep -E "passed|failed" $L
run() {  # tag env...
  tag=$1; shift
  ( env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_final_$tag.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_final_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], 'gemm_nt', d['roofline']['others']['gemm_nt_kernel']['ms_per_step'], 'frac', d['roofline']['frac'])
PY
}
run default A=1 && run persist DCLIP_GEMM_PERSIST=1 && run default2 A=1 && run persist2 DCLIP_GEMM_PERSIST=1
( cd _r3 && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_final_r3.json && python -c "
import json; d=json.loads(open('gpurun_out/r4_final_r3.json').read().strip().splitlines()[-1]); print('r3 tree', d['value'], d['ms_per_step'], 'gemm_nt', d['roofline']['others']['gemm_nt_kernel']['ms_per_step'], 'frac', d['roofline']['frac'])"
