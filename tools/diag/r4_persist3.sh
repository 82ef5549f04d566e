#!/bin/bash
# persistent GEMM launches with ticket counters: parity with tiny grids (every launch walks many tiles), then per-shape and step A/B
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_persist3.log; : > $L
timeout -k 10 150 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
DCLIP_GEMM_PERSIST=8 timeout -k 10 150 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
DCLIP_GEMM_PERSIST=16 DCLIP_GEMM320=2 timeout -k 10 150 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
DCLIP_GEMM_PERSIST=8 DCLIP_GEMM_TICKETS=0 timeout -k 10 150 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
DCLIP_GEMM_PERSIST=8 timeout -k 10 300 python -m pytest tests/test_towers_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
grep -E "passed|failed" $L
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  echo "== DCLIP_GEMM_PERSIST=$1 DCLIP_GEMM_TICKETS=$2" | tee -a $L
  DCLIP_GEMM_PERSIST=$1 DCLIP_GEMM_TICKETS=$2 timeout -k 10 200 python tools/diag/gemm_step_shapes.py 2>&1 | grep -v amdgpu.ids | tee -a $L || exit 1
done
run() {  # tag env...
  tag=$1; shift
  ( env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_persist3_$tag.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_persist3_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
}
run off DCLIP_GEMM_PERSIST=0 && run static DCLIP_GEMM_TICKETS=0 && run tickets A=1 && run off2 DCLIP_GEMM_PERSIST=0 && run static2 DCLIP_GEMM_TICKETS=0 && run tickets2 A=1
