#!/bin/bash
# persistent 256-/320-row GEMM launches (DCLIP_GEMM_PERSIST) and the loss's last-arriver finish: parity first, then the step A/B
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_persist.log; : > $L
timeout -k 10 400 python -m pytest tests/test_gemm_gpu.py tests/test_loss_gpu.py tests/test_kernels_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
# few workgroups: every launch with more than 8 / 16 tiles walks several tiles per workgroup (tile loop, prefetch, column sums at the end)
DCLIP_GEMM_PERSIST=8 timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
DCLIP_GEMM_PERSIST=16 DCLIP_GEMM320=2 timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
DCLIP_GEMM_PERSIST=8 timeout -k 10 300 python -m pytest tests/test_towers_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
grep -E "passed|failed" $L
run() {  # tag env...
  tag=$1; shift
  ( env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_persist_$tag.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_persist_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
}
run off DCLIP_GEMM_PERSIST=0 && run on DCLIP_GEMM_PERSIST=1 && run off2 DCLIP_GEMM_PERSIST=0 && run on2 DCLIP_GEMM_PERSIST=1 && run p128 DCLIP_GEMM_PERSIST=128 && run p512 DCLIP_GEMM_PERSIST=512
