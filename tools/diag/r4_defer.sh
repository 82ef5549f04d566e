#!/bin/bash
# the MLP residual add done by the next LayerNorm (DCLIP_DEFER_RESIDUAL, dclip_layernorm_fwd_add): parity, then the step A/B
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_defer.log; : > $L
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "layernorm" >> $L 2>&1 || { tail -30 $L; exit 1; }
timeout -k 10 700 python -m pytest tests/test_towers_gpu.py tests/test_configs_gpu.py tests/test_fullsize_gpu.py tests/test_trajectory_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -40 $L; exit 1; }
grep -E "passed|failed" $L
run() {  # tag env...
  tag=$1; shift
  ( env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_defer_$tag.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_defer_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
}
run epilogue DCLIP_DEFER_RESIDUAL=0 && run deferred A=1 && run epilogue2 DCLIP_DEFER_RESIDUAL=0 && run deferred2 A=1
