#!/bin/bash
# loss: default (rows, A, B, finalize + scalars) vs DCLIP_LOSS_FUSE=1 (last arrivers inside stripe B); GEMM shapes with / without persistence
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_persist2.log; : > $L
timeout -k 10 300 python -m pytest tests/test_loss_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
DCLIP_LOSS_FUSE=1 timeout -k 10 300 python -m pytest tests/test_loss_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; }
grep -E "passed|failed" $L
for pm in 0 1; do
  echo "== DCLIP_GEMM_PERSIST=$pm" | tee -a $L
  DCLIP_GEMM_PERSIST=$pm timeout -k 10 200 python tools/diag/gemm_step_shapes.py 2>&1 | tee -a $L || exit 1
done
run() {  # tag env...
  tag=$1; shift
  ( env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_persist2_$tag.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_persist2_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
}
run lossdef A=1 && run lossfuse DCLIP_LOSS_FUSE=1 && run lossdef2 A=1
