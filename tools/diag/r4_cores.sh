#!/bin/bash
# does a GEMM body that leaves registers free (256-row: 229 VGPRs, 192-row: 189) let the other towers' HBM-bound kernels share its CUs?
set -o pipefail
mkdir -p gpurun_out
run() {  # tag env...
  tag=$1; shift
  ( env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_cores_$tag.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_cores_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
}
run t320 A=1 && run t256 DCLIP_GEMM320=0 && run t192 DCLIP_GEMM320=6 && run t320d DCLIP_DEFER_RESIDUAL=1 && run t256d DCLIP_GEMM320=0 DCLIP_DEFER_RESIDUAL=1 && run t192d DCLIP_GEMM320=6 DCLIP_DEFER_RESIDUAL=1 && run t320d2 DCLIP_DEFER_RESIDUAL=2 && run t320b A=1
