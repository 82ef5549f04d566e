#!/bin/bash
# one im2row for teacher + student image towers (shared_image_patches): parity, then the step A/B
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_share.log; : > $L
timeout -k 10 600 python -m pytest tests/test_towers_gpu.py tests/test_configs_gpu.py tests/test_trajectory_gpu.py tests/test_amp_gpu.py tests/test_parallel_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -40 $L; exit 1; }
grep -E "passed|failed" $L
run() {  # tag env...
  tag=$1; shift
  ( env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_share_$tag.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_share_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
}
run own DCLIP_SHARE_PATCHES=0 && run shared A=1 && run own2 DCLIP_SHARE_PATCHES=0 && run shared2 A=1
( timeout -k 10 300 python bench.py --config image --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_share_image.json && python -c "
import json; d=json.loads(open('gpurun_out/r4_share_image.json').read().strip().splitlines()[-1]); print('image shared', d['value'], d['ms_per_step'])"
( DCLIP_SHARE_PATCHES=0 timeout -k 10 300 python bench.py --config image --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_share_image0.json && python -c "
import json; d=json.loads(open('gpurun_out/r4_share_image0.json').read().strip().splitlines()[-1]); print('image own', d['value'], d['ms_per_step'])"
