#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() {  # tag env...
  tag=$1; shift
  ( env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_knob_$tag.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_knob_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
}
run base A=1 && run tn9 DCLIP_TN256_MIN_TILES=9 && run base2 A=1 && run tn9b DCLIP_TN256_MIN_TILES=9 && run tnblk512 DCLIP_TN256_BLOCKS=512 && run tnblk240 DCLIP_TN256_BLOCKS=240
