#!/bin/bash
# same-box A/B: the round-3 tree (_r3/, built in the container) against HEAD, alternating runs
set -o pipefail
mkdir -p gpurun_out
one() {  # dir tag extra-args
  ( cd $1 && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $3 2> /dev/null | tail -1 ) > gpurun_out/ab_$2.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_$2.json').read().strip().splitlines()[-1])
print('$2', d['value'], d['ms_per_step'], 'gemm_nt frac', d['roofline']['frac'], 'others', {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
}
one _r3 r3_a "" && one . head_a "" && one . head_noprobe "--no-clock-probe" && one _r3 r3_b "" && one . head_b ""
