#!/bin/bash
# how the timed window changes the number (power-limited MFMA loop: clocks settle over the first seconds)
set -o pipefail
mkdir -p gpurun_out
run() {  # tag steps warmup
  ( timeout -k 10 400 python bench.py --steps $2 --warmup $3 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_window_$1.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_window_$1.json').read().strip().splitlines()[-1])
print('$1 steps $2 warmup $3:', d['value'], d['ms_per_step'], 'gemm_nt frac', d['roofline']['frac'], 'clock', d['clock_mhz_during_timed_steps']['median'])
PY
}
run a20 20 5 && run a50 50 10 && run a200 200 20 && run b20 20 5 && run b50 50 10 && run b200 200 20
