#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
one() {
  ( cd $1 && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $3 2> /dev/null | tail -1 ) > gpurun_out/ab_$2.json || return 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_$2.json').read().strip().splitlines()[-1])
print('$2', d['value'], d['ms_per_step'], 'gemm_nt frac', d['roofline']['frac'], 'clock', d.get('clock_mhz_during_timed_steps'))
PY
}
one . head_clk_a "" && one . head_noclk_a "--no-clock-probe" && one _r3 r3_c "" && one . head_clk_b "" && one . head_noclk_b "--no-clock-probe" || exit 1
timeout -k 10 200 python tools/diag/gemm_phases.py > gpurun_out/r4_gemm_phases.log 2>&1 || { tail gpurun_out/r4_gemm_phases.log; exit 1; }
DCLIP_GEMM_DUO=2 timeout -k 10 200 python tools/diag/duo_phases.py > gpurun_out/r4_duo_phases.log 2>&1 || { tail gpurun_out/r4_duo_phases.log; exit 1; }
tail -14 gpurun_out/r4_gemm_phases.log
