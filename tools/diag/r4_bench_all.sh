#!/bin/bash
# bench lines of every BASELINE configuration (with cpu_baseline) for profiles/r04_bench*.json
set -o pipefail
mkdir -p gpurun_out
for c in lclip image text lclip336; do
  sfx=$([ $c = lclip ] && echo "" || echo "_$c")
  timeout -k 10 400 python bench.py --config $c --steps 50 --warmup 10 2> gpurun_out/r04_bench$sfx.err | tail -1 > gpurun_out/r04_bench$sfx.json || { tail -5 gpurun_out/r04_bench$sfx.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04_bench$sfx.json').read())
print('$c', d['value'], d['unit'], d['ms_per_step'], 'gemm_nt', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], 'clock', (d.get('clock_mhz_during_timed_steps') or {}).get('median'))
PY
done
