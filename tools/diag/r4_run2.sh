#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_towers_gpu.py tests/test_configs_gpu.py tests/test_trajectory_gpu.py tests/test_checkpoint_gpu.py tests/test_fullsize_gpu.py -x -q -s > gpurun_out/r4_towers.log 2>&1 || { tail -40 gpurun_out/r4_towers.log; exit 1; }
grep -E "parity|passed|failed" gpurun_out/r4_towers.log | cut -c1-1500
for cfg in "0" "1" "2"; do
  DCLIP_GEMM_DUO=$cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_bench_duo$cfg.json 2> gpurun_out/r4_bench_duo$cfg.err || { tail -20 gpurun_out/r4_bench_duo$cfg.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_bench_duo$cfg.json').read().strip().splitlines()[-1])
print('DUO=$cfg', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'clock', d.get('clock_mhz_during_timed_steps'))
PY
done
