#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -x -q -k "tn or random_shapes" > gpurun_out/r4_tn_tests.log 2>&1 || { tail -30 gpurun_out/r4_tn_tests.log; exit 1; }
tail -2 gpurun_out/r4_tn_tests.log
timeout -k 10 200 python tools/diag/gemm_tn_phases.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
( timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_bench_tn$i.json || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_bench_tn$i.json').read().strip().splitlines()[-1])
print('HEAD', d['value'], d['ms_per_step'], 'gemm_nt frac', d['roofline']['frac'], {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
( cd _r3 && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 ) > gpurun_out/r4_bench_r3$i.json || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_bench_r3$i.json').read().strip().splitlines()[-1])
print('r3  ', d['value'], d['ms_per_step'], 'gemm_nt frac', d['roofline']['frac'], {k: v['ms_per_step'] for k, v in d['roofline']['others'].items()})
PY
done
