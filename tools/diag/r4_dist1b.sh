#!/bin/bash
set -o pipefail
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
timeout -k 10 600 python -m pytest tests/test_parallel_gpu.py tests/test_checkpoint_gpu.py -x -q > gpurun_out/r4_par_tests.log 2>&1 || { tail -30 gpurun_out/r4_par_tests.log; exit 1; }
tail -2 gpurun_out/r4_par_tests.log
for i in 1 2; do
( timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | tail -1 ) > gpurun_out/r4_plain$i.json
( DCLIP_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2> gpurun_out/r4_dist1_$i.err | tail -1 ) > gpurun_out/r4_dist1_$i.json
python - <<PY
import json
a=json.loads(open('gpurun_out/r4_plain$i.json').read()); b=json.loads(open('gpurun_out/r4_dist1_$i.json').read())
print('plain', a['value'], a['ms_per_step'], '| forced dist world 1', b['value'], b['ms_per_step'], b['config']['gradient_exchange'], 'loss', a['final_loss'], b['final_loss'])
PY
done
