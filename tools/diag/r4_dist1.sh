#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
export GPU_MAX_HW_QUEUES=8 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for i in 1 2; do
( timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | tail -1 ) > gpurun_out/r4_plain$i.json
( DCLIP_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2> gpurun_out/r4_dist1_$i.err | tail -1 ) > gpurun_out/r4_dist1_$i.json
python - <<PY
import json
a=json.loads(open('gpurun_out/r4_plain$i.json').read()); b=json.loads(open('gpurun_out/r4_dist1_$i.json').read())
print('plain', a['value'], a['ms_per_step'], '| forced dist world 1', b['value'], b['ms_per_step'], b['config']['gradient_exchange'])
PY
done
rm -rf gpurun_out/prof_r04_dist1
DCLIP_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_dist1 -- python3 bench.py --no-cpu-baseline --no-roofline --no-clock-probe --steps 6 --warmup 3 > gpurun_out/prof_r04_dist1.log 2>&1 || { tail -20 gpurun_out/prof_r04_dist1.log; exit 1; }
f=$(ls -S gpurun_out/prof_r04_dist1/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$f')))
for r in rows[:45]:
    n=r['Name'].replace('void ','').replace('(anonymous namespace)::','')[:70]
    print(f"{n:70s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
