#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
export GPU_MAX_HW_QUEUES=8
rm -rf gpurun_out/prof_r04_timeline
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r04_timeline -- python3 bench.py --no-cpu-baseline --no-roofline --no-clock-probe --steps 6 --warmup 3 > gpurun_out/prof_r04_timeline.log 2>&1 || { tail -20 gpurun_out/prof_r04_timeline.log; exit 1; }
f=$(ls -S gpurun_out/prof_r04_timeline/*/*kernel_trace.csv | head -1)
python3 tools/diag/timeline_gaps.py $f 10 | tee gpurun_out/r4_timeline.txt
