#!/bin/bash
# rocprofv3 --kernel-trace --stats of the plain-CLIP-student step (tools/diag/clip_student_step.py), towers on one stream as in tools/run_profiles.sh
# (co-running kernels stretch each other's durations in a trace).  Run on the GPU box from the repo root; writes gpurun_out/prof_clipstu{,_hidden}.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
export GPU_MAX_HW_QUEUES=8 DCLIP_MULTI_STREAM=0
for v in "" "--hidden"; do
  tag=clipstu${v:+_hidden}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 tools/diag/clip_student_step.py $v --steps 5 --warmup 2 > gpurun_out/prof_$tag.log 2>&1 || exit 1
  tail -1 gpurun_out/prof_$tag.log
done
