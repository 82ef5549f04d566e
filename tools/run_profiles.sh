#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_* (run on the GPU box from the repo root: bash tools/run_profiles.sh r02).
# Counters are collected in their own passes (--pmc never together with the trace domains beyond --kernel-trace).
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8        # read at HIP initialisation, which the profiler's preload performs before python starts
# the four towers normally run on four streams: co-running kernels stretch each other's durations in a kernel trace, so the
# profiled passes put every tower on one stream (same kernels, same launches; the multi-stream step time is bench.py's number)
export DCLIP_MULTI_STREAM=0
OUT=gpurun_out
B="python3 bench.py --no-cpu-baseline"
echo "[profiles] stats lclip"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -- $B --steps 5 --warmup 3 > $OUT/prof_${TAG}_stats.log 2>&1 || exit 1
for c in image text lclip336; do
  echo "[profiles] stats $c"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_${c}_stats -- $B --config $c --steps 5 --warmup 3 > $OUT/prof_${TAG}_${c}_stats.log 2>&1 || exit 1
done
echo "[profiles] pmc fetch"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -- $B --steps 1 --warmup 1 --no-roofline > $OUT/prof_${TAG}_fetch.log 2>&1 || exit 1
echo "[profiles] pmc write"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -- $B --steps 1 --warmup 1 --no-roofline > $OUT/prof_${TAG}_write.log 2>&1 || exit 1
echo "[profiles] pmc mfma"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_mfma -- $B --steps 1 --warmup 1 --no-roofline > $OUT/prof_${TAG}_mfma.log 2>&1 || exit 1
echo "[profiles] done"
