#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_* (run on the GPU box from the repo root: bash tools/run_profiles.sh r03).
# Counters are collected in their own passes (--pmc never together with the trace domains beyond --kernel-trace).
set -o pipefail
TAG=${1:-r03}
CONFIGS=${2:-"default image text lclip336"}     # subset per call when the box time is limited (gpurun caps a call at 20 min)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8        # read at HIP initialisation, which the profiler's preload performs before python starts
# the four towers normally run on four streams: co-running kernels stretch each other's durations in a kernel trace, so the
# profiled passes put every tower on one stream (same kernels, same launches; the multi-stream step time is bench.py's number)
export DCLIP_MULTI_STREAM=0
OUT=gpurun_out
B="python3 bench.py --no-cpu-baseline"
for c in $CONFIGS; do
  if [ $c = default ]; then sfx=""; args=""; else sfx="_$c"; args="--config $c"; fi
  echo "[profiles] stats $c"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}${sfx}_stats -- $B $args --steps 5 --warmup 3 > $OUT/prof_${TAG}${sfx}_stats.log 2>&1 || exit 1
done
# counter passes, one counter set per run (never combined with the other trace domains), for the default workload and per --config
pmc() {   # pmc <suffix> <config args...>
  local sfx=$1; shift
  echo "[profiles] pmc fetch $sfx"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}${sfx}_fetch -- $B "$@" --steps 1 --warmup 1 --no-roofline > $OUT/prof_${TAG}${sfx}_fetch.log 2>&1 || return 1
  echo "[profiles] pmc write $sfx"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}${sfx}_write -- $B "$@" --steps 1 --warmup 1 --no-roofline > $OUT/prof_${TAG}${sfx}_write.log 2>&1 || return 1
  echo "[profiles] pmc mfma $sfx"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/prof_${TAG}${sfx}_mfma -- $B "$@" --steps 1 --warmup 1 --no-roofline > $OUT/prof_${TAG}${sfx}_mfma.log 2>&1 || return 1
}
for c in $CONFIGS; do
  if [ $c = default ]; then pmc "" || exit 1; else pmc "_$c" --config $c || exit 1; fi
done
echo "[profiles] done"
