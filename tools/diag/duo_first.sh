#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
DCLIP_GEMM_DUO=2 DCLIP_DUO_MI=8 timeout -k 10 400 python -m pytest tests/test_gemm_gpu.py -x -q -k "not random_shapes and not tn" > gpurun_out/duo_t8.log 2>&1 || { tail -30 gpurun_out/duo_t8.log; exit 1; }
tail -2 gpurun_out/duo_t8.log
DCLIP_GEMM_DUO=2 DCLIP_DUO_MI=10 timeout -k 10 400 python -m pytest tests/test_gemm_gpu.py -x -q -k "not random_shapes and not tn" > gpurun_out/duo_t10.log 2>&1 || { tail -30 gpurun_out/duo_t10.log; exit 1; }
tail -2 gpurun_out/duo_t10.log
timeout -k 10 600 tools/diag/duo_ab.sh || exit 1
DCLIP_GEMM_DUO=2 timeout -k 10 200 python tools/diag/duo_phases.py > gpurun_out/duo_phases.log 2>&1 || { tail -20 gpurun_out/duo_phases.log; exit 1; }
cat gpurun_out/duo_phases.log
DCLIP_GEMM_DUO=2 timeout -k 10 300 python tools/diag/step_shapes.py > gpurun_out/duo_step_shapes.log 2>&1 || { tail -20 gpurun_out/duo_step_shapes.log; exit 1; }
tail -3 gpurun_out/duo_step_shapes.log
