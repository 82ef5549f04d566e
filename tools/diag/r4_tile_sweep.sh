#!/bin/bash
# per-(shape, epilogue) time of the step's gemm_nt launches under every tile height against the launcher's own choice
set -o pipefail
mkdir -p gpurun_out
for m in 1 0 2 6; do
  echo "== DCLIP_GEMM320=$m" | tee -a gpurun_out/r4_tile_sweep.log
  DCLIP_GEMM320=$m timeout -k 10 250 python tools/diag/gemm_duo_ab.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_tile_sweep.log || exit 1
done
