#!/bin/bash
# same-box A/B of the duo GEMM selection: isolated combos, then the in-situ single-stream step table
set -o pipefail
out=gpurun_out
mkdir -p $out
for cfg in "0 0 1" "2 0 1" "2 8 1" "2 10 1" "2 0 0"; do
  set -- $cfg
  DCLIP_GEMM_DUO=$1 DCLIP_DUO_MI=$2 DCLIP_DUO_PRIO=$3 python tools/diag/gemm_duo_ab.py > $out/duo_ab_$1_$2_$3.log 2>&1 || exit 1
  tail -1 $out/duo_ab_$1_$2_$3.log
done
