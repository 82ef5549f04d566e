#!/bin/bash
# lib_a = HEAD, lib_b = HEAD compiled with -DDCLIP_NT256_NO_PERSIST (the tile loop compiled out of every nt256 variant: the round-3 kernel structure)
set -o pipefail
mkdir -p gpurun_out
cp distillclip_amd/libdistillclip_hip.so /tmp/lib_ship.so
run() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'gemm_nt ms', r['others']['gemm_nt_kernel']['ms_per_step'], 'frac', r['frac'])"; }
for rep in 1 2; do
  for v in a b; do cp tools/diag/bin/lib_$v.so distillclip_amd/libdistillclip_hip.so; echo "== lib_$v"; run || { cp /tmp/lib_ship.so distillclip_amd/libdistillclip_hip.so; exit 1; }; done
done
cp tools/diag/bin/lib_a.so distillclip_amd/libdistillclip_hip.so; echo "== lib_a DCLIP_GEMM_PERSIST=0"; DCLIP_GEMM_PERSIST=0 run
cp /tmp/lib_ship.so distillclip_amd/libdistillclip_hip.so
