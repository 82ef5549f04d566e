#!/bin/bash
# four tile-height policies of dclip_gemm_nt over the step's (shape, epilogue) combinations; prints the table and the count-weighted totals
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for m in model 0 2 6; do
  if [ $m = model ]; then env -u DCLIP_GEMM320 python3 tools/diag/gemm_tile_sweep.py 2>/dev/null | tail -1 > gpurun_out/tile_sweep_$m.json
  else DCLIP_GEMM320=$m python3 tools/diag/gemm_tile_sweep.py 2>/dev/null | tail -1 > gpurun_out/tile_sweep_$m.json; fi
  echo "[tile sweep] policy $m done"
done
python3 - <<'PY'
import json
res = {m: json.load(open(f'gpurun_out/tile_sweep_{m}.json'))['combos'] for m in ('model', '0', '2', '6')}
tot = {m: 0.0 for m in res}
best_tot = 0.0
print(f"{'combination':32s} {'n':>3s} {'model':>8s} {'256':>8s} {'320':>8s} {'192':>8s}  best")
for k, v in res['model'].items():
    t = {m: res[m][k]['us'] for m in res}
    n = v['per_step']
    for m in res:
        tot[m] += n * t[m]
    b = min(('0', '2', '6'), key=lambda m: t[m])
    best_tot += n * min(t[b], t['model'])
    flag = '' if t['model'] <= 1.02 * t[b] else f"  <-- {({'0': 256, '2': 320, '6': 192})[b]} rows is {100 * (t['model'] / t[b] - 1):.0f} % faster"
    print(f"{k:32s} {n:3d} {t['model']:8.1f} {t['0']:8.1f} {t['2']:8.1f} {t['6']:8.1f}{flag}")
print('count-weighted ms per step:', {m: round(v / 1e3, 3) for m, v in tot.items()}, 'per-combination best:', round(best_tot / 1e3, 3))
PY
