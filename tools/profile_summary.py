#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/prof_<tag>_{stats,fetch,write}) into profiles/<tag>_*.

    python tools/profile_summary.py r01

Writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --stats table), profiles/<tag>_summary.md and
profiles/<tag>_traffic.json (per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE is doubled
as MI355X_MICROARCH.md §HBM prescribes for gfx950 wide coalesced reads; counter unit = KiB)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6          # steps + warmup (+2 probe steps) executed under --stats
title = sys.argv[3] if len(sys.argv) > 3 else 'l_clip dual step, B=512, 1x MI355X'


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0]


def family(n):
    n = short(n)
    if n.startswith('gemm_nt'):
        return 'gemm_nt'
    return re.sub(r'<.*', '', n)


out = os.path.join(ROOT, 'profiles')
os.makedirs(out, exist_ok=True)
def newest(pattern):
    """gpurun merges every call's outputs into the same directory: take the latest run's file"""
    fs = sorted(glob.glob(os.path.join(ROOT, pattern)), key=os.path.getmtime)
    return fs[-1:]


stats = newest(f'gpurun_out/prof_{tag}_stats/*/*_kernel_stats.csv')[0]
shutil.copy(stats, os.path.join(out, f'{tag}_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
fam = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    f = family(r['Name'])
    fam[f][0] += int(r['Calls'])
    fam[f][1] += float(r['TotalDurationNs'])

traffic = {}
for which, mult in (('fetch', 2.0), ('write', 1.0)):
    fs = newest(f'gpurun_out/prof_{tag}_{which}/*/*_counter_collection.csv')
    if not fs:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        f = family(r['Kernel_Name'])
        agg[f][0] += 1
        agg[f][1] += float(r['Counter_Value']) * 1024.0 * mult
    for f, (c, v) in agg.items():
        traffic.setdefault(f, {})[which + '_bytes_per_launch'] = v / c
        traffic[f]['launches_' + which] = c
for f, d in traffic.items():
    d['hbm_bytes_per_launch'] = d.get('fetch_bytes_per_launch', 0.0) + d.get('write_bytes_per_launch', 0.0)
json.dump(traffic, open(os.path.join(out, f'{tag}_traffic.json'), 'w'), indent=1, sort_keys=True)

# MFMA utilisation per family: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
mfma = {}
fs = newest(f'gpurun_out/prof_{tag}_mfma/*/*_counter_collection.csv')
if fs:
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        per[(r['Dispatch_Id'], family(r['Kernel_Name']))][r['Counter_Name']] = float(r['Counter_Value'])
    agg = collections.defaultdict(lambda: [0.0, 0.0])
    for (_, f), c in per.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
            agg[f][0] += c['SQ_VALU_MFMA_BUSY_CYCLES']
            agg[f][1] += c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0
    mfma = {f: v[0] / v[1] for f, v in agg.items() if v[1] > 0 and v[0] > 0}
    json.dump(mfma, open(os.path.join(out, f'{tag}_mfma_util.json'), 'w'), indent=1, sort_keys=True)

with open(os.path.join(out, f'{tag}_summary.md'), 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats, bench.py ({title}) — {tag}\n\n')
    f.write(f'total GPU kernel time {tot / 1e6:.1f} ms over the run ({steps} steps incl. warm-up / probe)\n\n')
    f.write('| kernel family | launches | total ms | avg us | % |\n|---|---|---|---|---|\n')
    for k, (c, v) in sorted(fam.items(), key=lambda x: -x[1][1])[:24]:
        f.write(f'| {k} | {c} | {v / 1e6:.2f} | {v / c / 1e3:.1f} | {100 * v / tot:.1f} |\n')
    f.write('\n## per kernel (top 30)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n')
    for r in rows[:30]:
        f.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
    if traffic:
        f.write('\n## HBM traffic per launch (PMC passes; 2*FETCH_SIZE + WRITE_SIZE, KiB counters)\n\n| family | MB / launch |\n|---|---|\n')
        for k, d in sorted(traffic.items(), key=lambda x: -x[1]['hbm_bytes_per_launch'] * x[1].get('launches_fetch', 1))[:12]:
            f.write(f"| {k} | {d['hbm_bytes_per_launch'] / 1e6:.1f} |\n")
if mfma:
    with open(os.path.join(out, f'{tag}_summary.md'), 'a') as f:
        f.write('\n## MFMA utilisation (PMC pass: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8))\n\n| family | MFMA pipe busy |\n|---|---|\n')
        for k, v in sorted(mfma.items(), key=lambda x: -x[1]):
            f.write(f'| {k} | {100 * v:.1f} % |\n')
print('wrote profiles/', tag)
