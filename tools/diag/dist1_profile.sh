cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
export DCLIP_FORCE_DIST=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_dist1 -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > gpurun_out/prof_r03_dist1.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_r03_dist1/*/*_kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms',tot/1e6)
for r in rows[:40]:
    print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1))
PY
tail -2 gpurun_out/prof_r03_dist1.log | cut -c1-300
