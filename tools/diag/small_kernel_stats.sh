cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8 DCLIP_MULTI_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 --no-roofline > gpurun_out/prof_small.log 2>&1
python3 - <<'PY'
import csv,glob,os
f=sorted(glob.glob('gpurun_out/prof_small/*/*_kernel_stats.csv'), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('pick_index','batch_sum','embed_','colsum8','im2row','token_table')):
        print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
