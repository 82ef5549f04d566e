"""how many kernels run concurrently during one step: reads the kernel trace of
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ms -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline
(multi-stream default).  Found the hardware-queue sharing that serialised the two student backwards."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import csv, glob, re, collections
f = glob.glob(os.path.join(ROOT, 'gpurun_out/prof_ms/*/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0][:60]
    wgs = (int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])) // max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, wgs))
ev.sort()
ad = [i for i, e in enumerate(ev) if e[2].startswith('adamw')]
lo, hi = ev[ad[-3]][1], ev[ad[-1]][1]
seg = sorted(e for e in ev if e[0] >= lo and e[1] <= hi)
print('step span ms', (hi - lo) / 1e6)
pts = sorted(set([e[0] for e in seg] + [e[1] for e in seg]))
solo = collections.Counter(); nrun = collections.Counter()
for a, b in zip(pts, pts[1:]):
    run = [e for e in seg if e[0] <= a and e[1] >= b]
    dt = (b - a) / 1e3
    nrun[len(run)] += dt
    if len(run) == 1: solo[(run[0][2], run[0][3] >= 240)] += dt
print('time by number of concurrently running kernels (us):', {k: round(v) for k, v in sorted(nrun.items())})
print('solo time by kernel:')
for (n, big), t in solo.most_common(18): print(f'   {t:8.1f} us  {"full-chip" if big else "SMALL    "}  {n}')
