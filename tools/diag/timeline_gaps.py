"""GPU idle time inside the four-stream step, from a rocprofv3 --kernel-trace CSV of bench.py: union of the kernels' [start, end] intervals over
the last full steps (delimited by the adamw4 launches), idle gaps longer than a threshold with the kernels on either side, and how many
kernels run concurrently over time.    python tools/diag/timeline_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda e: e[0])
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:48]
# steps: the first adamw4 launch of every step (two per step, one per tower): take every second one
ad = [i for i, e in enumerate(ev) if 'adamw4' in e[2]]
if len(ad) < 8:
    raise SystemExit('need a few steps in the trace')
bounds = [ev[i][0] for i in ad[::2]]
for si in range(len(bounds) - 4, len(bounds) - 1):
    t0, t1 = bounds[si], bounds[si + 1]
    seg = [e for e in ev if t0 <= e[0] < t1]
    busy, cur_s, cur_e, gaps = 0, None, None, []
    last_name = ''
    for s, e, n in seg:
        if cur_e is None:
            cur_s, cur_e, last_name = s, e, n
            continue
        if s > cur_e:
            busy += cur_e - cur_s
            if (s - cur_e) / 1e3 >= min_gap:
                gaps.append(((s - cur_e) / 1e3, (cur_e - t0) / 1e6, short(last_name), short(n)))
            cur_s, cur_e, last_name = s, e, n
        else:
            if e > cur_e:
                cur_e, last_name = e, n
    busy += cur_e - cur_s
    # concurrency histogram
    pts = sorted([(s, 1) for s, e, n in seg] + [(e, -1) for s, e, n in seg])
    lvl, prev, hist = 0, t0, collections.Counter()
    for t, d in pts:
        hist[lvl] += t - prev
        prev, lvl = t, lvl + d
    tot = t1 - t0
    print(f'step {si}: {tot / 1e6:.3f} ms, {len(seg)} kernels, GPU busy (union) {busy / 1e6:.3f} ms, idle {100 * (1 - busy / tot):.1f} % ; time with k kernels running: '
          + ', '.join(f'{k}: {100 * v / tot:.0f}%' for k, v in sorted(hist.items())))
    for g in sorted(gaps, reverse=True)[:12]:
        print(f'    idle {g[0]:7.1f} us at +{g[1]:6.2f} ms  after {g[2]}  before {g[3]}')
