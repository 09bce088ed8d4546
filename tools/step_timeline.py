"""Timeline of ONE training step from a rocprofv3 kernel trace (kernel_trace.csv): kernels in start order with queue,
start offset and duration, the idle time of the whole GPU inside the step, and per-kernel-name totals.
usage: python3 tools/step_timeline.py <kernel_trace.csv> [step_index_from_end=1] [--full]"""
import csv, sys, collections


def short(n):
    return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows))
ends = [i for i, k in enumerate(ks) if 'clip_sgd' in k[2]]
hi = ends[-back]
lo = ends[-back - 1] + 1
step = ks[lo:hi + 1]
t0 = step[0][0]
span = (step[-1][1] - t0) / 1e3
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('step span %.1f us, %d kernels, GPU idle (no kernel on any queue) %.1f us' % (span, len(step), span - busy / 1e3))
tot = collections.OrderedDict()
for s, e, n, q in step:
    key = (q, short(n))
    d = tot.setdefault(key, [0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
for (q, n), (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print('q%-3s %-62s x%-3d %9.1f us' % (q, n, c, d))
if '--full' in sys.argv:
    prev_end = {}
    for s, e, n, q in step:
        gap = (s - prev_end.get(q, s)) / 1e3
        prev_end[q] = e
        print('%9.1f q%-3s %8.1f (gap %6.1f) %s' % ((s - t0) / 1e3, q, (e - s) / 1e3, gap, short(n)))
