#!/usr/bin/env python
"""Where the chip idles inside a training step.

    rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python3 bench.py --steps 12 --warmup 4 \
        --no-cpu-baseline --no-extras
    python tools/step_idle_gaps.py gpurun_out/tl/*/tl_kernel_trace.csv

Reads the per-dispatch trace, finds the optimiser kernel (clip_sgd_kernel: once per step, last on the chain), and for
every step between two of them prints: wall time from the previous step's last kernel to this step's last kernel, the time
at least one kernel was resident (union over streams), and the largest idle gaps with the kernels on either side.  Idle
time is host-bound (launch queue ran dry) or a cross-stream wait; everything else is the chain.
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0][:60]


def main(path, top=8):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if 'clip_sgd_kernel' in r[2]]
    if len(ends) < 3:
        raise SystemExit('fewer than three steps in the trace')
    per_gap = defaultdict(list)
    walls, busys, all_gaps = [], [], []
    for a, b in zip(ends[:-1], ends[1:]):
        seg = rows[a:b + 1]                     # previous step's optimiser kernel .. this step's
        wall = seg[-1][1] - seg[0][1]
        busy, gaps = 0, []
        cur_end, cur_name = seg[0][1], seg[0][2]
        for s0, e0, n0 in seg[1:]:
            if s0 > cur_end:
                gaps.append((s0 - cur_end, short(cur_name), short(n0)))
                busy += e0 - s0
                cur_end, cur_name = e0, n0
            elif e0 > cur_end:
                busy += e0 - cur_end
                cur_end, cur_name = e0, n0
        walls.append(wall)
        busys.append(busy)
        all_gaps.append(gaps)
    med = sorted(walls)[len(walls) // 2]
    print('per-step wall ms:', ' '.join('%.2f' % (w / 1e6) for w in walls))
    keep = [i for i, w in enumerate(walls) if w < 1.25 * med]      # drop warm-up / leg boundaries (allocation, host setup)
    per_gap = defaultdict(list)
    for i in keep:
        for g, p, nx in all_gaps[i]:
            per_gap[(p, nx)].append(g)
    walls, busys = [walls[i] for i in keep], [busys[i] for i in keep]
    n = len(walls)
    print('steps: %d   wall %.3f ms   resident %.3f ms   idle %.3f ms (averages)' % (
        n, sum(walls) / n / 1e6, sum(busys) / n / 1e6, (sum(walls) - sum(busys)) / n / 1e6))
    agg = sorted(((sum(v) / n, len(v) / n, k) for k, v in per_gap.items()), reverse=True)
    print('largest idle gaps (us per step, occurrences per step, kernel before -> kernel after):')
    for us, cnt, (p, nx) in agg[:top * 3]:
        print('  %8.1f  %5.2f  %s -> %s' % (us / 1e3, cnt, p, nx))


if __name__ == '__main__':
    main(sys.argv[1])
