#!/usr/bin/env python3
"""What runs beside what?  Reads a rocprofv3 --kernel-trace CSV of a bench.py run, cuts it into steps at the optimizer
kernel, takes one step of the timed half and prints every launch of it in start order: offset from the step's start, duration,
hardware queue, workgroups, kernel.  Then a summary of the step by "how full is the chip": the time during which the launches
in flight together hold fewer than 256 workgroups (the chip has 256 CUs), by which kernels were running then.
    tools/timeline.py <kernel_trace.csv> [delimiter-kernel-substring] [step index from the end, default 2]"""
import csv
import sys


def short(name):
    n = name.replace('void ', '').replace('ssd::', '')
    return n.split('(')[0][:70]


def main():
    path = sys.argv[1]
    delim = sys.argv[2] if len(sys.argv) > 2 else 'momentum_kernel'
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            wg = 1
            for ax in 'XYZ':
                g = int(r.get('Grid_Size_' + ax, r.get('Grid_Size', 1)) or 1)
                w = int(r.get('Workgroup_Size_' + ax, r.get('Workgroup_Size', 1)) or 1)
                wg *= max(1, g // max(1, w))
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), wg))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if delim in r[2]]
    if len(cuts) < 4:
        print('fewer than 4 steps in the trace'); return
    a, b = cuts[-back - 1] + 1, cuts[-back] + 1
    seg = rows[a:b]
    t0 = rows[a - 1][1]
    queues = sorted({r[3] for r in seg})
    print(f'step of {len(seg)} launches, span {(seg[-1][1] - t0) / 1e3:.1f} us, queues {queues}')
    print('  start_us   dur_us  q     wgs  kernel')
    for s, e, name, q, wg in seg:
        print(f'{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {queues.index(q)} {wg:7d}  {short(name)}')
    # sweep: at every instant the set of running launches
    ev = []
    for i, (s, e, name, q, wg) in enumerate(seg):
        ev.append((s, 1, i)); ev.append((e, 0, i))
    ev.sort()
    live = set()
    under = {}
    t_prev = t0
    under_total = 0
    for t, kind, i in ev:
        if t > t_prev:
            wgs = sum(seg[j][4] for j in live)
            if wgs < 256:
                key = ' + '.join(sorted({short(seg[j][2])[:40] for j in live})) or '(idle)'
                under[key] = under.get(key, 0) + (t - t_prev)
                under_total += t - t_prev
        t_prev = t
        if kind:
            live.add(i)
        else:
            live.discard(i)
    print(f'\ntime with fewer than 256 workgroups in flight: {under_total / 1e3:.1f} us of {(seg[-1][1] - t0) / 1e3:.1f}')
    for k, v in sorted(under.items(), key=lambda kv: -kv[1])[:40]:
        print(f'{v / 1e3:9.1f} us  {k}')


if __name__ == '__main__':
    main()
