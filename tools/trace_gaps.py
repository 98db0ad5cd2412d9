#!/usr/bin/env python3
"""How much of a step is the GPU idle?  Reads a rocprofv3 --kernel-trace CSV of a bench.py run, cuts it into steps at the
optimizer kernel, and prints per step: span, union of the kernels' busy intervals, idle time, the largest gaps and the kernels
on either side.   tools/trace_gaps.py <kernel_trace.csv> [delimiter-kernel-substring]"""
import csv
import sys


def main():
    path = sys.argv[1]
    delim = sys.argv[2] if len(sys.argv) > 2 else 'momentum_kernel'
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if delim in r[2]]
    if len(cuts) < 4:
        print('fewer than 4 steps in the trace'); return
    steps = [(cuts[i] + 1, cuts[i + 1] + 1) for i in range(len(cuts) - 1)]
    steps = steps[len(steps) // 2:]                      # the timed half (after the warmup)
    tot_span = tot_busy = 0
    gaps = []
    for a, b in steps:
        seg = rows[a:b]
        t0 = rows[a - 1][1]                              # end of the previous optimizer kernel
        t1 = seg[-1][1]
        busy, cur_s, cur_e, last_name = 0, t0, t0, rows[a - 1][2]
        for s, e, name in seg:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, last_name, name))
                cur_s, cur_e = s, e
            if e >= cur_e:
                cur_e, last_name = e, name
        busy += cur_e - cur_s
        tot_span += t1 - t0; tot_busy += busy
    n = len(steps)
    print(f'{n} steps: span {tot_span / n / 1e3:.1f} us/step, some kernel running {tot_busy / n / 1e3:.1f} us/step, '
          f'idle {(tot_span - tot_busy) / n / 1e3:.1f} us/step ({100.0 * (tot_span - tot_busy) / tot_span:.1f} %), {len(rows[steps[0][0]:steps[-1][1]]) // n} launches/step')
    agg = {}
    for g, before, after in gaps:
        k = (before.split('(')[0][:60], after.split('(')[0][:60])
        c = agg.setdefault(k, [0, 0]); c[0] += g; c[1] += 1
    print('largest idle gaps (us/step, count/step, kernel before -> kernel after):')
    for k, (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:15]:
        print(f'  {g / n / 1e3:8.1f} {c / n:6.1f}   {k[0]}  ->  {k[1]}')


if __name__ == '__main__':
    main()
