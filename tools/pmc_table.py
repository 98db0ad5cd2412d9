#!/usr/bin/env python3
"""per-kernel averages of one rocprofv3 --pmc counter: pmc_table.py <COUNTER> <counter_collection.csv>"""
import collections
import csv
import sys

name, path = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    if r['Counter_Name'] != name:
        continue
    k = r['Kernel_Name'][:90]
    tot[k] += float(r['Counter_Value']); cnt[k] += 1
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f'{k}\t{name}\tlaunches={cnt[k]}\tavg={tot[k] / cnt[k]:.4g}\tsum={tot[k]:.4g}')
