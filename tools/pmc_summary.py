"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files:  python tools/pmc_summary.py <csv> [kernel-substring]"""
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
sub = sys.argv[2] if len(sys.argv) > 2 else ''
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:80]
    if sub not in k:
        continue
    tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in tot:
    print(k)
    for c, v in sorted(tot[k].items()):
        print(f'    {c:<34s} {v / cnt[(k, c)]:16.0f}   (n={cnt[(k, c)]})')
