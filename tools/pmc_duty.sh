#!/bin/bash
# Matrix-pipe duty and shader clock per kernel of a bench.py step (GPU box):  tools/pmc_duty.sh <tag> [bench flags]
# duty = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); clock = SQ_BUSY_CU_CYCLES / 256 CUs / kernel duration
TAG=${1:-duty}; shift || true
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/pd -o d -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-kernel-events "$@" > /dev/null 2>&1
python - /tmp/pd/d_counter_collection.csv /tmp/pd/d_kernel_trace.csv > $O/pmc_duty.txt <<'PY'
import csv, sys, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:80]
    cnt[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_BUSY_CU_CYCLES': n[k] += 1
dur = collections.defaultdict(float)
try:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[r['Kernel_Name'][:80]] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
except Exception as e:
    pass
rows = []
for k, c in cnt.items():
    busy, mf = c.get('SQ_BUSY_CU_CYCLES', 0), c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    if busy <= 0: continue
    ghz = busy / 256.0 / dur[k] if dur.get(k) else float('nan')
    rows.append((dur.get(k, 0), k, n[k], mf / (4 * busy), ghz))
print('%-82s %6s %10s %8s %8s' % ('kernel', 'calls', 'total_us', 'duty', 'GHz'))
for d, k, c, duty, ghz in sorted(rows, reverse=True):
    print('%-82s %6d %10.1f %8.3f %8.2f' % (k, c, d / 1000.0, duty, ghz))
PY
head -30 $O/pmc_duty.txt
