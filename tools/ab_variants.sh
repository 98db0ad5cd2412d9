#!/bin/bash
# Interleaved A/B of step variants on ONE box:  tools/ab_variants.sh <out.txt> <rounds> <dtype> "<name>:<ENV=.. ENV=..>" ...
# Every round runs every variant once (tools/step_time.py, 40 steps x 2); prints the per-variant minimum and median.
OUT=$1; ROUNDS=$2; DT=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
STEPS=40; [ "$DT" = f32 ] && STEPS=10
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs python tools/step_time.py --dtype $DT --steps $STEPS --reps 2 --tag $name 2>/dev/null >> $OUT
  done
done
python - $OUT <<'PY'
import sys, collections, statistics
d = collections.OrderedDict()
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) >= 5 and p[-1] == 'ms/step':
        d.setdefault((p[0], p[1]), []).extend(float(x) for x in p[2:-1])
for (k, dt), v in d.items():
    print('%-28s %s min %.3f  median %.3f  n=%d' % (k, dt, min(v), statistics.median(v), len(v)))
PY
