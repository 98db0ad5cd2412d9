#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05e; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for dt in bf16 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$dt -o dp -- python $R/bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-events --force-collectives --bucket-mb 0 > /dev/null 2> /tmp/err_$dt.txt
  f=/tmp/prof_$dt/dp_kernel_stats.csv
  ls /tmp/prof_$dt | head; tail -3 /tmp/err_$dt.txt
  echo "== $dt"; head -1 "$f"; grep -i -E "nccl|rccl|AllReduce|memcpy|copy|momentum|elementwise" "$f" | head -10
  cp "$f" "$O/dp_single_${dt}_kernel_stats.csv"
done
