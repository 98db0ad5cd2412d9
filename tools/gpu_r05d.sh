#!/bin/bash
# data-parallel plumbing on a single-rank RCCL group: plain step vs bucketed (44 MB = 3 buckets, 16 MB = 7) vs one all-reduce
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05d; mkdir -p "$O"; cd "$R"
run() { python bench.py --dtype $1 --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-kernel-events "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f img/s %7.3f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  for dt in bf16 f32; do
    printf "%-50s " "$dt rep$rep plain"; run $dt
    printf "%-50s " "$dt rep$rep force-collectives bucket 44 MB"; run $dt --force-collectives --bucket-mb 44
    printf "%-50s " "$dt rep$rep force-collectives bucket 16 MB"; run $dt --force-collectives --bucket-mb 16
    printf "%-50s " "$dt rep$rep force-collectives single all-reduce"; run $dt --force-collectives --bucket-mb 0
  done
done | tee "$O/dp_plumbing.txt"
timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer > /dev/null 2> "$O/per_layer_bf16.txt"
