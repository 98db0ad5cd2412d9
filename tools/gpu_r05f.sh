#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05f; mkdir -p "$O"; cd "$R"
run() { python bench.py --dtype $1 --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-kernel-events "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f img/s %7.3f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  for dt in bf16 f32; do
    printf "%-60s " "$dt rep$rep plain"; run $dt
    printf "%-60s " "$dt rep$rep null collective, bucket 44 MB"; run $dt --force-collectives --null-collective --bucket-mb 44
    printf "%-60s " "$dt rep$rep null collective, bucket 16 MB"; run $dt --force-collectives --null-collective --bucket-mb 16
    printf "%-60s " "$dt rep$rep null collective, single"; run $dt --force-collectives --null-collective --bucket-mb 0
    printf "%-60s " "$dt rep$rep RCCL single-rank, bucket 44 MB"; run $dt --force-collectives --bucket-mb 44
  done
done | tee "$O/dp_plumbing_null.txt"
