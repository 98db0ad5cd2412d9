#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05g; mkdir -p "$O"; cd "$R"
run() { python bench.py --dtype $1 --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-kernel-events "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f img/s %7.3f ms' % (d['value'], d['ms_per_step']))"; }
{
echo "# one MI355X; bench.py --force-collectives = the data-parallel step on a single-rank RCCL group (staged backward, bucketed all-reduce behind the weight-gradient stream, update after the last bucket)"
for rep in 1 2; do
  for dt in bf16 f32; do
    printf "%-64s " "$dt rep$rep plain step"; run $dt
    printf "%-64s " "$dt rep$rep DP step, 44 MB buckets (8 HW queues reserved: default)"; run $dt --force-collectives --bucket-mb 44
    printf "%-64s " "$dt rep$rep DP step, 16 MB buckets"; run $dt --force-collectives --bucket-mb 16
    printf "%-64s " "$dt rep$rep DP step, one all-reduce after backward"; run $dt --force-collectives --bucket-mb 0
    printf "%-64s " "$dt rep$rep DP step, 44 MB buckets, GPU_MAX_HW_QUEUES=4 (round 4)"; GPU_MAX_HW_QUEUES=4 run $dt --force-collectives --bucket-mb 44
    printf "%-64s " "$dt rep$rep DP step, 44 MB buckets, bf16 messages"; run $dt --force-collectives --bucket-mb 44 --allreduce-dtype bf16
  done
done
} | tee "$O/dp_hw_queues.txt"
