#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06aa
O=$R/gpurun_out/$TAG; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o e -- \
    python "$R/bench.py" --mode train_e2e --dtype bf16 --e2e-workers 8 --e2e-serial-steps 0 > "$O/e2e_under_rocprof.json" 2>/dev/null
cp /tmp/pe/e_kernel_stats.csv "$O/rocprofv3_kernel_stats_e2e_bf16.csv"
head -30 "$O/rocprofv3_kernel_stats_e2e_bf16.csv" | cut -c1-200
tail -c 600 "$O/e2e_under_rocprof.json"
