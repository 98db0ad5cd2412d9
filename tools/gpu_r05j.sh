#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05p; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
export SSD_SMALL_KSPLIT=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o t -- python $R/bench.py --dtype bf16 --steps 6 --warmup 6 --no-cpu-baseline --no-secondary --no-kernel-events > /dev/null 2>&1
python $R/tools/timeline.py /tmp/pt/t_kernel_trace.csv > "$O/timeline_bf16.txt" 2>&1
python $R/tools/trace_gaps.py /tmp/pt/t_kernel_trace.csv > "$O/trace_gaps_bf16.txt" 2>&1
head -3 "$O/trace_gaps_bf16.txt"
