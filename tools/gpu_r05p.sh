#!/bin/bash
# timelines of the default step, both dtypes (round 5, final tree)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05p; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for DT in bf16 f32; do
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_$DT -o t -- python $R/bench.py --dtype $DT --steps 6 --warmup 6 --no-cpu-baseline --no-secondary --no-kernel-events > /dev/null 2>&1
python $R/tools/timeline.py /tmp/pt_$DT/t_kernel_trace.csv > "$O/timeline_$DT.txt" 2>&1
python $R/tools/trace_gaps.py /tmp/pt_$DT/t_kernel_trace.csv > "$O/trace_gaps_$DT.txt" 2>&1
head -1 "$O/trace_gaps_$DT.txt"
done
