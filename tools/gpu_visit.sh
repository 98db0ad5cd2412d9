#!/bin/bash
# Round-6 evidence visit: the whole -m gpu suite, smoke, the default bench line, both profile sets, duty tables, timelines.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
TAG=${1:-r06_zz}
cd "$R"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/${TAG}_smoke.txt
GPU_EXTRA="prof profbf16" bash tools/gpu_round.sh $TAG
bash tools/pmc_duty.sh ${TAG}_duty > /dev/null 2>&1
bash tools/pmc_duty.sh ${TAG}_duty_bf16 --dtype bf16 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_$dt -o t -- python "$R/bench.py" --dtype $dt --steps 8 --warmup 8 --no-cpu-baseline --no-secondary --no-kernel-events > /dev/null 2>&1
  python "$R/tools/timeline.py" /tmp/pt_$dt/t_kernel_trace.csv > "$R/gpurun_out/$TAG/timeline_$dt.txt" 2>&1
done
ls "$R/gpurun_out/$TAG" "$R/gpurun_out/${TAG}_prof" "$R/gpurun_out/${TAG}_prof_bf16" | head -60
