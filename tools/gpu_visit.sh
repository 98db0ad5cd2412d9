#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06aw
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_winograd.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests.txt
timeout 600 python tools/bench_conv.py conv1_2,conv2_1,conv2_2,conv4_2 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a $O/bench_conv.txt
python tools/step_time.py --dtype f32 --steps 10 --reps 3 --tag cc8192 2>&1 | grep -v amdgpu.ids | tee $O/step.txt
SSD_WINO_MIN_CC=4096 python tools/step_time.py --dtype f32 --steps 10 --reps 3 --tag cc4096_conv1_2_too 2>&1 | grep -v amdgpu.ids | tee -a $O/step.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o w -- python $R/tools/bench_conv.py conv1_2 > /dev/null 2>&1
python - <<'PY' | tee $O/conv1_2_kernels.txt
import csv,glob
for f in glob.glob('/tmp/pw/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f))):
        if 'ssd' in r['Name']: print('%-80s calls %5s avg %10.1f us' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
