#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06ab
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_augment.py tests/test_gpu_feeder.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -5
for rep in 1 2; do
for t in _base .; do
  (cd $R/$t && python bench.py --mode augment --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tree=$t', d['value'], 'img/s', d['roofline']['avg_launch_us'], 'us/batch')")
done; done | tee "$O/augment_gather_ab.txt"
cd /tmp && export TMPDIR=/tmp
for t in _base .; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_$$ -o a -- python "$R/$t/bench.py" --mode augment --no-cpu-baseline --steps 50 --warmup 5 > /dev/null 2>&1
  echo "tree=$t"; grep -i augment /tmp/pa_$$/a_kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*)//' ; rm -rf /tmp/pa_$$
done | tee -a "$O/augment_gather_ab.txt"
