#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05n; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_pool_fusion.py tests/test_gpu_bf16.py -q -p no:cacheprovider -x 2>&1 | tail -4
for dt in bf16 f32; do python tools/step_time.py --dtype $dt --steps $([ $dt = bf16 ] && echo 40 || echo 12) --reps 3; done 2>/dev/null | tee "$O/step_time.txt"
