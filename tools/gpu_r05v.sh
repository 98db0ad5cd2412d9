#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05v; mkdir -p "$O"; cd "$R"
rm -f "$O"/ab_*.txt
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_pool_fusion.py tests/test_gpu_parallel.py tests/test_gpu_bf16.py tests/test_gpu_detect.py -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "rec:SSD_STOP_EVENTS=0" "stop:SSD_STOP_EVENTS=1"
timeout 600 tools/ab_variants.sh "$O/ab_f32.txt" 2 f32 "rec:SSD_STOP_EVENTS=0" "stop:SSD_STOP_EVENTS=1"
