#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06bc
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
bash tools/ab_variants.sh "$O/ab_filter_split_f32.txt" 3 f32 "split:SSD_WINO_FILTER_SPLIT=1" "all_first:SSD_WINO_FILTER_SPLIT=0"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_winograd.py tests/test_gpu_drivers.py tests/test_gpu_parallel.py tests/test_gpu_bench_config.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|error" | tee $O/tests.txt
