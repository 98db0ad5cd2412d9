#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06ag
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_tail.py tests/test_gpu_model.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/ab_variants.sh "$O/ab_cast_split_bf16.txt" 5 bf16 "cast_split_0:SSD_CAST_SPLIT=0" "cast_split_1:SSD_CAST_SPLIT=1"
