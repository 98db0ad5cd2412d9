#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06be
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_model.py tests/test_gpu_pool_fusion.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror" | tee $O/tests.txt
bash tools/ab_variants.sh "$O/ab_mask_bits_f32.txt" 3 f32 "bits:SSD_WINO_MASK_BITS=1" "fp32_mask:SSD_WINO_MASK_BITS=0"
