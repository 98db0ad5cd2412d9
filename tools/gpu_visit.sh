#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06aj
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_pool_fusion.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_variants.sh "$O/ab_dgrad_n64_tile_f32.txt" 4 f32 "n64_tile_64x64:SSD_DGRAD_N64_TILE=3" "n64_tile_128x64:SSD_DGRAD_N64_TILE=1"
