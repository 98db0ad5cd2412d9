#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06t
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_tail.py tests/test_gpu_parallel.py tests/test_gpu_bf16.py -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -5
bash tools/ab_variants.sh "$O/ab_tail_bf16.txt" 4 bf16 "per_layer:SSD_TAIL_FUSE=0" "chain_fwd:SSD_TAIL_FUSE=1" "chain_bwd:SSD_TAIL_FUSE=2" "chain_both:SSD_TAIL_FUSE=3"
