#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05o; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_pool_fusion.py tests/test_gpu_parallel.py -q -p no:cacheprovider -x 2>&1 | tail -3
rm -f "$O/ab_bf16.txt" "$O/ab_f32.txt"
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "side:SSD_BW_SMALL_HEADS_MAIN=0" "main:SSD_BW_SMALL_HEADS_MAIN=1"
timeout 400 tools/ab_variants.sh "$O/ab_f32.txt" 2 f32 "side:SSD_BW_SMALL_HEADS_MAIN=0" "main:SSD_BW_SMALL_HEADS_MAIN=1"
