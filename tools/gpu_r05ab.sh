#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05ab; mkdir -p "$O"; cd "$R"
rm -f "$O"/ab_*.txt
SSD_F32_LOADER=3 SSD_F32_LOADER_MINK=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -x -k "conv" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
SSD_F32_LOADER=3 timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_pool_fusion.py -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 900 tools/ab_variants.sh "$O/ab_f32.txt" 2 f32 "base:SSD_F32_LOADER=0" "fwd:SSD_F32_LOADER=1" "dgrad:SSD_F32_LOADER=2" "both:SSD_F32_LOADER=3"
