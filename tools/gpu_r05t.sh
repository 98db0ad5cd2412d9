#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05t; mkdir -p "$O"; cd "$R"
rm -f "$O"/ab2_*.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -x -k conv 2>&1 | grep -E "passed|failed" | tail -2
SSD_F32_BIG=1 timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -p no:cacheprovider -x -k "conv or benchmarked" 2>&1 | grep -E "passed|failed" | tail -2
timeout 900 tools/ab_variants.sh "$O/ab2_f32.txt" 3 f32 "base:SSD_SETPRIO=2" "big1:SSD_SETPRIO=2 SSD_F32_BIG=1" "big700:SSD_SETPRIO=2 SSD_F32_BIG=700"
python bench.py --mode layers --dtype f32 2>/dev/null | head -0
