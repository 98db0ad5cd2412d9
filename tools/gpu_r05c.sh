#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05c; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_pool_fusion.py -q -p no:cacheprovider -s > "$O/pool_fusion.log" 2>&1; echo "pool_fusion rc=$?"; grep -E "passed|failed|rel-L2|Error" "$O/pool_fusion.log" | tail -30
rm -f "$O/ab_bf16.txt" "$O/ab_f32.txt"
timeout 500 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "fuse0:SSD_POOL_FUSE=0" "fuse3:SSD_POOL_FUSE=3" "fuse7:SSD_POOL_FUSE=7"
timeout 400 tools/ab_variants.sh "$O/ab_f32.txt" 2 f32 "fuse0:SSD_POOL_FUSE=0" "fuse2:SSD_POOL_FUSE=2" "fuse3:SSD_POOL_FUSE=3"
