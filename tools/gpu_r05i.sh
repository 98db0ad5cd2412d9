#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05i; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_pool_fusion.py -q -p no:cacheprovider -s -k "first_layer or step" > "$O/pool_fusion.log" 2>&1; echo "pool_fusion rc=$?"; grep -E "passed|failed|rel-L2|Error" "$O/pool_fusion.log" | tail -12
timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer > /dev/null 2> "$O/per_layer_bf16.txt"
grep -E "conv1_|first" "$O/per_layer_bf16.txt"
rm -f "$O/ab_bf16.txt"
timeout 500 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "fuse3:SSD_POOL_FUSE=3" "fuse7:SSD_POOL_FUSE=7"
