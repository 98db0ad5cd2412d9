#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05m; mkdir -p "$O"; cd "$R"
rm -f "$O/ab_bf16.txt"
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "k4:SSD_SMALL_KSPLIT=1" "k2:SSD_SMALL_KSPLIT=2" "k2_noheadrows8:SSD_SMALL_KSPLIT=2 SSD_WGRAD_ROWS8_HEADS=0" "k4_noheadrows8:SSD_WGRAD_ROWS8_HEADS=0"
