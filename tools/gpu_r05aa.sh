#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05aa; mkdir -p "$O"; cd "$R"
rm -f "$O"/ab_*.txt
timeout 900 tools/ab_variants.sh "$O/ab_f32.txt" 2 f32 "p50:SSD_FWD_LANE0_PCT=50" "p56:SSD_FWD_LANE0_PCT=56" "p62:SSD_FWD_LANE0_PCT=62" "p69:SSD_FWD_LANE0_PCT=69"
timeout 600 tools/ab_variants.sh "$O/ab_bf16.txt" 2 bf16 "p50:SSD_FWD_LANE0_PCT=50" "p56:SSD_FWD_LANE0_PCT=56" "p62:SSD_FWD_LANE0_PCT=62" "p69:SSD_FWD_LANE0_PCT=69"
