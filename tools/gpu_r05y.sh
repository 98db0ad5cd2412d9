#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05y; mkdir -p "$O"; cd "$R"
rm -f "$O"/ab_*.txt
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 4 bf16 "base:SSD_TAIL_PREFETCH=0" "pref:SSD_TAIL_PREFETCH=1"
timeout 600 tools/ab_variants.sh "$O/ab_f32.txt" 3 f32 "base:SSD_TAIL_PREFETCH=0" "pref:SSD_TAIL_PREFETCH=1"
