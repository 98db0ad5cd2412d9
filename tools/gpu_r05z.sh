#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05z; mkdir -p "$O"; cd "$R"
rm -f "$O"/ab_*.txt
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "base:SSD_ABLATE=" "noreduce:SSD_ABLATE=reduce" "notail:SSD_ABLATE=tail"
timeout 600 tools/ab_variants.sh "$O/ab_f32.txt" 2 f32 "base:SSD_ABLATE=" "noreduce:SSD_ABLATE=reduce" "notail:SSD_ABLATE=tail"
