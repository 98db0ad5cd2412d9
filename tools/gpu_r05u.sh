#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05u; mkdir -p "$O"; cd "$R"
rm -f "$O"/ab_*.txt
L=$R/ssd_tensorflow_amd/libssdvgg_hip_prio.so
timeout 900 tools/ab_variants.sh "$O/ab_f32.txt" 3 f32 "base:SSD_X=0" "prio:SSD_LIB=$L"
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "base:SSD_X=0" "prio:SSD_LIB=$L"
