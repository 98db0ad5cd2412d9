#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06az
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
bash tools/ab_variants.sh "$O/ab_wino_phase_f32.txt" 3 f32 "base:SSD_X=0" "fw_phase:SSD_WINO_FW_PHASE=1" "bw_phase:SSD_WINO_BW_PHASE=1" "both:SSD_WINO_FW_PHASE=1 SSD_WINO_BW_PHASE=1"
