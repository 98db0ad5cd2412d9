#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06bd
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
bash tools/ab_variants.sh "$O/ab_bw_order_bf16.txt" 5 bf16 "heads_last:SSD_BW_BIG_HEADS_LAST=1" "graph_order:SSD_BW_BIG_HEADS_LAST=0"
bash tools/ab_variants.sh "$O/ab_bw_order_f32.txt" 3 f32 "graph_order:SSD_BW_BIG_HEADS_LAST=0" "heads_last:SSD_BW_BIG_HEADS_LAST=1"
