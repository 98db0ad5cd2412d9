#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06ak
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
for g in 2048 1024 768 512; do
  SSD_FIRST_GRID_F32=$g timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-overlap --per-layer 2>&1 > /dev/null | grep -E 'conv_first_fwd' | sed "s/^/grid=$g /"
done | tee "$O/per_layer_conv1_1_grid_f32.txt"
