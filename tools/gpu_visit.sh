#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06ad
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
for g in 256 512 768 1024 2048 4096; do
  SSD_FIRST_GRID=$g timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer 2>&1 > /dev/null | grep -E 'conv_first_fwd' | sed "s/^/grid=$g /"
done | tee "$O/per_layer_conv1_1_grid.txt"
