#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06ax
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 2>&1 | grep -v amdgpu.ids | tail -14 | tee $O/gpu_tests.log
python bench.py --no-secondary --no-cpu-baseline --per-layer > $O/bench_f32.json 2> $O/per_layer_f32.txt; tail -c 600 $O/bench_f32.json
