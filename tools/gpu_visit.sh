#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_pool_fusion.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer 2>&1 > /dev/null | grep -E 'conv_first_fwd'
