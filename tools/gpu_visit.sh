#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout 600 python -m pytest tests/test_gpu_tail.py -q -p no:cacheprovider -s -k "step" 2>&1 | grep -v amdgpu.ids | grep -E "chain vs|worst|^E |passed|failed" | head -20
