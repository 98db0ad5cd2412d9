#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-full}
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 1400 python -m pytest tests -m gpu -q --durations=20 -p no:cacheprovider > "$O/gpu_tests.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/gpu_tests.log"
tail -30 "$O/gpu_tests.log"
