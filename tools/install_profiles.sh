#!/bin/bash
# Copies one GPU visit's evidence set (tools/gpu_round.sh <tag> with GPU_EXTRA="prof profbf16") from gpurun_out/ into
# profiles/ under the names DESIGN.md cites, and drops an older set.   tools/install_profiles.sh <tag> [old-tag-to-remove]
set -e
TAG=$1; OLD=${2:-}
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
for f in gpurun_out/${TAG}_prof/*; do b=$(basename $f); [ $b = bench.stderr ] || cp $f profiles/${TAG}_$b; done
for f in gpurun_out/${TAG}_prof_bf16/*; do b=$(basename $f); [ $b = bench.stderr ] || cp $f profiles/${TAG}_bf16_$b; done
cp gpurun_out/$TAG/pytest.log profiles/${TAG}_gpu_tests.log
cp gpurun_out/$TAG/bench.json profiles/${TAG}_bench_default_with_secondary.json
[ -z "$OLD" ] || git rm -q --ignore-unmatch profiles/${OLD}_*
ls profiles | grep "^${TAG}_" | wc -l
