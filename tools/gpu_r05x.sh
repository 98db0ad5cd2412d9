#!/bin/bash
# bf16 evidence set + duty tables of both dtypes on the final tree
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
bash tools/profile_round.sh r05_final_bf16 --dtype bf16 > gpurun_out/r05_final_bf16_log.txt 2>&1
bash tools/pmc_duty.sh r05_final_bf16 --dtype bf16 --no-secondary > /dev/null 2>&1
bash tools/pmc_duty.sh r05_final --no-secondary > /dev/null 2>&1
ls gpurun_out/r05_final_bf16 gpurun_out/r05_final | head -40
