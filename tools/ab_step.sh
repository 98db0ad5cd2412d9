#!/bin/bash
# A/B of the step-schedule switches on ONE box (box-to-box spread is +-1 % fp32, +-4 % bf16): every setting is timed twice,
# interleaved, 40 steps each.   tools/ab_step.sh <tag> [f32|bf16]      AB_CONFIGS="A=1;B=0 C=1" overrides the list (';'-separated)
TAG=${1:-ab}; DT=${2:-bf16}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
run() { env "$@" python bench.py --dtype $DT --steps ${AB_STEPS:-40} --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events ${AB_FLAGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f img/s %7.3f ms' % (d['value'], d['ms_per_step']))"; }
DEFAULT="SSD_NOP=1;SSD_EARLY_UPDATE=0;SSD_BW_SIDE=0;SSD_REDUCE_GROUPED=0;SSD_POOL_RECORD=0;SSD_EARLY_UPDATE=0 SSD_BW_SIDE=0 SSD_REDUCE_GROUPED=0 SSD_POOL_RECORD=0"
IFS=';' read -ra CFGS <<< "${AB_CONFIGS:-$DEFAULT}"
for rep in 1 2; do
  for cfg in "${CFGS[@]}"; do
    printf "%-80s " "$DT rep$rep [$cfg]"; run $cfg
  done
done | tee "$O/ab_$DT.txt"
