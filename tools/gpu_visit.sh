#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06ar
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
for lr in 0.00075 0.0006 0.0005; do
  timeout 300 python tools/learn_probe.py f32 "0.0003;$lr;0.0001" "96;768" 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee -a $O/learn_probe.txt
done
SSD_WINOGRAD=0 timeout 300 python tools/learn_probe.py f32 "0.0003;0.00075;0.0001" "96;768" 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee -a $O/learn_probe.txt
SSD_WINOGRAD=0 timeout 300 python tools/learn_probe.py f32 "0.0003;0.0006;0.0001" "96;768" 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee -a $O/learn_probe.txt
