#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06bh
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
for v in "" frag2 "" frag2; do
  lib=""; [ -n "$v" ] && lib="$R/ssd_tensorflow_amd/libssdvgg_hip_$v.so"
  echo "== variant ${v:-product}" | tee -a $O/gemm_frag2.txt
  SSD_LIB=$lib timeout 300 python tools/bench_conv.py conv2_2,conv3_2,conv4_2,conv5_2 2>&1 | grep -v amdgpu.ids | sed 's/ fwd.*| wino_fwd/ wino_fwd/' | tee -a $O/gemm_frag2.txt
done
SSD_LIB=$R/ssd_tensorflow_amd/libssdvgg_hip_frag2.so timeout 300 python -m pytest tests/test_gpu_winograd.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tee -a $O/gemm_frag2.txt
