#!/bin/bash
# which streams share what?  (the backward chain's kernels start only when the main stream's big dispatch has no workgroups left)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05q; mkdir -p "$O"; cd "$R"
rm -f "$O/ab_bf16.txt"
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 2 bf16 "base:SSD_X=0" "swap:SSD_STREAM_SWAP=1" "dummy1:SSD_DUMMY_STREAMS=1" "dummy2:SSD_DUMMY_STREAMS=2" "dummy3:SSD_DUMMY_STREAMS=3" "q8:GPU_MAX_HW_QUEUES=8" "q8d1:GPU_MAX_HW_QUEUES=8 SSD_DUMMY_STREAMS=1" "q8d2:GPU_MAX_HW_QUEUES=8 SSD_DUMMY_STREAMS=2"
