#!/bin/bash
# data gradients carry their completion event (SSD_STOP_EVENTS) x backward order (SSD_BW_DEFER)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05s; mkdir -p "$O"; cd "$R"
rm -f "$O"/ab_*.txt
for us in 5 20 100; do timeout 60 tools/probes/event_gap.bin $us 64; done > "$O/event_gap.txt" 2>&1
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_pool_fusion.py tests/test_gpu_parallel.py tests/test_gpu_bf16.py -q -p no:cacheprovider -x 2>&1 | tail -3
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "rec:SSD_STOP_EVENTS=0" "stop:SSD_STOP_EVENTS=1" "rec_d2:SSD_STOP_EVENTS=0 SSD_BW_DEFER=2" "stop_d2:SSD_STOP_EVENTS=1 SSD_BW_DEFER=2"
timeout 600 tools/ab_variants.sh "$O/ab_f32.txt" 2 f32 "rec:SSD_STOP_EVENTS=0" "stop:SSD_STOP_EVENTS=1"
