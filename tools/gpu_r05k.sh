#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05k; mkdir -p "$O"; cd "$R"
timeout 300 python -m pytest tests/test_gpu_bench_config.py -q -p no:cacheprovider -k "lane_settings" 2>&1 | tail -3
rm -f "$O/ab_bf16.txt" "$O/ab_f32.txt"
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "base:SSD_FWD_MERGE_TAIL=0" "merge:SSD_FWD_MERGE_TAIL=1" "merge_k2:SSD_SMALL_KSPLIT=2" "merge_k0:SSD_SMALL_KSPLIT=0" "merge_prio:SSD_SIDE_PRIO=1" "merge_k2_prio:SSD_SMALL_KSPLIT=2 SSD_SIDE_PRIO=1"
timeout 400 tools/ab_variants.sh "$O/ab_f32.txt" 2 f32 "base:SSD_FWD_MERGE_TAIL=0" "merge:SSD_FWD_MERGE_TAIL=1" "merge_prio:SSD_SIDE_PRIO=1"
