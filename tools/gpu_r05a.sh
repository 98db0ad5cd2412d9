#!/bin/bash
# round 5, visit a: fused-pool parity first, then the pruned kernels' suites, then the same-box A/B of the fusion.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05a; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_pool_fusion.py -q -p no:cacheprovider --durations=8 > "$O/pool_fusion.log" 2>&1; echo "pool_fusion rc=$?" | tee -a "$O/pool_fusion.log"
tail -25 "$O/pool_fusion.log"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_model.py tests/test_gpu_boxes.py tests/test_gpu_parallel.py -q -p no:cacheprovider --durations=8 > "$O/kernels.log" 2>&1; echo "kernels rc=$?" | tee -a "$O/kernels.log"
tail -15 "$O/kernels.log"
rm -f "$O/ab_bf16.txt" "$O/ab_f32.txt"
timeout 500 tools/ab_variants.sh "$O/ab_bf16.txt" 2 bf16 "fuse0:SSD_POOL_FUSE=0" "fuse1:SSD_POOL_FUSE=1" "fuse2:SSD_POOL_FUSE=2" "fuse3:SSD_POOL_FUSE=3"
timeout 400 tools/ab_variants.sh "$O/ab_f32.txt" 1 f32 "fuse0:SSD_POOL_FUSE=0" "fuse1:SSD_POOL_FUSE=1" "fuse2:SSD_POOL_FUSE=2" "fuse3:SSD_POOL_FUSE=3"
