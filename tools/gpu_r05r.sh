#!/bin/bash
# backward order: the big heads behind the chain (SSD_BW_DEFER), both dtypes, + a timeline of the candidates
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05r; mkdir -p "$O"; cd "$R"
rm -f "$O"/ab_*.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -p no:cacheprovider -x -k "step_vgg300 or layer_local" 2>&1 | tail -2
for V in 1 2 3; do SSD_BW_DEFER=$V timeout 300 python -m pytest tests/test_gpu_model.py -q -p no:cacheprovider -x -k "step_vgg300 or benchmarked" 2>&1 | tail -1; done
timeout 900 tools/ab_variants.sh "$O/ab_bf16.txt" 3 bf16 "base:SSD_BW_DEFER=0" "d1:SSD_BW_DEFER=1" "d2:SSD_BW_DEFER=2" "d3:SSD_BW_DEFER=3"
timeout 600 tools/ab_variants.sh "$O/ab_f32.txt" 2 f32 "base:SSD_BW_DEFER=0" "d1:SSD_BW_DEFER=1" "d2:SSD_BW_DEFER=2" "d3:SSD_BW_DEFER=3"
cd /tmp; export TMPDIR=/tmp
for V in 2 3; do
SSD_BW_DEFER=$V rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_$V -o t -- python $R/bench.py --dtype bf16 --steps 6 --warmup 6 --no-cpu-baseline --no-secondary --no-kernel-events > /dev/null 2>&1
python $R/tools/timeline.py /tmp/pt_$V/t_kernel_trace.csv > "$O/timeline_bf16_d$V.txt" 2>&1
done
