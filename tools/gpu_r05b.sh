#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05b; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_pool_fusion.py -q -p no:cacheprovider > "$O/pool_fusion.log" 2>&1; echo "pool_fusion rc=$?"; tail -4 "$O/pool_fusion.log"
timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer > "$O/bench_bf16_serial.json" 2> "$O/per_layer_bf16.txt"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer > "$O/bench_f32_serial.json" 2> "$O/per_layer_f32.txt"
timeout 300 python bench.py --dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-secondary > "$O/bench_bf16.json" 2> /dev/null
python -c "
import json
for f in ('bench_bf16','bench_bf16_serial','bench_f32_serial'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d.get('kernel_ms_sum_per_step'))
"
