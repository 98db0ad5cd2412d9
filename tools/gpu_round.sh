#!/bin/bash
# One GPU-box visit: the -m gpu suite (no -x: every failure in one visit), then the default bench line and the
# per-layer tables.  tools/gpu_round.sh <tag> [pytest -k expression]     -> gpurun_out/<tag>/
set -u
TAG=${1:-visit}
KEXPR=${2:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
if [ -n "$KEXPR" ]; then
    timeout 1500 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider -k "$KEXPR" > "$O/pytest.log" 2>&1
else
    timeout 1500 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider > "$O/pytest.log" 2>&1
fi
echo "pytest rc=$?" >> "$O/pytest.log"
tail -40 "$O/pytest.log"
timeout 600 python bench.py > "$O/bench.json" 2> "$O/bench.stderr"; echo "bench rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer > /dev/null 2> "$O/per_layer_f32.txt"
timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer > /dev/null 2> "$O/per_layer_bf16.txt"
python - "$O/bench.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print('no bench line:', e); sys.exit(0)
print('HEADLINE', d['value'], d['unit'], d['ms_per_step'], 'ms', 'mfma_frac', d.get('model_mfma_frac'), 'roofline', d['roofline'] and (d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic']))
print('losses_check', d.get('losses_check'))
for k in ('bf16', 'vgg512_b16', 'vgg512_b16_bf16', 'infer_b128', 'decode_b128'):
    s = d.get(k)
    if s:
        print(k, s.get('value'), s.get('ms_per_step'), s.get('model_mfma_frac'), s.get('error'), s.get('roofline') and (s['roofline']['kernel'][:60], s['roofline']['frac']))
print('cpu', d.get('cpu_baseline'))
PY
# optional extras: GPU_EXTRA="power heads" tools/gpu_round.sh <tag>
for x in ${GPU_EXTRA:-}; do
    case $x in
    power) timeout 200 python tools/power_probe.py > "$O/power_clock.txt" 2>&1; cat "$O/power_clock.txt";;
    heads) for t in -1 0 1 2 3 4 5 6 7 8; do echo "SSD_TILE_BF16=$t"; SSD_TILE_BF16=$t timeout 100 python tools/bench_conv.py head1,head0,head2 bf16; done > "$O/heads_tiles_bf16.txt" 2>&1
           echo "SSD_GATHER_ROWS_BF16=0" >> "$O/heads_tiles_bf16.txt"; SSD_GATHER_ROWS_BF16=0 timeout 100 python tools/bench_conv.py head1,head0,head2 bf16 >> "$O/heads_tiles_bf16.txt" 2>&1
           cat "$O/heads_tiles_bf16.txt";;
    trace) cd /tmp && export TMPDIR=/tmp
           rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o d -- python "$R/bench.py" --mode decode --batch 128 --steps 20 --warmup 3 --no-cpu-baseline > "$O/bench_decode_under_rocprof.json" 2>/dev/null
           cp /tmp/pd/d_kernel_stats.csv "$O/rocprofv3_kernel_stats_decode.csv"
           rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o e -- python "$R/bench.py" --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap > "$O/bench_bf16_under_rocprof_serialized.json" 2>/dev/null
           cp /tmp/pe/e_kernel_stats.csv "$O/rocprofv3_kernel_stats_bf16_serialized.csv"
           cd "$R"; head -3 "$O/rocprofv3_kernel_stats_decode.csv" | cut -c1-200; grep -E "heads_kernel|loss_|sumsq|detect|reduce_grouped|cast_filters|l2norm" "$O/rocprofv3_kernel_stats_bf16_serialized.csv" | cut -c1-220;;
    headsweep) for t in 0 1 2 3 4 5 6 7 8; do echo "ROWS=0 SSD_TILE_BF16=$t"; SSD_GATHER_ROWS_BF16=0 SSD_TILE_BF16=$t timeout 100 python tools/bench_conv.py head1,head0 bf16; done > "$O/heads_tiles2_bf16.txt" 2>&1; cat "$O/heads_tiles2_bf16.txt";;
    prof) bash tools/profile_round.sh ${TAG}_prof > "$O/profile_round.log" 2>&1; tail -5 "$O/profile_round.log";;
    profbf16) bash tools/profile_round.sh ${TAG}_prof_bf16 --dtype bf16 > "$O/profile_round_bf16.log" 2>&1; tail -5 "$O/profile_round_bf16.log";;
    esac
done
