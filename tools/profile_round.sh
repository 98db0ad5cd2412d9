#!/bin/bash
# Collects the per-round evidence set on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag> [extra bench.py flags for the headline run]
# writes gpurun_out/<tag>/{bench.json, per_layer_events.txt, rocprofv3 kernel stats (serialized and
# overlapped), pmc_FETCH_SIZE/WRITE_SIZE, infer/decode/vgg512 bench lines}.  Copy into profiles/<tag>_*.
set -u
TAG=${1:-rXX}
shift || true
EXTRA="$*"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp

# 1. kernel trace, one kernel at a time (durations comparable with bench.py's HIP events)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o a -- \
    python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap $EXTRA \
    > "$O/bench_under_rocprof_serialized.json" 2>/dev/null
cp /tmp/pa/a_kernel_stats.csv "$O/rocprofv3_kernel_stats_serialized.csv"
# 2. kernel trace of the overlapped step (what the headline runs)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- \
    python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $EXTRA \
    > "$O/bench_under_rocprof_overlapped.json" 2>/dev/null
cp /tmp/pb/b_kernel_stats.csv "$O/rocprofv3_kernel_stats_overlapped.csv"
# 2b. GPU idle time per step: a trace of overlapped steps only (no per-launch events, no serialized second half)
rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o g -- \
    python "$R/bench.py" --steps 8 --warmup 8 --no-cpu-baseline --no-secondary --no-kernel-events $EXTRA > /dev/null 2>&1
python "$R/tools/trace_gaps.py" /tmp/pg/g_kernel_trace.csv > "$O/trace_gaps_overlapped.txt" 2>&1
# 3. HBM traffic counters, one pass each, kernel trace only
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pc_$C -o c -- \
        python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-overlap --no-kernel-events $EXTRA >/dev/null 2>&1
    python - "$C" /tmp/pc_$C/c_counter_collection.csv > "$O/pmc_$C.txt" <<'EOF'
import csv, sys, collections
name, path = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    if r['Counter_Name'] != name: continue
    k = r['Kernel_Name'][:90]
    tot[k] += float(r['Counter_Value']); cnt[k] += 1
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f'{k}\t{name}\tlaunches={cnt[k]}\tavg={tot[k] / cnt[k]:.4g}\tsum={tot[k]:.4g}')
EOF
done
# 3b. the same counters and the kernel trace for the decode + NMS pass (config 5)
if [ -z "$EXTRA" ]; then
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pdc_$C -o c -- \
        python "$R/bench.py" --mode decode --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events >/dev/null 2>&1
    python "$R/tools/pmc_table.py" "$C" /tmp/pdc_$C/c_counter_collection.csv > "$O/decode_pmc_$C.txt"
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pdd -o d -- \
    python "$R/bench.py" --mode decode --batch 128 --steps 20 --warmup 3 --no-cpu-baseline > "$O/bench_decode_under_rocprof.json" 2>/dev/null
cp /tmp/pdd/d_kernel_stats.csv "$O/rocprofv3_kernel_stats_decode_b128.csv"
fi
# 4. per-layer event table + the headline line (with the CPU baseline)
cd "$R"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer $EXTRA > /dev/null 2> "$O/per_layer_events.txt"
python bench.py $EXTRA > "$O/bench.json" 2> "$O/bench.stderr"
python bench.py --mode infer --batch 128 --no-cpu-baseline $EXTRA > "$O/bench_infer_b128.json" 2>/dev/null
python bench.py --preset vgg512 --batch 16 --no-cpu-baseline $EXTRA > "$O/bench_vgg512_b16.json" 2>/dev/null
if [ -z "$EXTRA" ]; then
    python bench.py --mode decode --batch 128 --no-cpu-baseline > "$O/bench_decode_b128.json" 2>/dev/null
fi
cat "$O/bench.json"
head -4 "$O/rocprofv3_kernel_stats_serialized.csv" | cut -c1-160
