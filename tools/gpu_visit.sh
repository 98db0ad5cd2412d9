#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06u
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 -s > "$O/pytest_all.log" 2>&1
echo "rc=$?" >> "$O/pytest_all.log"
grep -v amdgpu.ids "$O/pytest_all.log" | grep -E "passed|failed|FAILED|rc=|data set|default handle|chain vs|s call" | tail -40
timeout 600 python bench.py > "$O/bench.json" 2> "$O/bench.stderr"; echo "bench rc=$?"
python - "$O/bench.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print('no bench line:', e); sys.exit(0)
print('HEADLINE', d['value'], d['unit'], d['ms_per_step'], 'ms', 'roofline', d['roofline'] and (d['roofline']['kernel'], d['roofline']['frac']))
for k in ('bf16', 'vgg512_b16', 'vgg512_b16_bf16', 'infer_b128', 'infer_b128_bf16', 'decode_b128', 'train_e2e', 'train_e2e_bf16'):
    s = d.get(k)
    if s:
        print(k, s.get('value'), s.get('ms_per_step'), s.get('error'))
print('cpu', d.get('cpu_baseline'))
PY
