set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_q; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -8
for dt in bf16 f32; do
  timeout 300 python bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer > /dev/null 2> $O/per_layer_$dt.txt
  timeout 200 python bench.py --dtype $dt --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', d['value'], d['ms_per_step'], 'issue', d['host_issue_ms_per_step'])"
done
grep -E "conv_dgrad|wgrad_reduce" $O/per_layer_bf16.txt | cut -c1-110 | head -60
grep -E "conv_dgrad" $O/per_layer_f32.txt | cut -c1-110 | head -40
