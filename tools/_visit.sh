set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_p; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "pool or l2norm or test_gpu_model or bf16_step or lane or step0" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -5
for dt in bf16 f32; do
  timeout 300 python bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --per-layer > /dev/null 2> $O/per_layer_$dt.txt
  grep -E "maxpool|l2norm" $O/per_layer_$dt.txt | cut -c1-120
  timeout 200 python bench.py --dtype $dt --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', d['value'], d['ms_per_step'], 'issue', d['host_issue_ms_per_step'])"
done
