"""CPU: the shape of bench.py's JSON line (no GPU: only the pure functions) and the trace tools' parsing."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module_under_test', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_line_ends_in_a_compact_summary_and_keeps_the_contract_keys():
    b = _bench()
    roof = dict(bound='mfma', kernel='k', achieved=1.0, peak=2.0, unit='TFLOP/s', frac=0.5, traffic=None)
    out = dict(metric='m', value=1.0, unit='images/s', n_gpus=1, steps=20, warmup=5, ms_per_step=50.0, higher_is_better=True, scaling='weak',
               vs_baseline=None, dtype='f32', data='synthetic', config={'workload': 'w'}, roofline=roof,
               kernel_ms_per_step={'a' * 40: 1.0}, cpu_baseline={'value': 0.7}, losses_check={'ok': True},
               bf16=dict(value=4400.0, unit='images/s', ms_per_step=7.3, model_mfma_frac=0.33, roofline=dict(frac=0.38), losses_check={'ok': True}),
               decode_b128=dict(value=1.9e6, unit='images/s', ms_per_step=0.067, roofline=dict(frac=0.28)),
               train_e2e_bf16=dict(value=4100.0, unit='images/s', vs_resident_input=0.94, detections_collected=1406),
               vgg512_b16={'error': 'RuntimeError: boom'})
    o = b.ordered_for_tail(out)
    keys = list(o)
    assert keys[0] == 'kernel_ms_per_step' and keys[-1] == 'summary'
    assert set(out) | {'summary'} == set(o), 'nothing is dropped from the line'
    assert keys.index('bf16') > keys.index('decode_b128') and keys.index('value') > keys.index('bf16'), 'blocks first, the contract keys after them'
    s = o['summary']
    assert s['bf16'] == dict(value=4400.0, unit='images/s', ms_per_step=7.3, model_mfma_frac=0.33, roofline_frac=0.38, losses_ok=True)
    assert s['train_e2e_bf16']['detections_collected'] == 1406 and s['vgg512_b16'] == 'error'
    tail = json.dumps(o)[-1200:]
    assert '"bf16": {"value": 4400.0' in tail, 'a reader of the tail of the line sees the bf16 value'


def test_trace_tools_parse_a_kernel_trace(tmp_path):
    """tools/timeline.py and tools/trace_gaps.py on a synthetic rocprofv3 kernel trace: five steps delimited by the optimizer kernel,
    two queues, one gap."""
    path = tmp_path / 'trace.csv'
    rows = []
    t = 1000
    for step in range(5):
        rows.append(dict(Start_Timestamp=t, End_Timestamp=t + 5000, Kernel_Name='void ssd::big_kernel<1, 2>(Args)', Queue_Id=1, Grid_Size_X=256 * 1024, Workgroup_Size_X=256))
        rows.append(dict(Start_Timestamp=t + 1000, End_Timestamp=t + 3000, Kernel_Name='ssd::side_kernel(int)', Queue_Id=2, Grid_Size_X=256 * 4, Workgroup_Size_X=256))
        rows.append(dict(Start_Timestamp=t + 6000, End_Timestamp=t + 8000, Kernel_Name='ssd::tiny_kernel(int)', Queue_Id=1, Grid_Size_X=256 * 4, Workgroup_Size_X=256))
        rows.append(dict(Start_Timestamp=t + 8000, End_Timestamp=t + 9000, Kernel_Name='ssd::momentum_kernel(float*)', Queue_Id=1, Grid_Size_X=256 * 4096, Workgroup_Size_X=256))
        t += 10000
    with open(path, 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader(); w.writerows(rows)
    tl = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'timeline.py'), str(path)], capture_output=True, text=True, timeout=60)
    assert tl.returncode == 0, tl.stderr
    assert 'step of 4 launches' in tl.stdout and 'big_kernel<1, 2>' in tl.stdout
    assert 'time with fewer than 256 workgroups in flight: 4.0 us' in tl.stdout      # 1 + 1 us idle, 2 us of the 4-workgroup kernel alone
    gp = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_gaps.py'), str(path)], capture_output=True, text=True, timeout=60)
    assert gp.returncode == 0, gp.stderr
    assert 'idle 2.0 us/step' in gp.stdout


def test_self_check_prices_the_winograd_step_by_executed_flops():
    """model_mfma_frac (algorithmic direct-convolution FLOPs) may exceed 1 only in a block that says which algorithm ran
    (DESIGN.md 4.9); the fraction of the peak the matrix pipe was ASKED for may never."""
    import pytest
    b = _bench()
    base = dict(ms_per_step=24.0, kernel_ms_per_step={'wino_gemm_128x128': 7.0}, kernel_ms_sum_per_step=26.0,
                roofline=dict(frac=0.72), model_mfma_frac=1.6, executed_mfma_frac=0.47)
    with pytest.raises(SystemExit):
        b.self_check(dict(base), True)                                   # > 1 without the algorithm note
    assert 'fractions' in b.self_check(dict(base, algorithm='winograd'), True)
    with pytest.raises(SystemExit):
        b.self_check(dict(base, algorithm='winograd', executed_mfma_frac=1.02), True)
    with pytest.raises(SystemExit):
        b.self_check(dict(base, algorithm='winograd', model_mfma_frac=4.5), True)
    assert 'wino_gemm_128x128' in b.KERNEL_SYMBOLS and 'wino_out_unpool' in b.KERNEL_SYMBOLS
