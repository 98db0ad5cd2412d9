"""-m gpu: bench.py as the driver launches it -- one process, and two ranks under torch.distributed.run (both on GPU 0
over gloo: the plumbing of `--gpus N`, a multi-GPU node is not available to the builder) -- prints ONE JSON line with the
contract's keys; and the end-to-end mode in a fresh process."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
        'config', 'roofline', 'cpu_baseline'}


def _one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_single_process_line():
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '3', '--warmup', '1', '--batch', '4', '--no-secondary', '--no-cpu-baseline'],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert KEYS <= set(d) and d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['value'] > 0 and d['scaling'] == 'weak'
    assert d['roofline']['bound'] == 'mfma' and 0 < d['roofline']['frac'] <= 1 and d['self_check'].startswith('dominant kernel <= step')
    assert abs(d['value'] - 4 * 3 / (d['ms_per_step'] * 3e-3)) < 0.02 * d['value']        # value = images of the timed steps / their time


def test_two_ranks_under_torchrun():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '4', '--backend', 'gloo',
                        '--same-device', '--no-secondary', '--no-cpu-baseline', '--no-kernel-events'],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = _one_line(r.stdout)                                              # rank 0 alone prints
    assert KEYS <= set(d) and d['n_gpus'] == 2 and d['config']['global_batch'] == 8 and d['config']['parallelism'] == 'dp2'
    assert d['config']['replicas_agree'] is True and 'bucketed' in d['config']['allreduce'] and d['value'] > 0


def test_end_to_end_mode_in_a_fresh_process():
    r = subprocess.run([sys.executable, 'bench.py', '--mode', 'train_e2e', '--batch', '4', '--e2e-steps', '4', '--e2e-workers', '2',
                        '--e2e-serial-steps', '2'], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d['value'] > 0 and d['feeder_workers'] == 2 and d['steps'] == 4 and d['serial_feeder']['steps'] == 2
    assert set(d['feeder_ms_per_step']) >= {'consumer_wait', 'worker_wait', 'slot_wait', 'upload'}
    assert all(v == v and v > 0 for v in d['mean_losses'].values())
