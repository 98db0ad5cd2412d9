#!/usr/bin/env python3
"""Shader clock and board power WHILE the conv kernels run (the "power cap" explanation of the
bf16 gather plateau rested on an inferred clock).  A sampler thread reads the GPU's clock / power sensors every
~50 ms (sysfs hwmon + pp_dpm_sclk when visible, else `rocm-smi --showclocks --showpower --json`) while the main
thread runs one kernel back to back for a few seconds per phase; per phase: achieved TFLOP/s, mean / min / max
sclk (MHz) and power (W).      python tools/power_probe.py > profiles/<tag>_power_clock.txt
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch                                     # noqa: E402
from gpu_util import lib, check, ptr, conv_geom  # noqa: E402


def find_sysfs():
    for card in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        hw = glob.glob(card + '/hwmon/hwmon*')
        if hw and (os.path.exists(hw[0] + '/power1_average') or os.path.exists(hw[0] + '/power1_input')) and os.path.exists(card + '/pp_dpm_sclk'):
            return card, hw[0]
    return None, None


CARD, HWMON = find_sysfs()


def read_sensors():
    """(sclk MHz or None, power W or None)"""
    if CARD:
        try:
            sclk = None
            f1 = HWMON + '/freq1_input'
            if os.path.exists(f1):
                sclk = int(open(f1).read()) / 1e6
            else:
                for line in open(CARD + '/pp_dpm_sclk'):
                    if line.strip().endswith('*'):
                        sclk = float(line.split(':')[1].strip().rstrip('*').strip().lower().replace('mhz', ''))
            pfile = HWMON + ('/power1_average' if os.path.exists(HWMON + '/power1_average') else '/power1_input')
            return sclk, int(open(pfile).read()) / 1e6
        except Exception:      # noqa: BLE001
            pass
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        c = next(iter(d.values()))
        sclk = power = None
        for k, v in c.items():
            kl = k.lower()
            if 'sclk' in kl and 'level' not in kl and sclk is None:
                sclk = float(str(v).strip('()').lower().replace('mhz', ''))
            if 'power' in kl and '(w)' in kl and power is None:
                power = float(v)
        return sclk, power
    except Exception:      # noqa: BLE001
        return None, None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.on = True
        self.rows = []

    def run(self):
        while self.on:
            self.rows.append(read_sensors())
            time.sleep(0.05 if CARD else 0.0)


def stats(vals):
    vals = [v for v in vals if v is not None]
    return 'n/a' if not vals else f'mean {sum(vals) / len(vals):7.1f}  min {min(vals):7.1f}  max {max(vals):7.1f}  ({len(vals)} samples)'


MON_STREAM = None


def phase(name, fn, flops, seconds=3.0):
    """fn back to back for `seconds`; meanwhile (a) the sysfs / rocm-smi sensors are polled from a thread (they lag by
    seconds on this box) and (b) ONE wave of a monitor kernel on a second stream samples the shader clock itself every
    100 us (s_memtime cycles per 10,000 ticks of the constant 100 MHz counter): that is the clock the kernels ran at."""
    global MON_STREAM
    if MON_STREAM is None:
        MON_STREAM = torch.cuda.Stream()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    nsamp = int(seconds * 1e4 * 0.6)              # the monitor ends before the load does
    mon = torch.zeros(nsamp, dtype=torch.int32, device='cuda')
    s = Sampler(); s.start()
    t0 = time.perf_counter(); n = 0
    for _ in range(20):
        fn()
    check(lib.ssd_op_clock_monitor(mon.data_ptr(), nsamp, 10000, MON_STREAM.cuda_stream))
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        if n % 500 == 0:
            torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.on = False; s.join()
    rows = s.rows[2:] or s.rows
    tf = flops * (n + 20) / dt / 1e12 if flops else 0.0
    mhz = mon.cpu().numpy().astype(float) / 10000 * 100
    mhz = mhz[mhz > 0]
    q = lambda p: float(__import__('numpy').percentile(mhz, p)) if len(mhz) else float('nan')
    print(f'{name:44s} {tf:8.1f} TFLOP/s | in-kernel shader clock MHz: mean {mhz.mean() if len(mhz) else float("nan"):7.1f}  p5 {q(5):7.1f}  p50 {q(50):7.1f}  '
          f'p95 {q(95):7.1f} ({len(mhz)} x 100 us) | sensors: sclk MHz {stats([r[0] for r in rows])} | power W {stats([r[1] for r in rows])}', flush=True)


def conv_fns(hw, ci, co, k, bf16, zero=False, B=32):
    ph, pw, ho, wo = conv_geom(hw, hw, k, 1, 1, 'SAME')
    geom = (B, hw, hw, ci, ho, wo, co, k, k, 1, 1, ph, pw)
    fl = 2.0 * B * ho * wo * co * ci * k * k
    dt = torch.bfloat16 if bf16 else torch.float32
    x = torch.relu(torch.randn((B, hw, hw, ci), device='cuda')).to(dt)
    w = torch.randn((k, k, ci, co), device='cuda') * 0.05
    dy = (torch.randn((B, ho, wo, co), device='cuda') * (torch.rand((B, ho, wo, co), device='cuda') > 0.5)).to(dt)
    if zero:
        x.zero_(); w.zero_(); dy.zero_()
    bias = torch.zeros(co, device='cuda'); y = torch.empty((B, ho, wo, co), device='cuda', dtype=dt)
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(bias)
    if bf16:
        wio = torch.empty((k * k, ci, co), device='cuda', dtype=dt); woi = torch.empty((k * k, co, ci), device='cuda', dtype=dt)
        check(lib.ssd_op_cast_filter(ptr(w), ptr(wio), ptr(woi), k * k, ci, co, None))
        ws = torch.empty((lib.ssd_op_conv2d_wgrad_bf16_ws_floats(*geom),), device='cuda')
        keep = (x, w, dy, bias, y, dx, dw, db, wio, woi, ws)
        return fl, keep, dict(
            fwd=lambda: check(lib.ssd_op_conv2d_fwd_bf16(ptr(x), ptr(woi), ptr(bias), ptr(y), 0, *geom, 1, None)),
            dgrad=lambda: check(lib.ssd_op_conv2d_dgrad_bf16(ptr(dy), ptr(wio), ptr(dx), ptr(x), 0, *geom, None)),
            wgrad=lambda: check(lib.ssd_op_conv2d_wgrad_bf16(ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(w), 0.0005, ptr(ws), *geom, None)))
    ws = torch.empty((lib.ssd_op_conv2d_wgrad_ws_floats(*geom),), device='cuda')
    keep = (x, w, dy, bias, y, dx, dw, db, ws)
    return fl, keep, dict(
        fwd=lambda: check(lib.ssd_op_conv2d_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), *geom, 1, None)),
        dgrad=lambda: check(lib.ssd_op_conv2d_dgrad(ptr(dy), ptr(w), ptr(dx), ptr(x), 0, *geom, None)),
        wgrad=lambda: check(lib.ssd_op_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(w), 0.0005, ptr(ws), *geom, None)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == 'sustained':      # round 4: any layer back to back under the environment's kernel choice
        table = {'conv2_2': (150, 128, 128), 'conv3_2': (75, 256, 256), 'conv4_2': (38, 512, 512), 'conv5_2': (19, 512, 512),
                 'conv2_1': (150, 64, 128), 'conv3_1': (75, 128, 256), 'conv4_1': (38, 256, 512), 'conv1_2': (300, 64, 64)}
        hw, ci, co = table[sys.argv[2]]
        bf16 = not (len(sys.argv) > 3 and sys.argv[3] == 'f32')
        fl, keep, fns = conv_fns(hw, ci, co, 3, bf16)
        env = ' '.join(f'{k[4:]}={v}' for k, v in sorted(os.environ.items()) if k.startswith('SSD_') and k not in ('SSD_BENCH_RELU',))
        for tag in ('fwd', 'dgrad', 'wgrad'):
            phase(f'{sys.argv[2]} {"bf16" if bf16 else "f32"} {tag} [{env or "default"}]', fns[tag], fl, 1.6)
        return
    if len(sys.argv) > 1 and sys.argv[1] in ('conv5_2', 'head1'):      # round 4: a 19x19 layer under whatever tile the environment selects
        fl, keep, fns = conv_fns(19, 512, 512, 3, True) if sys.argv[1] == 'conv5_2' else conv_fns(19, 1024, 152, 3, True)
        for tag in (('fwd', 'dgrad', 'wgrad') if sys.argv[1] == 'conv5_2' else ('fwd',)):
            phase(f'{sys.argv[1]} b32 bf16 (post-relu operands) {tag} [N64={os.environ.get("SSD_GATHER_ROWS_N64_BF16", "rule")}]', fns[tag], fl, 2.0)
        return
    print('sensors:', (CARD + ' + ' + HWMON) if CARD else 'rocm-smi --showclocks --showpower --json', '| first reading', read_sensors())
    phase('idle (monitor wave only)', lambda: time.sleep(0.001), 0.0, 1.5)
    for label, bf16, zero in (('fp32', False, False), ('bf16 (post-relu operands)', True, False), ('bf16 all-zero operands', True, True)):
        fl, keep, fns = conv_fns(38, 512, 512, 3, bf16, zero)
        for tag in ('fwd', 'dgrad', 'wgrad'):
            phase(f'conv4_2 b32 {label} {tag}', fns[tag], fl)
        del keep, fns
        torch.cuda.empty_cache()
    fl, keep, fns = conv_fns(75, 256, 256, 3, True)
    for tag in ('fwd', 'wgrad'):
        phase(f'conv3_2 b32 bf16 (post-relu operands) {tag}', fns[tag], fl)


if __name__ == '__main__':
    main()
