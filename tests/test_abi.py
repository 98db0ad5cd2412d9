"""CPU: the C-ABI library loads without a GPU and exports every symbol include/ssdvgg_hip.h
declares; host-only entry points (presets, errors) behave like the reference's."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'ssdvgg_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ssd_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from ssd_tensorflow_amd import _lib
    names = declared_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(_lib.lib, n), f'{n} declared in include/ssdvgg_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in _lib.py'
    assert set(_lib.SIGNATURES) == set(names)


def test_presets_and_errors_without_gpu():
    from ssd_tensorflow_amd._lib import lib, last_error
    w = C.c_int(); h = C.c_int(); a = C.c_int(); m = C.c_int()
    assert lib.ssd_preset_info(b'vgg300', w, h, a, m) == 0 and (w.value, h.value, a.value, m.value) == (300, 300, 8732, 6)
    assert lib.ssd_preset_info(b'vgg512', w, h, a, m) == 0 and (w.value, a.value, m.value) == (512, 24564, 7)
    assert lib.ssd_preset_info(b'vgg999', w, h, a, m) != 0 and last_error() == 'No such preset: vgg999'
    size = C.c_int(); scale = C.c_double(); nt = C.c_int()
    assert lib.ssd_preset_map(b'vgg300', 1, size, scale, nt) == 0 and (size.value, scale.value, nt.value) == (19, 0.2, 6)
    assert lib.ssd_preset_map(b'vgg300', 9, size, scale, nt) != 0
    # arena = the reference's 26,285,486 / 26,959,300 parameters + the fused heads' zero padding columns
    # (widths rounded up to 8 channels: 150 -> 152 on maps 1-3, 100 -> 104 on maps 0, 4, 5)
    assert lib.ssd_arena_floats(b'vgg300', 20) == (26285486 + 2 * 9 * (1024 + 512 + 256) + 3 * 2
                                                  + 4 * 9 * (512 + 256 + 256) + 3 * 4)
    assert lib.ssd_arena_floats(b'vgg512', 20) >= 26959300
    assert b'gfx950' in lib.ssd_version()


def test_mirror_module_surface():
    from ssd_tensorflow_amd import ssdutils as su, utils as ut, ssdvgg, transforms
    for name in ('SSD_PRESETS', 'get_preset_by_name', 'get_anchors_for_preset', 'anchors2array', 'decode_boxes',
                 'suppress_overlaps', 'SSDMap', 'SSDPreset', 'Anchor'):
        assert hasattr(su, name)
    with pytest.raises(RuntimeError, match='No such preset: nope'):
        su.get_preset_by_name('nope')
    p = su.get_preset_by_name('vgg512')
    assert p.num_anchors == 24564 and len(p.maps) == 7 and p.maps[0].size == (64, 64) and p.extra_scale == 1.05
    # utils geometry: int() truncation and the non-identity round trip
    assert ut.prop2abs(ut.Point(0.5 / 38, 0.5 / 38), ut.Size(0.1, 0.1), ut.Size(1000, 1000))[:2] == (-36, 63)
    c, s = ut.abs2prop(5, 996, 0, 0, ut.Size(1000, 1000))
    rt = ut.prop2abs(c, s, ut.Size(1000, 1000))
    assert rt[0] in (4, 5) and rt[1] == 996
    b = ut.normalize_box(ut.Box('x', 1, ut.Point(0.5, 0.5), ut.Size(2.0, 0.5)))
    assert ut.prop2abs(b.center, b.size, ut.Size(1000, 1000))[1] <= 999
    assert ut.str2bool('Yes') is True and ut.str2bool('0') is False
    net = ssdvgg.SSDVGG(None, 'vgg300')
    assert net.original_scopes[0] == 'conv1_1' and 'classifiers/classifier5_3' in net.new_scopes
    assert len(net.new_scopes) == 8 + 4 + 6 + 6 + 6 + 4 + 4
    lr = ssdvgg.LearningRate([0.1, 0.01], [10])
    assert lr.values == [0.1, 0.01] and lr.boundaries == [10]
    with pytest.raises(ValueError):
        ssdvgg.LearningRate([0.1], [10])


def test_bench_keeps_stdout_to_one_json_line():
    """bench.py's contract is ONE JSON line on stdout; libraries print there as well (RCCL's version banner).  While the
    benchmark runs file descriptor 1 points at stderr, the line itself goes to the real stdout."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench; q = bench._QuietStdout(); "
            "os.write(1, b'library banner\\n'); print('python noise'); q.emit(json.dumps({'value': 1})); os.write(1, b'late noise\\n')") % root
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {'value': 1}, r.stdout
    assert 'library banner' in r.stderr and 'python noise' in r.stderr and 'late noise' in r.stderr
