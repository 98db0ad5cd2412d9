"""Helpers for the -m gpu parity tests: device buffers come from torch (plumbing), every
computation goes through the C ABI of libssdvgg_hip.so."""
import numpy as np
import torch

from ssd_tensorflow_amd import _lib
from ssd_tensorflow_amd._lib import lib, check

DEV = 'cuda:0'


def dev(a):
    """numpy -> contiguous device tensor"""
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def ptr(t):
    return None if t is None else t.data_ptr()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def rel_err(got, ref):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    denom = np.sqrt((ref ** 2).sum()) + 1e-30
    return float(np.sqrt(((got - ref) ** 2).sum()) / denom)


def max_rel(got, ref):
    """max |got-ref| / max|ref| : scale-aware max error"""
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def same_pad(n, k, s, d=1):
    keff = (k - 1) * d + 1
    out = -(-n // s)
    tot = max((out - 1) * s + keff - n, 0)
    return tot // 2, out


def conv_geom(hi, wi, k, stride, dil, padding):
    """(pad_h, pad_w, ho, wo) with TF semantics; padding 'SAME' | 'VALID' | 'BR1' (tf.pad +1 then VALID)"""
    if padding == 'SAME':
        ph, ho = same_pad(hi, k, stride, dil)
        pw, wo = same_pad(wi, k, stride, dil)
        return ph, pw, ho, wo
    extra = 1 if padding == 'BR1' else 0
    keff = (k - 1) * dil + 1
    return 0, 0, (hi + extra - keff) // stride + 1, (wi + extra - keff) // stride + 1
