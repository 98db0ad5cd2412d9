"""ctypes binding of libssdvgg_hip.so (include/ssdvgg_hip.h).

The library is the product; there is no Python or CPU fallback.  If it is missing or
does not load, importing this module raises, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SSD_LIB: a build of the SAME sources with other compile-time definitions, for same-box A/Bs -- csrc/Makefile VARIANT)
LIB_PATH = os.environ.get('SSD_LIB') or os.path.join(_HERE, 'libssdvgg_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
        f'or `make -C {os.path.join(_HERE, "csrc")}` (hipcc, gfx950). There is no CPU fallback.')

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME
# libamdhip64.so.7) + HSA runtime.  Loading it FIRST makes this library's DT_NEEDED
# libamdhip64.so.7 resolve to the copy torch uses, so device pointers, streams and the
# device itself are shared.  (Loaded the other way round the process ends up with two HSA
# runtimes and the second one sees no device.)  A pure C consumer without torch simply
# gets /opt/rocm's runtime.
try:
    import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)
except ImportError:     # pragma: no cover
    torch = None

lib = C.CDLL(LIB_PATH)

p_f32 = C.POINTER(C.c_float)
p_f64 = C.POINTER(C.c_double)
p_i32 = C.POINTER(C.c_int)
p_i64 = C.POINTER(C.c_longlong)
vp = C.c_void_p
i32 = C.c_int
f32 = C.c_float
sz = C.c_size_t
cstr = C.c_char_p
handle = C.c_void_p

# name -> (restype, argtypes); every symbol include/ssdvgg_hip.h declares
SIGNATURES = {
    'ssd_last_error': (cstr, []),
    'ssd_version': (cstr, []),
    'ssd_preset_info': (i32, [cstr, p_i32, p_i32, p_i32, p_i32]),
    'ssd_preset_map': (i32, [cstr, i32, p_i32, p_f64, p_i32]),
    'ssd_anchors': (i32, [cstr, i32, vp]),
    'ssd_anchors_abs': (i32, [cstr, i32, vp]),
    'ssd_jaccard_overlap': (i32, [i32, vp, vp, i32, vp]),
    'ssd_encode_labels': (i32, [cstr, i32, i32, vp, vp, vp, i32, vp]),
    'ssd_encode_labels_dev': (i32, [cstr, i32, i32, vp, vp, vp, i32, vp, vp]),
    'ssd_encode_labels_ws_bytes': (sz, [i32]),
    'ssd_encode_labels_resident': (i32, [cstr, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp]),
    'ssd_decode_nms': (i32, [cstr, i32, i32, vp, i32, f32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    'ssd_decode_nms_ws_bytes': (sz, [cstr, i32]),
    'ssd_decode_nms_dev': (i32, [cstr, i32, vp, vp, i32, f32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    'ssd_anchors_dev': (i32, [cstr, vp, vp, vp]),
    'ssd_average_precision': (i32, [i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, C.c_double, vp, vp]),
    'ssd_arena_floats': (sz, [cstr, i32]),
    'ssd_augment_ws_bytes': (sz, [i32, i32, i32]),
    'ssd_augment_batch_dev': (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    'ssd_sampler_trials': (i32, [vp, i32, vp, vp, i32, i32, vp, i32, vp, vp]),
    'ssd_create': (i32, [cstr, i32, i32, i32, i32, C.c_ulonglong, vp, vp, vp, C.POINTER(handle)]),
    'ssd_create_dtype': (i32, [cstr, i32, i32, i32, i32, C.c_ulonglong, vp, vp, vp, i32, C.POINTER(handle)]),
    'ssd_get_dtype': (i32, [handle, p_i32]),
    'ssd_destroy': (i32, [handle]),
    'ssd_set_stream': (i32, [handle, vp]),
    'ssd_num_variables': (i32, [handle]),
    'ssd_variable_info': (i32, [handle, i32, C.c_char_p, i32, p_i32, p_i32]),
    'ssd_load_variable': (i32, [handle, cstr, vp, sz]),
    'ssd_save_variable': (i32, [handle, cstr, vp, sz]),
    'ssd_save_gradient': (i32, [handle, cstr, vp, sz]),
    'ssd_save_momentum': (i32, [handle, cstr, vp, sz]),
    'ssd_load_momentum': (i32, [handle, cstr, vp, sz]),
    'ssd_set_optimizer': (i32, [handle, vp, vp, i32, f32, f32]),
    'ssd_get_global_step': (i32, [handle, p_i64]),
    'ssd_set_global_step': (i32, [handle, C.c_longlong]),
    'ssd_train_step': (i32, [handle, vp, vp, i32, vp, vp]),
    'ssd_eval_step': (i32, [handle, vp, vp, i32, vp, vp]),
    'ssd_infer': (i32, [handle, vp, i32, vp]),
    'ssd_forward_backward_dev': (i32, [handle, vp, vp, i32]),
    'ssd_apply_gradients_dev': (i32, [handle, f32]),
    'ssd_forward_dev': (i32, [handle, vp, vp, i32]),
    'ssd_backward_begin_dev': (i32, [handle, vp, i32]),
    'ssd_backward_next_dev': (i32, [handle, sz, i32, C.POINTER(sz), C.POINTER(sz), p_i32]),
    'ssd_backward_ranges': (i32, [handle, sz, C.POINTER(sz), C.POINTER(sz), i32, p_i32]),
    'ssd_set_wgrad_stream': (i32, [handle, vp]),
    'ssd_set_loss_normalizer': (i32, [handle, f32]),
    'ssd_null_gradients_dev': (i32, [handle]),
    'ssd_train_step_dev': (i32, [handle, vp, vp, i32]),
    'ssd_eval_step_dev': (i32, [handle, vp, vp, i32]),
    'ssd_infer_dev': (i32, [handle, vp, i32]),
    'ssd_result_dev': (i32, [handle, C.POINTER(vp)]),
    'ssd_get_result': (i32, [handle, i32, vp]),
    'ssd_get_losses': (i32, [handle, vp]),
    'ssd_get_losses_step': (i32, [handle, i32, vp]),
    'ssd_set_result_dev': (i32, [handle, vp, i32]),
    'ssd_arenas': (i32, [handle, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]),
    'ssd_detect_last': (i32, [handle, i32, f32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    'ssd_detect_last_dev': (i32, [handle, i32, f32, i32, i32, i32, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    'ssd_detect_fetch': (i32, [handle, i32, vp, vp, vp, vp, vp]),
    'ssd_detect_host': (i32, [handle, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), p_i32, p_i32]),
    'ssd_nms_boxes': (i32, [i32, i32, vp, vp, vp, C.c_double, vp, p_i32]),
    'ssd_set_overlap': (i32, [handle, i32]),
    'ssd_profile_enable': (i32, [handle, i32]),
    'ssd_profile_report': (i32, [handle, C.c_char_p, sz]),
    'ssd_activation_shape': (i32, [handle, cstr, p_i32, p_i32, p_i32]),
    'ssd_activation': (i32, [handle, cstr, i32, vp, sz]),
    'ssd_op_conv2d_fwd': (i32, [vp, vp, vp, vp] + [i32] * 14 + [vp]),
    'ssd_op_conv2d_dgrad': (i32, [vp, vp, vp, vp, i32] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_wgrad_ws_floats': (sz, [i32] * 13),
    'ssd_op_conv2d_wgrad': (i32, [vp, vp, vp, vp, vp, f32, vp] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_wino_ws_floats': (sz, [i32] * 13),
    'ssd_op_conv2d_wino_bits_words': (sz, [i32] * 13),
    'ssd_op_conv2d_wino_fwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32] + [i32] * 14 + [vp]),
    'ssd_op_conv2d_wino_dgrad': (i32, [vp, vp, vp, vp, vp, i32, vp, i32, i32, vp, i32] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_wino_wgrad': (i32, [vp, vp, vp, vp, vp, f32, vp, i32] + [i32] * 13 + [vp]),
    'ssd_op_cast_filter': (i32, [vp, vp, vp, i32, i32, i32, vp]),
    'ssd_op_conv2d_fwd_bf16': (i32, [vp, vp, vp, vp, i32] + [i32] * 14 + [vp]),
    'ssd_op_conv2d_dgrad_bf16': (i32, [vp, vp, vp, vp, i32] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_fwd_bf16_chain': (i32, [vp, vp, vp, vp, i32] + [i32] * 14 + [vp]),
    'ssd_op_conv2d_dgrad_bf16_chain': (i32, [vp, vp, vp, vp, i32] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_wgrad_bf16_direct': (i32, [vp, vp, vp, vp, vp, f32] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_wgrad_bf16_ws_floats': (sz, [i32] * 13),
    'ssd_op_conv2d_wgrad_bf16': (i32, [vp, vp, vp, vp, vp, f32, vp] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_first_fwd_bf16': (i32, [vp, vp, vp, vp] + [i32] * 14 + [vp]),
    'ssd_op_conv2d_first_wgrad_bf16_ws_floats': (sz, [i32] * 13),
    'ssd_op_conv2d_first_wgrad_bf16': (i32, [vp, vp, vp, vp, vp, f32, vp] + [i32] * 13 + [vp]),
    'ssd_op_maxpool_rec_fwd': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    'ssd_op_maxpool_rec_bwd': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'ssd_op_conv2d_fwd_pool': (i32, [vp, vp, vp, vp, vp] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_fwd_pool_bf16': (i32, [vp, vp, vp, vp, vp] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_dgrad_unpool': (i32, [vp, vp, vp, vp, i32, i32] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_dgrad_unpool_bf16': (i32, [vp, vp, vp, vp, i32, i32] + [i32] * 13 + [vp]),
    'ssd_op_conv2d_dgrad_first_wgrad_bf16_ws_floats': (sz, [i32, i32, i32]),
    'ssd_op_conv2d_dgrad_first_wgrad_bf16': (i32, [vp, vp, vp, vp, vp, vp, vp, f32, vp, i32, i32, i32, vp]),
    'ssd_pool_fusion': (i32, [handle, p_i32, i32, p_i32]),
    'ssd_debug_set_ablate': (i32, [cstr]),
    'ssd_op_maxpool_fwd': (i32, [vp, vp] + [i32] * 10 + [vp]),
    'ssd_op_maxpool_bwd': (i32, [vp, vp, vp, i32, i32] + [i32] * 10 + [vp]),
    'ssd_op_clock_monitor': (i32, [vp, i32, C.c_uint, vp]),
    'ssd_grads_to_bf16': (i32, [i32, vp, vp, sz, vp]),
    'ssd_grads_from_bf16': (i32, [i32, vp, vp, sz, vp]),
    'ssd_op_l2norm_fwd': (i32, [vp, vp, vp, i32, i32, vp]),
    'ssd_op_l2norm_bwd_ws_floats': (sz, [i32, i32]),
    'ssd_op_l2norm_bwd': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)        # AttributeError here = the .so does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


# GPU ordinal of the free functions (anchors, label encoding, decode + NMS, AP).  One per process, like the
# handle-owning SSDVGG's Session.device; a multi-GPU driver sets it once to its LOCAL_RANK (train.py does).
# Every C entry point makes its GPU current for the call only and restores the caller's device.
_device = 0


def set_device(device):
    global _device
    _device = int(device)


def device():
    return _device


def last_error():
    return (lib.ssd_last_error() or b'').decode()


def check(rc, exc=RuntimeError):
    """0 = ok; anything else raises with the library's message."""
    if rc != 0:
        raise exc(last_error())


def np_ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    if not a.flags['C_CONTIGUOUS']:
        raise ValueError('array must be C-contiguous')
    return a.ctypes.data_as(C.c_void_p)
