"""ctypes binding of libuniver_hip.so -- the ONLY way the Python layer classes reach the GPU.

Every prototype below mirrors include/univer_hip.h.  There is no fallback: if the shared library
is missing or a call fails, HipError is raised (a GPU run must never silently compute on the host).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = 'libuniver_hip.so'

F32, F64, F16 = 0, 1, 2
DP_UNIQUE_ID_BYTES = 128


def f16_scaled(k):
    """UOCR_F16_SCALED(k): binary16 activations whose gradients carry the factor 2^k."""
    return F16 | (int(k) << 8)
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3
LOSS_DICE, LOSS_JACCARD = 0, 1


class HipError(RuntimeError):
    pass


def lib_path():
    return os.path.join(os.path.dirname(_HERE), _LIB_NAME)


def dtype_code(np_dtype):
    import numpy as np
    dt = np.dtype(np_dtype)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    if dt == np.float16:
        return F16
    raise HipError(f'unsupported dtype {dt}: the HIP backend stores float16 / float32 / float64')


_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
_ctx = C.c_void_p

# name -> argument types (after the implicit return type int)
_PROTOS = {
    'uocr_ctx_create': [_i, _sz, C.POINTER(_ctx)],
    'uocr_ctx_create_cu_mask': [_i, _sz, C.POINTER(C.c_uint32), _i, C.POINTER(_ctx)],
    'uocr_ctx_destroy': [_ctx],
    'uocr_ctx_set_stream': [_ctx, _vp],
    'uocr_ctx_reserve_workspace': [_ctx, _sz],
    'uocr_ctx_set_option': [_ctx, C.c_char_p, _i],
    'uocr_malloc': [_ctx, _sz, C.POINTER(_vp)],
    'uocr_free': [_ctx, _vp],
    'uocr_memset_zero': [_ctx, _vp, _sz],
    'uocr_h2d': [_ctx, _vp, _vp, _sz],
    'uocr_d2h_sync': [_ctx, _vp, _vp, _sz],
    'uocr_d2d': [_ctx, _vp, _vp, _sz],
    'uocr_stream_sync': [_ctx],
    'uocr_event_create': [C.POINTER(_vp)],
    'uocr_event_destroy': [_vp],
    'uocr_event_record': [_ctx, _vp],
    'uocr_stream_wait_event': [_ctx, _vp],
    'uocr_event_elapsed_ms_sync': [_vp, _vp, C.POINTER(C.c_float)],
    'uocr_event_synchronize': [_vp],
    'uocr_graph_begin_capture': [_ctx],
    'uocr_graph_end_capture': [_ctx, C.POINTER(_vp)],
    'uocr_graph_launch': [_ctx, _vp],
    'uocr_graph_destroy': [_vp],
    'uocr_ctx_set_loss_snapshot': [_ctx, _vp, _i, _vp, _i, _vp],
    'uocr_wgrad_defer_begin': [_ctx],
    'uocr_wgrad_defer_flush': [_ctx, _i],
    'uocr_device_info': [_ctx, C.c_char_p, _sz, C.POINTER(_i), C.POINTER(_sz)],
    'uocr_conv2d_fwd': [_ctx, _i, _vp, _vp, _vp, _vp] + [_i] * 13 + [_d, _i, _i, _d],
    'uocr_conv2d_bwd_data': [_ctx, _i, _vp, _vp, _vp] + [_i] * 13 + [_vp, _i, _d],
    'uocr_conv2d_bwd_weight': [_ctx, _i, _vp, _vp, _vp, _vp] + [_i] * 13 + [_d, _i, _i],
    'uocr_conv_pair_fwd': [_ctx, _i] + [_vp] * 6 + [_i] * 4 + [_d, _i, _i, _d, _i],
    'uocr_conv_pair_bwd': [_ctx, _i] + [_vp] * 11 + [_i] * 4 + [_d, _i, _i, _d, _i, _i],
    'uocr_upconv2x_fwd': [_ctx, _i, _vp, _vp, _vp, _vp] + [_i] * 9 + [_i, _i, _d, _vp],
    'uocr_upconv2x_bwd_data': [_ctx, _i, _vp, _vp, _vp] + [_i] * 9 + [_vp, _i, _d, _vp],
    'uocr_upconv2x_bwd_weight': [_ctx, _i, _vp, _vp, _vp, _vp] + [_i] * 9 + [_i, _i],
    'uocr_maxpool2d_fwd': [_ctx, _i, _vp, _vp, _vp] + [_i] * 12,
    'uocr_maxpool2d_bwd': [_ctx, _i, _vp, _vp, _vp] + [_i] * 12,
    'uocr_upsample2d_fwd': [_ctx, _i, _vp, _vp] + [_i] * 6,
    'uocr_upsample2d_bwd': [_ctx, _i, _vp, _vp] + [_i] * 6,
    'uocr_act_fwd': [_ctx, _i, _i, _d, _vp, _vp, _sz],
    'uocr_act_bwd': [_ctx, _i, _i, _d, _vp, _vp, _vp, _sz],
    'uocr_act_bwd_from_output': [_ctx, _i, _i, _d, _vp, _vp, _vp, _sz],
    'uocr_dense_fwd': [_ctx, _i, _vp, _vp, _vp, _i, _i, _i],
    'uocr_dense_bwd': [_ctx, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    'uocr_dense_fwd_act': [_ctx, _i, _vp, _vp, _vp, _i, _i, _i, _i, _d],
    'uocr_dense_bwd_act': [_ctx, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d],
    'uocr_fixed_width_fwd': [_ctx, _i, _vp, _vp] + [_i] * 5,
    'uocr_fixed_width_bwd': [_ctx, _i, _vp, _vp] + [_i] * 5,
    'uocr_copy_2d': [_ctx, _i, _vp, _sz, _vp, _sz, _sz, _sz],
    'uocr_add': [_ctx, _i, _vp, _vp, _vp, _sz],
    'uocr_axpy': [_ctx, _i, _d, _vp, _vp, _sz],
    'uocr_scale': [_ctx, _i, _d, _vp, _sz],
    'uocr_fill': [_ctx, _i, _vp, _d, _sz],
    'uocr_convert': [_ctx, _i, _vp, _i, _vp, _sz],
    'uocr_u8_to_float': [_ctx, _i, _vp, _vp, _d, _sz],
    'uocr_seg_loss': [_ctx, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    'uocr_softmax_ce': [_ctx, _i, _vp, _vp, _vp, _vp, _i, _i],
    'uocr_sigmoid_ce': [_ctx, _i, _vp, _vp, _vp, _vp, _i, _sz],
    'uocr_l2_reg': [_ctx, _i, _vp, _vp, _sz, _d, _vp, _i],
    'uocr_l1_reg': [_ctx, _i, _vp, _vp, _sz, _d, _vp, _i],
    'uocr_adam_step': [_ctx, _i, _vp, _vp, _vp, _vp, _sz, _d, _d, _d, _d],
    'uocr_momentum_step': [_ctx, _i, _vp, _vp, _vp, _sz, _d, _d],
    'uocr_momentum_step_fused': [_ctx, _i, _vp, _vp, _vp, _sz, _d, _d, _i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_int), C.POINTER(C.c_double), _vp, _i, _vp],
    'uocr_adam_step_fused': [_ctx, _i, _vp, _vp, _vp, _vp, _sz, _d, _d, _d, _d, _i, C.POINTER(C.c_longlong),
                             C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_double), _vp, _i, _vp],
    'uocr_rmsprop_step': [_ctx, _i, _vp, _vp, _vp, _sz, _d, _d, _d],
    'uocr_has_nan': [_ctx, _i, _vp, _sz, _vp],
    'uocr_dp_init': [_ctx, _i, _i, _vp],
    'uocr_dp_info': [_ctx, C.POINTER(_i), C.POINTER(_i)],
    'uocr_dp_allreduce_sum': [_ctx, _vp, _sz, _i],
    'uocr_dp_broadcast': [_ctx, _vp, _sz, _i, _i],
    'uocr_dp_finalize': [_ctx],
}
# declared in the header with a non-int return type
_SPECIAL = {
    'uocr_abi_version': (C.c_int, []),
    'uocr_dp_get_unique_id': (C.c_int, [_vp]),
    'uocr_dp_version': (C.c_int, [C.POINTER(C.c_int)]),
    'uocr_last_error': (C.c_char_p, [_ctx]),
    'uocr_ctx_get_stream': (C.c_void_p, [_ctx]),
}

ABI_SYMBOLS = sorted(list(_PROTOS) + list(_SPECIAL))

_lib = None


def get_lib():
    """Load libuniver_hip.so (once) and attach prototypes.  Raises HipError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise HipError(
            f'{path} not found: build it with ./build.sh (or __graft_entry__.build()); '
            f'the HIP backend has no host fallback')
    try:
        import torch  # noqa: F401  (loads the process-wide HIP runtime first)
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    for name, (restype, argtypes) in _SPECIAL.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
