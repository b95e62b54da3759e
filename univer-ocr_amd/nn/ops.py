"""Kernel wrappers: DeviceArray in, DeviceArray out, every byte of arithmetic in libuniver_hip.so.

This is the seam that replaces the reference's per-layer `_forward_gpu/_backward_gpu` closures
(layers/layers.py:169-197) and its CuPy expressions.  Output shapes follow the reference's
get_output_shapes; arrays returned are always NEW allocations, as in the reference.
"""
import math

import numpy as np

from ..hip import lib as hiplib
from ..hip.lib import HipError
from .gpu import CP, DeviceArray, DeviceScalar

ACT_CODES = {None: hiplib.ACT_NONE, 'relu': hiplib.ACT_RELU, 'leaky': hiplib.ACT_LEAKY,
             'sigmoid': hiplib.ACT_SIGMOID}


def _rt():
    return CP.runtime()


def _same_dtype(*arrays, grad=None):
    """dtype code of a call.  The first array is an ACTIVATION tensor and decides: float32 / float64 calls need
    every array in that type; float16 calls (UOCR_F16: binary16 activations, float32 parameters) take float16
    and float32 arrays together.  `grad`: the incoming activation gradient of a backward op -- the power-of-two
    factor it carries (DeviceArray.gscale) travels in the code's scale bits (UOCR_F16_SCALED)."""
    base = arrays[0].code & 0xff
    for a in arrays[1:]:
        other = a.code & 0xff
        if other != base and not (base == hiplib.F16 and other == hiplib.F32):
            raise HipError('mixed dtypes in one op: ' + ', '.join(str(x.dtype) for x in arrays))
    if base == hiplib.F16 and grad is not None:
        return hiplib.f16_scaled(grad.gscale)
    return base


def _like_grad(array, dy):
    """`array` is the activation gradient a backward op derived from `dy`: it carries the same factor."""
    array.gscale = dy.gscale
    return array


def f16_grad_scale_log2(kind, n):
    """log2 of the factor a loss kernel gives its float16 gradient so that it sits around 2^-4 .. 1 instead of
    around 1 / n (n = pixels per image for the segmentation losses, rows for the cross-entropies), far below
    binary16's normal range for a 2-Mpixel page.  CP.f16_grad_scale_log2 overrides."""
    if CP.f16_grad_scale_log2 is not None:
        return int(CP.f16_grad_scale_log2)
    k = int(math.floor(math.log2(max(1, n)))) - (4 if kind == 'seg' else 1)
    return max(0, min(24, k))


def as_device(x):
    """Accept host arrays at the layer boundary (the reference's layers accept whatever CP.cp is)."""
    if isinstance(x, DeviceArray):
        return x
    return CP.copy(x)


# ---- Convolutional2D ---------------------------------------------------------------------------
def conv_out_hw(h, w, ks, stride, padding):
    """convolutional.py:290-301."""
    oh = math.floor((h + 2 * padding[0] - (ks[0] - 1) - 1) / stride[0] + 1)
    ow = math.floor((w + 2 * padding[1] - (ks[1] - 1) - 1) / stride[1] + 1)
    return oh, ow


def _conv_dims(x_shape, w_shape, stride, padding):
    n, h, wd, cin = x_shape
    kh, kw, wcin, cout = w_shape
    if cin != wcin:
        raise AssertionError(f'channels mismatch: input has {cin}, weights expect {wcin}')
    oh, ow = conv_out_hw(h, wd, (kh, kw), stride, padding)
    if oh <= 0 or ow <= 0:
        raise HipError(f'convolution output is empty for input {x_shape}, kernel {(kh, kw)}')
    return (n, h, wd, cin, cout, kh, kw, stride[0], stride[1], padding[0], padding[1], oh, ow)


def conv2d_fwd(x, w, b, stride, padding, pad_value=0.0, bias=True, act=None, alpha=0.0):
    dims = _conv_dims(x.shape, w.shape, stride, padding)
    code = _same_dtype(x, w, b)
    y = CP.empty((dims[0], dims[11], dims[12], dims[4]), x.dtype)
    _rt().call('uocr_conv2d_fwd', code, x.ptr, w.ptr, b.ptr, y.ptr, *dims, float(pad_value),
               int(bool(bias)), ACT_CODES[act], float(alpha))
    return y


def conv2d_bwd_data(dy, w, x_shape, stride, padding, x_act=None, act=None, alpha=0.0):
    """dx of the conv; with `x_act` (the conv's input = output of a fused LeakyReLU / Sigmoid) the
    kernel stores dx * act'(x_act), i.e. the gradient w.r.t. that activation's INPUT."""
    dims = _conv_dims(x_shape, w.shape, stride, padding)
    code = _same_dtype(dy, w, grad=dy) if x_act is None else _same_dtype(dy, w, x_act, grad=dy)
    if dy.shape != (dims[0], dims[11], dims[12], dims[4]):
        raise AssertionError(f'grad shape {dy.shape} does not match the layer output')
    if x_act is not None and x_act.shape != tuple(x_shape):
        raise AssertionError(f'activation tensor {x_act.shape} != conv input {tuple(x_shape)}')
    dx = _like_grad(CP.empty(x_shape, dy.dtype), dy)
    _rt().call('uocr_conv2d_bwd_data', code, dy.ptr, w.ptr, dx.ptr, *dims,
               None if x_act is None else x_act.ptr, ACT_CODES[act if x_act is not None else None], float(alpha))
    return dx


def conv2d_bwd_weight(x, dy, dw, db, stride, padding, pad_value=0.0, bias=True, accumulate=True):
    dims = _conv_dims(x.shape, dw.shape, stride, padding)
    code = _same_dtype(x, dy, dw, db, grad=dy)
    with _rt().side(x, dy):               # nothing on the lane reads dw before the end of the backward pass
        _rt().call('uocr_conv2d_bwd_weight', code, x.ptr, dy.ptr, dw.ptr, db.ptr, *dims, float(pad_value),
                   int(bool(bias)), int(bool(accumulate)))
    _rt().keep(x, dy)                     # (read at the flush if the call was deferred: Runtime.defer_wgrad)


def conv_pair_fwd(x, w1, b1, w2, b2, pad_value1=0.0, bias1=True, bias2=True, alpha=0.01, act2=hiplib.ACT_NONE):
    """conv3x3(1->16) + LeakyReLU + conv3x3(16->1) [+ Sigmoid] as one kernel (float32, see univer_hip.h)."""
    n, h, wd, _ = x.shape
    code = _same_dtype(x, w1, b1, w2, b2)
    y = CP.empty((n, h, wd, 1), x.dtype)
    _rt().call('uocr_conv_pair_fwd', code, x.ptr, w1.ptr, b1.ptr, w2.ptr, b2.ptr, y.ptr, n, h, wd, w1.shape[3],
               float(pad_value1), int(bool(bias1)), int(bool(bias2)), float(alpha), int(act2))
    return y


def conv_pair_bwd(x, y, dy, w1, b1, w2, dw1, db1, dw2, db2, pad_value1=0.0, bias1=True, bias2=True, alpha=0.01,
                  act2=hiplib.ACT_NONE, need_dx=True, accumulate=True):
    n, h, wd, _ = x.shape
    code = _same_dtype(x, y, dy, w1, b1, w2, dw1, db1, dw2, db2, grad=dy)
    dx = _like_grad(CP.empty(x.shape, x.dtype), dy) if need_dx else None
    _rt().call('uocr_conv_pair_bwd', code, x.ptr, y.ptr, dy.ptr, w1.ptr, b1.ptr, w2.ptr, dw1.ptr, db1.ptr, dw2.ptr,
               db2.ptr, dx.ptr if need_dx else None, n, h, wd, w1.shape[3], float(pad_value1), int(bool(bias1)),
               int(bool(bias2)), float(alpha), int(act2), int(bool(accumulate)))
    return dx


def _up_dims(x_low_shape, w_shape, padding):
    n, hl, wl, cin = x_low_shape
    kh, kw, wcin, cout = w_shape
    assert wcin == cin
    return n, hl, wl, cin, cout, kh, kw, padding[0], padding[1]


def upconv2x_fwd(x_low, w, b, padding, bias=True, act=None, alpha=0.0, weff=None):
    """Upsample2D(2) + conv (stride 1) on the low-res tensor (float32 5x5 4->4 only, see univer_hip.h).
    `weff`: a float32 array of 576 that receives the per-phase weights for upconv2x_bwd_data of the same step."""
    dims = _up_dims(x_low.shape, w.shape, padding)
    code = _same_dtype(x_low, w, b)
    y = CP.empty((dims[0], 2 * dims[1], 2 * dims[2], dims[4]), x_low.dtype)
    _rt().call('uocr_upconv2x_fwd', code, x_low.ptr, w.ptr, b.ptr, y.ptr, *dims, int(bool(bias)), ACT_CODES[act],
               float(alpha), None if weff is None else weff.ptr)
    return y


def upconv2x_bwd_data(dy, w, x_low_shape, padding, x_act=None, act=None, alpha=0.0, weff=None):
    """`weff`: what upconv2x_fwd of this layer wrote in the same step (w unchanged since), or None."""
    dims = _up_dims(x_low_shape, w.shape, padding)
    code = _same_dtype(dy, w, grad=dy)
    dx = _like_grad(CP.empty(x_low_shape, dy.dtype), dy)
    _rt().call('uocr_upconv2x_bwd_data', code, dy.ptr, w.ptr, dx.ptr, *dims, None if x_act is None else x_act.ptr,
               ACT_CODES[act if x_act is not None else None], float(alpha), None if weff is None else weff.ptr)
    return dx


def upconv2x_bwd_weight(x_low, dy, dw, db, padding, bias=True, accumulate=True):
    dims = _up_dims(x_low.shape, dw.shape, padding)
    code = _same_dtype(x_low, dy, dw, db, grad=dy)
    with _rt().side(x_low, dy):
        _rt().call('uocr_upconv2x_bwd_weight', code, x_low.ptr, dy.ptr, dw.ptr, db.ptr, *dims, int(bool(bias)),
                   int(bool(accumulate)))


# ---- MaxPool2D ------------------------------------------------------------------------------------
def pool_out_hw(h, w, ks, stride, padding, ceil_mode):
    """maxpool.py:204-216."""
    rnd = math.ceil if ceil_mode else math.floor
    return (rnd((h + 2 * padding[0] - (ks[0] - 1) - 1) / stride[0] + 1),
            rnd((w + 2 * padding[1] - (ks[1] - 1) - 1) / stride[1] + 1))


def maxpool2d_fwd(x, ks, stride, padding, ceil_mode=False):
    n, h, wd, c = x.shape
    oh, ow = pool_out_hw(h, wd, ks, stride, padding, ceil_mode)
    y = CP.empty((n, oh, ow, c), x.dtype)
    mask = CP.empty((n, ks[0] * oh, ks[1] * ow, c), np.uint8)
    _rt().call('uocr_maxpool2d_fwd', x.code, x.ptr, y.ptr, mask.ptr, n, h, wd, c, ks[0], ks[1], stride[0],
               stride[1], padding[0], padding[1], oh, ow)
    return y, mask


def maxpool2d_bwd(dy, mask, x_shape, ks, stride, padding):
    n, h, wd, c = x_shape
    oh, ow = dy.shape[1], dy.shape[2]
    dx = _like_grad(CP.empty(x_shape, dy.dtype), dy)
    _rt().call('uocr_maxpool2d_bwd', dy.code, dy.ptr, mask.ptr, dx.ptr, n, h, wd, c, ks[0], ks[1], stride[0],
               stride[1], padding[0], padding[1], oh, ow)
    return dx


# ---- Upsample2D -----------------------------------------------------------------------------------
def upsample2d_fwd(x, scale):
    n, h, wd, c = x.shape
    y = CP.empty((n, h * scale[0], wd * scale[1], c), x.dtype)
    _rt().call('uocr_upsample2d_fwd', x.code, x.ptr, y.ptr, n, h, wd, c, scale[0], scale[1])
    return y


def upsample2d_bwd(dy, x_shape, scale):
    n, h, wd, c = x_shape
    if dy.shape != (n, h * scale[0], wd * scale[1], c):
        raise AssertionError(f'grad shape {dy.shape} does not match the upsampled shape')
    dx = _like_grad(CP.empty(x_shape, dy.dtype), dy)
    _rt().call('uocr_upsample2d_bwd', dy.code, dy.ptr, dx.ptr, n, h, wd, c, scale[0], scale[1])
    return dx


# ---- activations -----------------------------------------------------------------------------------
def act_fwd(kind, x, alpha=0.0):
    y = CP.empty(x.shape, x.dtype)
    _rt().call('uocr_act_fwd', x.code, ACT_CODES[kind], float(alpha), x.ptr, y.ptr, x.size)
    return y


def act_bwd(kind, x, dy, alpha=0.0):
    code = _same_dtype(x, dy)
    if x.size != dy.size:
        raise AssertionError(f'grad size {dy.shape} != input size {x.shape}')
    dx = _like_grad(CP.empty(dy.shape, dy.dtype), dy)
    _rt().call('uocr_act_bwd', code, ACT_CODES[kind], float(alpha), x.ptr, dy.ptr, dx.ptr, x.size)
    return dx


def act_bwd_from_output(kind, y, dy, alpha=0.0):
    code = _same_dtype(y, dy)
    dx = _like_grad(CP.empty(dy.shape, dy.dtype), dy)
    _rt().call('uocr_act_bwd_from_output', code, ACT_CODES[kind], float(alpha), y.ptr, dy.ptr, dx.ptr, y.size)
    return dx


# ---- FullyConnected ---------------------------------------------------------------------------------
def dense_fwd(x, w, act=None, alpha=0.0):
    """y = act([x, 1] . w): `act` is the LeakyRelu / Sigmoid that follows the layer (None: plain dense)."""
    m, n_in = x.shape
    if w.shape[0] != n_in + 1:
        raise AssertionError(f'weights {w.shape} do not fit input {x.shape} (bias row included)')
    y = CP.empty((m, w.shape[1]), x.dtype)
    _rt().call('uocr_dense_fwd_act', _same_dtype(x, w), x.ptr, w.ptr, y.ptr, m, n_in, w.shape[1], ACT_CODES[act],
               float(alpha))
    return y


def dense_bwd(x, w, dy, dw, accumulate=True, need_dx=True, x_act=None, x_alpha=0.0):
    """`x_act`: x is the output of that (fused) activation and dx is wanted w.r.t. the activation's INPUT."""
    m, n_in = x.shape
    n_out = w.shape[1]
    code = _same_dtype(x, w, dy, dw, grad=dy)
    dx = _like_grad(CP.empty((m, n_in), dy.dtype), dy) if need_dx else None
    rt = _rt()
    if rt.side_on and need_dx:            # dw on the lane's side stream, dx (the critical path) on the lane
        with rt.side(x, dy):
            rt.call('uocr_dense_bwd_act', code, x.ptr, w.ptr, dy.ptr, None, dw.ptr, m, n_in, n_out,
                    int(bool(accumulate)), ACT_CODES[None], 0.0)
        rt.call('uocr_dense_bwd_act', code, x.ptr, w.ptr, dy.ptr, dx.ptr, None, m, n_in, n_out, 0, ACT_CODES[x_act],
                float(x_alpha))
        return dx
    rt.call('uocr_dense_bwd_act', code, x.ptr, w.ptr, dy.ptr, dx.ptr if need_dx else None, dw.ptr, m, n_in,
            n_out, int(bool(accumulate)), ACT_CODES[x_act], float(x_alpha))
    rt.keep(x, dy)
    return dx


# ---- Conv2DToBatchedFixedWidthed + Flatten + FullyConnected as ONE implicit GEMM -------------------------
# The windows layer (convolutional.py:330-373) copies every `width`-column window of (n, h, W, c) into its own
# batch entry, Flatten strings it out as (h, j, c) and the dense layer multiplies by w[(h, j, c), n_out]: that is a
# convolution with a (h, width) kernel, padding (0, width // 2) and the output cut to W columns, whose weights
# are the dense layer's rows and whose bias is its last row.  Run through the conv entry points (the implicit-GEMM
# loaders gather the windows on the fly) the 8x larger windows tensor and its gradient never exist.
def _windows_dims(x_shape, w, width):
    n, h, wd, c = x_shape
    n_in = h * width * c
    if w.shape[0] != n_in + 1:
        raise AssertionError(f'weights {w.shape} do not fit {width}-column windows of {tuple(x_shape)} (bias row included)')
    if wd < width:
        raise AssertionError(f'Input width must be >= than output width, found: {wd} < {width}')
    dims = (n, h, wd, c, w.shape[1], h, width, 1, 1, 0, width // 2, 1, wd)
    return dims, n_in * w.shape[1] * w.t.element_size()


def windows_dense_fwd(x, w, width, act=None, alpha=0.0):
    """(n, h, W, c) -> (n * W, n_out): fixed-width windows, flatten and dense (+ its activation) in one kernel."""
    dims, bias_off = _windows_dims(x.shape, w, width)
    y = CP.empty((dims[0] * dims[2], dims[4]), x.dtype)
    _rt().call('uocr_conv2d_fwd', _same_dtype(x, w), x.ptr, w.ptr, w.ptr + bias_off, y.ptr, *dims, 0.0, 1,
               ACT_CODES[act], float(alpha))
    return y


def windows_dense_bwd(x, w, dy, dw, width, accumulate=True, x_act=None, act=None, alpha=0.0):
    """dw (bias row included) and the gradient w.r.t. x (times act'(x) when x is the output of a fused
    activation, as conv2d_bwd_data)."""
    dims, bias_off = _windows_dims(x.shape, w, width)
    code = _same_dtype(x, w, dy, dw, grad=dy)
    if dy.shape != (dims[0] * dims[2], dims[4]):
        raise AssertionError(f'grad shape {dy.shape} does not match the layer output')
    with _rt().side(x, dy):
        _rt().call('uocr_conv2d_bwd_weight', code, x.ptr, dy.ptr, dw.ptr, dw.ptr + bias_off, *dims, 0.0, 1,
                   int(bool(accumulate)))
    _rt().keep(x, dy)
    dx = _like_grad(CP.empty(x.shape, dy.dtype), dy)
    _rt().call('uocr_conv2d_bwd_data', code, dy.ptr, w.ptr, dx.ptr, *dims, None if x_act is None else x_act.ptr,
               ACT_CODES[act if x_act is not None else None], float(alpha))
    return dx


# ---- Conv2DToBatchedFixedWidthed -------------------------------------------------------------------
def fixed_width_fwd(x, width):
    n, h, wd, c = x.shape
    y = CP.empty((n * wd, h, width, c), x.dtype)
    _rt().call('uocr_fixed_width_fwd', x.code, x.ptr, y.ptr, n, h, wd, c, width)
    return y


def fixed_width_bwd(dy, x_shape, width):
    n, h, wd, c = x_shape
    dx = _like_grad(CP.empty(x_shape, dy.dtype), dy)
    _rt().call('uocr_fixed_width_bwd', dy.code, dy.ptr, dx.ptr, n, h, wd, c, width)
    return dx


# ---- Concat / sums -----------------------------------------------------------------------------------
def concat(arrays, axis=-1):
    nd = arrays[0].ndim
    axis = axis % nd
    lead = arrays[0].shape[:axis]
    rows = int(np.prod(lead)) if lead else 1
    tails = [int(np.prod(a.shape[axis:])) for a in arrays]
    out_shape = list(arrays[0].shape)
    out_shape[axis] = sum(a.shape[axis] for a in arrays)
    out = CP.empty(out_shape, arrays[0].dtype)
    total = sum(tails)
    off = 0
    esz = out.t.element_size()
    for a, cols in zip(arrays, tails):
        _same_dtype(out, a)
        _rt().call('uocr_copy_2d', out.code, out.ptr + off * esz, total, a.ptr, cols, rows, cols)
        off += cols
    return out


def split(array, shapes, axis=-1):
    """Inverse of concat: slices of `array` with the given shapes along `axis` (Concat.backward)."""
    nd = array.ndim
    axis = axis % nd
    rows = int(np.prod(array.shape[:axis])) if axis else 1
    total = int(np.prod(array.shape[axis:]))
    outs, off = [], 0
    esz = array.t.element_size()
    for shp in shapes:
        cols = int(np.prod(shp[axis:]))
        out = _like_grad(CP.empty(shp, array.dtype), array)
        _rt().call('uocr_copy_2d', out.code, out.ptr, cols, array.ptr + off * esz, total, rows, cols)
        outs.append(out)
        off += cols
    return outs


def add(a, b):
    if a.shape != b.shape:
        raise AssertionError(f'cannot add {a.shape} and {b.shape}')
    if a.gscale != b.gscale:
        raise HipError(f'cannot add gradients that carry different scales (2^{a.gscale} and 2^{b.gscale})')
    out = _like_grad(CP.empty(a.shape, a.dtype), a)
    _rt().call('uocr_add', _same_dtype(a, b), a.ptr, b.ptr, out.ptr, a.size)
    return out


def axpy(alpha, x, y):
    _rt().call('uocr_axpy', _same_dtype(x, y), float(alpha), x.ptr, y.ptr, x.size)


def scale_(x, alpha):
    _rt().call('uocr_scale', x.code, float(alpha), x.ptr, x.size)


def fill_(x, value):
    _rt().call('uocr_fill', x.code, x.ptr, float(value), x.size)


def zero_(x):
    _rt().call('uocr_memset_zero', x.ptr, x.nbytes)


def u8_to_float(src_u8, scale=1.0 / 255.0, dtype=None, out=None):
    if out is None:
        out = CP.empty(src_u8.shape, CP.dtype if dtype is None else dtype)
    elif out.shape != src_u8.shape:
        raise AssertionError(f'u8_to_float: out {out.shape} != source {src_u8.shape}')
    _rt().call('uocr_u8_to_float', out.code, src_u8.ptr, out.ptr, float(scale), out.size)
    return out


# ---- losses --------------------------------------------------------------------------------------------
def _loss_slot():
    return CP.loss_slot()


def _loss_code(pred, gt, grad, kind, n):
    """dtype code of a loss call; a float16 gradient gets its power-of-two scale here (see f16_grad_scale_log2)."""
    code = _same_dtype(pred, gt)
    if code == hiplib.F16 and grad is not None:
        grad.gscale = f16_grad_scale_log2(kind, n)
        return hiplib.f16_scaled(grad.gscale)
    return code


def _finish_loss(slot):
    return DeviceScalar(slot.t) if CP.lazy_losses else float(slot.t.item())


def seg_loss(kind, pred, gt, need_grad=True, out_act=None):
    """out_act='sigmoid': pred is a Sigmoid output and the gradient is w.r.t. the Sigmoid's input."""
    n, h, w, c = pred.shape
    if gt.shape != pred.shape:
        raise AssertionError(f'ground truth {gt.shape} != prediction {pred.shape}')
    grad = CP.empty(pred.shape, pred.dtype) if need_grad else None
    slot = _loss_slot()
    _rt().call('uocr_seg_loss', _loss_code(pred, gt, grad, 'seg', h * w), hiplib.LOSS_DICE if kind == 'dice' else hiplib.LOSS_JACCARD,
               pred.ptr, gt.ptr, grad.ptr if need_grad else None, slot.ptr, n, h * w, c, ACT_CODES[out_act])
    return _finish_loss(slot), grad


def softmax_ce(pred, gt, need_grad=True):
    m, c = pred.shape
    if gt.shape != pred.shape:
        raise AssertionError(f'ground truth {gt.shape} != prediction {pred.shape}')
    grad = CP.empty(pred.shape, pred.dtype) if need_grad else None
    slot = _loss_slot()
    _rt().call('uocr_softmax_ce', _loss_code(pred, gt, grad, 'ce', m), pred.ptr, gt.ptr, grad.ptr if need_grad else None,
               slot.ptr, m, c)
    return _finish_loss(slot), grad


def sigmoid_ce(pred, gt, need_grad=True):
    if gt.shape != pred.shape:
        raise AssertionError(f'ground truth {gt.shape} != prediction {pred.shape}')
    grad = CP.empty(pred.shape, pred.dtype) if need_grad else None
    slot = _loss_slot()
    _rt().call('uocr_sigmoid_ce', _loss_code(pred, gt, grad, 'ce', gt.shape[0]), pred.ptr, gt.ptr, grad.ptr if need_grad else None,
               slot.ptr, gt.shape[0], pred.size)
    return _finish_loss(slot), grad


def regularize(kind, w, grad, strength, slot=None, accumulate=False):
    """grad += dR/dw; returns the loss (or adds it into `slot` when given)."""
    own = slot is None
    if own:
        slot = _loss_slot()
    _rt().call('uocr_l2_reg' if kind == 'l2' else 'uocr_l1_reg', _same_dtype(w, grad), w.ptr, grad.ptr, w.size,
               float(strength), slot.ptr, int(bool(accumulate)))
    return _finish_loss(slot) if own else None


# ---- optimizers ------------------------------------------------------------------------------------------
def adam_step(w, g, v, a, lr, beta1, beta2, eps):
    _rt().call('uocr_adam_step', _same_dtype(w, g, v, a), w.ptr, g.ptr, v.ptr, a.ptr, w.size, float(lr),
               float(beta1), float(beta2), float(eps))


def _range_args(ranges):
    import ctypes as C
    n = len(ranges)
    lo = (C.c_longlong * max(n, 1))(*[r[1] for r in ranges])
    hi = (C.c_longlong * max(n, 1))(*[r[2] for r in ranges])
    kind = (C.c_int * max(n, 1))(*[{'l1': 1, 'l2': 2, 1: 1, 2: 2}[r[0][0]] for r in ranges])
    strength = (C.c_double * max(n, 1))(*[float(r[0][1]) for r in ranges])
    return n, lo, hi, kind, strength


def momentum_step_fused(w, g, v, lr, momentum, ranges, zero_grad=True, hyper=None):
    """Regularisers of `ranges` = [((kind, strength), lo, hi), ...] (at most 4) + Momentum update + gradient
    reset in one pass (univer_hip.h: uocr_momentum_step_fused).  Returns the regularisation loss.  `hyper`: a
    float64 DeviceArray {lr, momentum, -, -} the kernel reads instead of the by-value arguments (HIP graphs)."""
    n, lo, hi, kind, strength = _range_args(ranges)
    slot = _loss_slot() if n else None
    _rt().call('uocr_momentum_step_fused', _same_dtype(w, g, v), w.ptr, g.ptr, v.ptr, w.size, float(lr), float(momentum),
               n, lo, hi, kind, strength, slot.ptr if n else None, int(bool(zero_grad)),
               None if hyper is None else hyper.ptr)
    return _finish_loss(slot) if n else 0


def adam_step_fused(w, g, v, a, lr, beta1, beta2, eps, ranges, zero_grad=True, hyper=None):
    n, lo, hi, kind, strength = _range_args(ranges)
    slot = _loss_slot() if n else None
    _rt().call('uocr_adam_step_fused', _same_dtype(w, g, v, a), w.ptr, g.ptr, v.ptr, a.ptr, w.size, float(lr),
               float(beta1), float(beta2), float(eps), n, lo, hi, kind, strength, slot.ptr if n else None,
               int(bool(zero_grad)), None if hyper is None else hyper.ptr)
    return _finish_loss(slot) if n else 0


def momentum_step(w, g, v, lr, momentum):
    _rt().call('uocr_momentum_step', _same_dtype(w, g, v), w.ptr, g.ptr, v.ptr, w.size, float(lr), float(momentum))


def rmsprop_step(w, g, a, lr, rho, eps):
    _rt().call('uocr_rmsprop_step', _same_dtype(w, g, a), w.ptr, g.ptr, a.ptr, w.size, float(lr), float(rho),
               float(eps))


def has_nan(x):
    flag = CP.empty((1,), np.int32)
    _rt().call('uocr_has_nan', x.code, x.ptr, x.size, flag.ptr)
    return bool(flag.t.item())


class _Ops:
    add = staticmethod(add)


CP.ops = _Ops
