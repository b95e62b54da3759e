"""float16 storage mode (UOCR_F16, BASELINE configs[4]: 1024x2048 pages, the HBM-bound regime): activations are
binary16 in HBM, parameters / gradients of parameters / every accumulation float32 (float64 for the long
reductions), activation gradients carry a power-of-two scale (UOCR_F16_SCALED).

The checker is the float64 oracle run on the SAME binary16-rounded inputs (x16 = float64(float16(x))), so the
only differences are (a) one rounding to binary16 per stored activation element (relative 2^-11 = 4.9e-4) and
(b) float32 accumulation.  Tolerances (normalised max error, max|a-b| / max|b|), stated per check:
    a tensor stored in binary16 by ONE kernel from exact inputs ........ 1e-3
    dw / db / losses (float32 / float64 outputs, exact binary16 inputs) .. 2e-5
    a whole net (5-8 stored layers): prediction 3e-3, parameter gradients 1e-2
"""
import numpy as np
import pytest

from conftest import rel_linf
from oracle import nn_oracle as O

pytestmark = pytest.mark.gpu

TOL_STORE = 1e-3
TOL_EXACT = 2e-5


def r16(a):
    """what the device holds after an upload in float16 mode"""
    return np.asarray(a, dtype=np.float64).astype(np.float16).astype(np.float64)


@pytest.fixture
def f16():
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float16')
    yield CP
    CP.set_dtype('float32')
    CP.f16_grad_scale_log2 = None


def params32(CP, *arrays):
    return [CP.copy(a, np.float32) for a in arrays]


def test_dtype_plumbing(f16):
    from univer_ocr_amd.hip import lib as hiplib
    from univer_ocr_amd.nn.layers import Convolutional2D
    CP = f16
    x = CP.copy(np.ones((2, 3)))
    assert x.dtype == np.float16 and x.code == hiplib.F16 and x.nbytes == 12
    x.gscale = 5
    assert x.code == hiplib.f16_scaled(5) == (2 | (5 << 8)) and x.reshape(3, 2).gscale == 5
    conv = Convolutional2D((3, 3), 1, 4, padding=1)
    assert conv.w.value.dtype == np.float32 and conv.w.grad.dtype == np.float32     # float32 master weights
    assert CP.param_dtype() == np.float32


def test_elementwise_and_feed_kernels(f16):
    from univer_ocr_amd.nn import ops
    CP = f16
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 17, 19, 4))
    g = rng.standard_normal(x.shape)
    x[0, 0, 0, :2] = 0.0                                  # the >= 0 branch at exactly 0
    xd, gd = CP.copy(x), CP.copy(g)
    x16, g16 = r16(x), r16(g)
    for kind, fwd, bwd in (('relu', O.relu_fwd, O.relu_bwd),
                           ('leaky', lambda a: O.leaky_relu_fwd(a, 0.01), lambda a, b: O.leaky_relu_bwd(a, b, 0.01)),
                           ('sigmoid', O.sigmoid_fwd, O.sigmoid_bwd)):
        alpha = 0.01 if kind == 'leaky' else 0.0
        y = ops.act_fwd(kind, xd, alpha)
        assert y.dtype == np.float16
        assert rel_linf(CP.asnumpy(y), fwd(x16)) <= TOL_STORE, kind
        gd.gscale = 3
        dx = ops.act_bwd(kind, xd, gd, alpha)
        assert dx.gscale == 3                             # the scale travels with the gradient
        assert rel_linf(CP.asnumpy(dx), bwd(x16, g16)) <= TOL_STORE, kind
    gd.gscale = 0
    assert rel_linf(CP.asnumpy(ops.add(xd, gd)), x16 + g16) <= TOL_STORE
    # uint8 page feed -> binary16 (my_model/pipeline.py): 16 pixels per lane and the ragged fallback
    for count in (16 * 40, 16 * 40 + 7):
        u8 = rng.integers(0, 256, count).astype(np.uint8)
        out = ops.u8_to_float(CP.copy(u8, np.uint8), 1.0 / 255.0)
        assert out.dtype == np.float16
        ref = (u8.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float16)
        assert np.array_equal(CP.asnumpy(out), ref)
    # convert f32 <-> f16 through the ABI
    a32 = CP.copy(x, np.float32)
    half = CP.empty(x.shape, np.float16)
    CP.runtime().call('uocr_convert', a32.code, a32.ptr, half.code, half.ptr, a32.size)
    assert np.array_equal(CP.asnumpy(half), x.astype(np.float32).astype(np.float16))


# every conv of the page nets (my_model/model.py:138-247) + the generic kernel (odd shape)
CONVS = [
    # (x shape, kernel, cout, stride, padding, pad_value)
    ((2, 40, 72, 1), (5, 5), 1, (2, 2), (2, 2), 0.0),     # Paragraph down
    ((2, 40, 72, 1), (5, 5), 1, (1, 1), (2, 2), 0.0),     # Paragraph end (px kernels, aligned rows)
    ((2, 40, 72, 1), (5, 5), 4, (2, 2), (2, 2), 0.0),     # Line down_1 (tiled dw)
    ((2, 40, 72, 4), (5, 5), 4, (2, 2), (2, 2), 0.0),     # Line down_2
    ((2, 40, 72, 4), (5, 5), 4, (1, 1), (2, 2), 0.0),     # 4 -> 4 stride 1 (LDS-tiled forward)
    ((2, 40, 72, 4), (5, 5), 2, (1, 1), (2, 2), 0.0),     # Line end (t542 kernels)
    ((2, 33, 47, 4), (5, 5), 2, (1, 1), (2, 2), 0.0),     # ... ragged tiles, odd width
    ((1, 70, 130, 4), (5, 5), 2, (1, 1), (2, 2), 0.75),   # ... several 32 x 64 tiles per strip, padding value
    ((2, 37, 66, 4), (5, 5), 4, (1, 1), (2, 2), 0.0),     # 4 -> 4 stride 1, ragged
    ((2, 33, 47, 4), (5, 5), 4, (2, 2), (2, 2), 0.5),     # Line down_2, odd sizes, padding value
    ((1, 70, 140, 4), (5, 5), 4, (2, 2), (2, 2), 0.0),    # ... more than one tile
    ((2, 33, 47, 1), (5, 5), 1, (2, 2), (2, 2), 0.5),     # Paragraph down, odd sizes, padding value
    ((1, 70, 141, 1), (5, 5), 1, (1, 1), (2, 2), 0.25),   # Paragraph end, several tiles, odd width
    ((1, 37, 141, 1), (5, 5), 4, (2, 2), (2, 2), 0.0),    # Line down_1, odd width
    ((1, 3, 5, 4), (5, 5), 2, (1, 1), (2, 2), 0.5),       # images smaller than a tile / than a staging unit
    ((2, 2, 3, 4), (5, 5), 4, (2, 2), (2, 2), 0.0),
    ((2, 1, 1, 1), (5, 5), 1, (1, 1), (2, 2), 0.25),
    ((1, 2, 7, 1), (5, 5), 4, (2, 2), (2, 2), 0.0),
    ((1, 4, 3, 1), (5, 5), 1, (2, 2), (2, 2), 0.0),
    ((2, 21, 35, 1), (3, 3), 16, (1, 1), (1, 1), 0.5),    # Monochrome conv_1 unfused, padding value
    ((2, 21, 35, 16), (3, 3), 1, (1, 1), (1, 1), 0.0),    # Monochrome conv_2 unfused
    ((3, 11, 13, 6), (4, 4), 7, (2, 1), (1, 2), 0.25),    # generic kernels
]


@pytest.mark.parametrize('case', range(len(CONVS)))
def test_conv_kernels_f16(case, f16):
    from univer_ocr_amd.nn import ops
    CP = f16
    xs, ks, cout, st, pd, pv = CONVS[case]
    rng = np.random.default_rng(100 + case)
    X = rng.standard_normal(xs)
    w = rng.standard_normal((*ks, xs[3], cout)) * 0.2
    b = rng.standard_normal(cout)
    X16 = r16(X)
    w32, b32 = w.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
    h16_fwd = ks == (5, 5) and (xs[3] == 4 or (st == (1, 1) and cout == 1))
    h16_dx = ks == (5, 5) and (xs[3] == 4 or (st == (2, 2) and cout == 4))
    # binary16-MFMA kernels (conv_h16.hip): the float32 master weights enter the matrix cores rounded to binary16
    ref_y = O.conv2d_fwd(X16, r16(w32) if h16_fwd else w32, b32, st, pd, pv, True)
    g = rng.standard_normal(ref_y.shape)
    g16 = r16(g)
    ref_dx, ref_dw, ref_db = O.conv2d_bwd(X16, r16(w32) if h16_dx else w32, g16, st, pd, pv, True)
    Xd, gd = CP.copy(X), CP.copy(g)
    wd, bd = params32(CP, w, b)
    y = ops.conv2d_fwd(Xd, wd, bd, st, pd, pv, True)
    assert y.dtype == np.float16
    assert rel_linf(CP.asnumpy(y), ref_y) <= TOL_STORE
    # fused activation epilogue + the dx mask of a consumer (what Model.enable_fusion uses)
    ya = ops.conv2d_fwd(Xd, wd, bd, st, pd, pv, True, act='leaky', alpha=0.01)
    assert rel_linf(CP.asnumpy(ya), O.leaky_relu_fwd(ref_y, 0.01)) <= TOL_STORE
    gd.gscale = 4                                         # the gradient carries 2^4: dx keeps it, dw removes it
    dx = ops.conv2d_bwd_data(gd, wd, Xd.shape, st, pd)
    assert dx.dtype == np.float16 and dx.gscale == 4
    assert rel_linf(CP.asnumpy(dx), ref_dx) <= TOL_STORE
    mask_src = CP.copy(np.where(rng.random(xs) < 0.5, -1.0, 1.0) * np.abs(X))
    dxm = ops.conv2d_bwd_data(gd, wd, Xd.shape, st, pd, x_act=mask_src, act='leaky', alpha=0.01)
    slope = np.where(CP.asnumpy(mask_src).astype(np.float64) >= 0, 1.0, 0.01)
    assert rel_linf(CP.asnumpy(dxm), ref_dx * slope) <= TOL_STORE
    dw, db = CP.full(w.shape, 0.5, np.float32), CP.full(b.shape, 0.25, np.float32)
    ops.conv2d_bwd_weight(Xd, gd, dw, db, st, pd, pv, True, accumulate=True)
    assert rel_linf(CP.asnumpy(dw), ref_dw / 16 + 0.5) <= TOL_EXACT
    assert rel_linf(CP.asnumpy(db), ref_db / 16 + 0.25) <= TOL_EXACT


@pytest.mark.parametrize('ch,hl,wl', [(4, 24, 40), (1, 24, 40), (4, 17, 33), (1, 19, 21), (4, 40, 70), (4, 2, 3), (4, 1, 1)])
def test_upconv2x_f16(ch, hl, wl, f16):
    """Upsample2D(2) + conv5x5 on the low-res tensor (uocr_upconv2x_*) against the two layers of the oracle."""
    from univer_ocr_amd.nn import ops
    CP = f16
    rng = np.random.default_rng(ch * 100 + hl)
    xl = rng.standard_normal((2, hl, wl, ch))
    w = rng.standard_normal((5, 5, ch, ch)) * 0.2
    b = rng.standard_normal(ch)
    xl16 = r16(xl)
    w32, b32 = w.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
    up = O.upsample2d_fwd(xl16, (2, 2))
    ref_y = O.conv2d_fwd(up, w32, b32, 1, 2, 0.0, True)
    g16 = r16(rng.standard_normal(ref_y.shape))
    dx_hi, ref_dw, ref_db = O.conv2d_bwd(up, w32, g16, 1, 2, 0.0, True)
    ref_dx = O.upsample2d_bwd(dx_hi, (2, 2))
    xd, gd = CP.copy(xl), CP.copy(g16)
    wd, bd = params32(CP, w, b)
    y = ops.upconv2x_fwd(xd, wd, bd, (2, 2), True)
    # 4 channels: binary16 MFMAs with the phase-summed weights rounded to binary16 (conv_h16.hip): 3e-3
    assert y.dtype == np.float16 and rel_linf(CP.asnumpy(y), ref_y) <= (3e-3 if ch == 4 else TOL_STORE)
    ya = ops.upconv2x_fwd(xd, wd, bd, (2, 2), True, act='leaky', alpha=0.01)
    assert rel_linf(CP.asnumpy(ya), O.leaky_relu_fwd(ref_y, 0.01)) <= (3e-3 if ch == 4 else TOL_STORE)
    gd.gscale = 2
    dx = ops.upconv2x_bwd_data(gd, wd, xd.shape, (2, 2))
    # 4 channels: binary16 MFMAs over dy with the phase-summed weights rounded to binary16 (conv_h16.hip): 3e-3
    assert dx.gscale == 2 and rel_linf(CP.asnumpy(dx), ref_dx) <= (3e-3 if ch == 4 else TOL_STORE)
    mask_src = CP.copy(np.where(rng.random(xl.shape) < 0.5, -1.0, 1.0) * np.abs(xl))
    dxm = ops.upconv2x_bwd_data(gd, wd, xd.shape, (2, 2), x_act=mask_src, act='leaky', alpha=0.01)
    slope = np.where(CP.asnumpy(mask_src).astype(np.float64) >= 0, 1.0, 0.01)
    assert rel_linf(CP.asnumpy(dxm), ref_dx * slope) <= (3e-3 if ch == 4 else TOL_STORE)
    dw, db = CP.zeros(w.shape, np.float32), CP.zeros(b.shape, np.float32)
    ops.upconv2x_bwd_weight(xd, gd, dw, db, (2, 2), True, accumulate=False)
    assert rel_linf(CP.asnumpy(dw), ref_dw / 4) <= TOL_EXACT
    assert rel_linf(CP.asnumpy(db), ref_db / 4) <= TOL_EXACT


@pytest.mark.parametrize('shape', [(2, 32, 64), (3, 45, 70), (1, 16, 32), (2, 14, 30), (1, 5, 3)])
def test_conv_pair_f16(shape, f16):
    """The fused Monochrome block (uocr_conv_pair_*) with binary16 x / y / dy / dx on binary16 MFMAs: the weights,
    a1 = lrelu(conv_1) and d_a1 enter the matrix cores rounded to binary16 (a1 / d_a1 are what a layer-by-layer
    run in this mode stores; the float32 master weights are rounded as operands), accumulation is float32.
    Two checkers: (1) the float64 oracle with exactly those roundings -- what is left is float32 accumulation and
    rare 1-ulp ties: stored tensors 1e-3 (their own rounding), dw / db 2e-4; (2) the unrounded float64 oracle:
    y within 4e-3 (a few operand roundings of 2^-11 each); the backward is only piecewise continuous in the
    weights (a z1 within rounding of 0 switches its LeakyReLU slope between 1 and 0.01), so there 99 % of the
    dx pixels must lie within 4e-3 and the parameter gradients (sums of ~10^4 random-sign terms, where ONE switched
    position is ~1 % of the sum) within 5e-2: a sanity bound, (1) is the parity check."""
    from univer_ocr_amd.hip import lib as hiplib
    from univer_ocr_amd.nn import ops
    CP = f16
    n, h, w_ = shape
    rng = np.random.default_rng(h)
    x = rng.random((n, h, w_, 1))
    w1, b1 = rng.standard_normal((3, 3, 1, 16)) * 0.4, rng.standard_normal(16) * 0.1
    w2, b2 = rng.standard_normal((3, 3, 16, 1)) * 0.2, rng.standard_normal(1) * 0.1
    x16 = r16(x)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    g16 = r16(rng.standard_normal((n, h, w_, 1)))
    xd, gd = CP.copy(x), CP.copy(g16)
    p = params32(CP, w1, b1, w2, b2)
    y = ops.conv_pair_fwd(xd, *p, act2=hiplib.ACT_SIGMOID)
    assert y.dtype == np.float16
    y16 = CP.asnumpy(y).astype(np.float64)           # backward from the STORED (binary16) output, as the kernel sees it
    grads = [CP.zeros(a.shape, np.float32) for a in (w1, b1, w2, b2)]
    gd.gscale = 6
    dx = ops.conv_pair_bwd(xd, y, gd, p[0], p[1], p[2], *grads, act2=hiplib.ACT_SIGMOID, accumulate=False)
    assert dx.dtype == np.float16 and dx.gscale == 6
    for rounded, tol_store, tol_sum in ((True, TOL_STORE, 2e-4), (False, 4e-3, 4e-3)):
        q = r16 if rounded else (lambda a: a)
        z1 = O.conv2d_fwd(x16, q(f(w1)), f(b1), 1, 1, 0.0, True)
        a1 = q(O.leaky_relu_fwd(z1, 0.01))
        z2 = O.conv2d_fwd(a1, q(f(w2)), f(b2), 1, 1, 0.0, True)
        assert rel_linf(y16, O.sigmoid_fwd(z2)) <= tol_store, rounded
        gz2 = q(g16 * y16 * (1 - y16))
        ga1, ref_dw2, ref_db2 = O.conv2d_bwd(a1, q(f(w2)), gz2, 1, 1, 0.0, True)
        gz1 = q(O.leaky_relu_bwd(z1, ga1, 0.01))
        ref_dx, ref_dw1, ref_db1 = O.conv2d_bwd(x16, q(f(w1)), gz1, 1, 1, 0.0, True)
        if rounded:
            assert rel_linf(CP.asnumpy(dx), ref_dx) <= tol_store
        else:
            err = np.abs(CP.asnumpy(dx).astype(np.float64) - ref_dx) / np.abs(ref_dx).max()
            assert np.quantile(err, 0.99) <= tol_store
            tol_sum = 5e-2
        # (db1 sums d_a1 BEFORE its rounding to binary16 -- closer to the unrounded oracle than to the rounded one)
        for name, got, ref in zip(('dw1', 'db1', 'dw2', 'db2'), grads, (ref_dw1, ref_db1, ref_dw2, ref_db2)):
            assert rel_linf(CP.asnumpy(got), ref / 64) <= (4e-3 if name == 'db1' and rounded else tol_sum), (name, rounded)


@pytest.mark.parametrize('c,fold', [(1, True), (2, True), (1, False)])
def test_seg_loss_f16_scaled_gradient(c, fold, f16):
    """Dice on binary16 predictions: the loss itself (sums of exact inputs) matches to 1e-7, the gradient
    is written times 2^k (k chosen from the page size) and matches after removing the factor."""
    from univer_ocr_amd.nn import ops
    CP = f16
    rng = np.random.default_rng(7)
    n, h, w = 2, 64, 96
    pred = rng.random((n, h, w, c))
    gt = (rng.random((n, h, w, c)) > 0.6).astype(np.float64)
    p16 = r16(pred)
    ref_loss, ref_grad = O.dice_loss(p16, gt)
    if fold:
        ref_grad = ref_grad * p16 * (1 - p16)
    loss, grad = ops.seg_loss('dice', CP.copy(pred), CP.copy(gt), True, out_act='sigmoid' if fold else None)
    k = grad.gscale
    assert k == ops.f16_grad_scale_log2('seg', h * w) == 8          # floor(log2(6144)) - 4
    assert abs(float(loss) - ref_loss) <= 1e-7 * abs(ref_loss)
    got = CP.asnumpy(grad).astype(np.float64) / 2 ** k
    assert rel_linf(got, ref_grad) <= TOL_STORE


def test_seg_loss_f16_needs_the_scale_at_page_size(f16):
    """At 1024 x 2048 the Dice gradient is ~1e-6: below binary16's smallest normal number (6.1e-5), where only a
    few mantissa bits are left.  Written times 2^17 it keeps the full 11 bits; written unscaled it does not."""
    from univer_ocr_amd.nn import ops
    CP = f16
    rng = np.random.default_rng(8)
    pred = rng.random((1, 1024, 2048, 1))
    gt = (rng.random(pred.shape) > 0.6).astype(np.float64)
    p16 = r16(pred)
    _, ref_grad = O.dice_loss(p16, gt)
    ref_grad = ref_grad * p16 * (1 - p16)
    assert np.max(np.abs(ref_grad)) < 6.1e-5
    pd, gd = CP.copy(pred), CP.copy(gt)
    _, grad = ops.seg_loss('dice', pd, gd, True, out_act='sigmoid')
    assert grad.gscale == 17
    scaled_err = rel_linf(CP.asnumpy(grad).astype(np.float64) / 2 ** 17, ref_grad)
    CP.f16_grad_scale_log2 = 0
    _, plain = ops.seg_loss('dice', pd, gd, True, out_act='sigmoid')
    plain_err = rel_linf(CP.asnumpy(plain).astype(np.float64), ref_grad)
    assert plain.gscale == 0
    assert scaled_err <= TOL_STORE < plain_err, (scaled_err, plain_err)


def test_seg_loss_f16_gradient_saturates_instead_of_overflowing(f16):
    """A sparse channel early in training: three positive label pixels and a prediction of almost nothing on a
    1024 x 2048 page.  Without the folded Sigmoid the Dice gradient is ~2 / (sum p + sum g); times the static 2^17 scale
    it exceeds binary16's largest number.  The loss kernels clamp to +-65504 (an inf would turn dw and the weights
    into NaN on the next step); the loss value itself is exact."""
    from univer_ocr_amd.nn import ops
    CP = f16
    pred = np.full((1, 1024, 2048, 1), 1e-7)
    gt = np.zeros_like(pred)
    gt[0, 5, 7, 0] = gt[0, 600, 900, 0] = gt[0, 1023, 2047, 0] = 1.0
    loss, grad = ops.seg_loss('dice', CP.copy(pred), CP.copy(gt), True)
    g = CP.asnumpy(grad).astype(np.float64)
    assert grad.gscale == 17 and np.all(np.isfinite(g))
    assert np.max(np.abs(g)) == 65504.0                   # saturated, where the unclamped value is 2^17 * O(1)
    assert np.isfinite(float(loss))


def nest(flat):
    out = {}
    for key, value in flat.items():
        layer, pname = key.rsplit('/', 1)
        out.setdefault(layer, {})[pname] = value.tolist()
    return out


def f16_net_step(CP, name, shape, X, y, fuse=True):
    """One compute_loss_and_gradients of net `name` in float16 mode and the oracle's on the binary16-rounded
    page with float32-rounded weights.  Returns the error triple (prediction, loss, worst parameter gradient)."""
    from univer_ocr_amd.my_model.model import NET_MAKERS
    from univer_ocr_amd.nn.optimizers import Momentum
    net = O.make_net(name)
    for pn in net.params:
        net.params[pn] = net.params[pn].astype(np.float32).astype(np.float64)
    model = NET_MAKERS[name](shape, Momentum(lr=0.01, momentum=0))
    model.set_weights(nest(net.params))
    model.enable_fusion(fuse)
    ref_losses, ref_pred, _ = net.loss_and_grads(r16(X), y)
    losses = model.compute_loss_and_gradients(CP.copy(X), CP.copy(y))
    pred = CP.asnumpy(model.layers_outputs[0]).astype(np.float64)
    assert model.layers_outputs[0].dtype == np.float16
    if fuse:
        assert len(model._pairs_used) == (name == 'Monochrome') and len(model._ups_used) == 2 * (name != 'Monochrome')
    perr = rel_linf(pred, ref_pred)
    lerr = abs(float(losses['output_losses'][0]) - ref_losses['output_losses'][0]) / abs(ref_losses['output_losses'][0])
    gerr = max(rel_linf(CP.asnumpy(p.grad), net.grads[pn]) for pn, p in model.params().items())
    return perr, lerr, gerr


@pytest.mark.parametrize('name', ['Monochrome', 'Paragraph', 'Line'])
@pytest.mark.parametrize('fuse', [True, False])
def test_page_net_step_f16_vs_oracle(name, fuse, f16):
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    data = make_page_batch(2, 64, 128, 16, seed=31)
    tag_x, tag_y = {'Monochrome': ('image', 'monochrome'), 'Paragraph': ('monochrome', 'paragraph'),
                    'Line': ('monochrome', 'line')}[name]
    perr, lerr, gerr = f16_net_step(f16, name, data[tag_x].shape, data[tag_x], data[tag_y], fuse)
    assert perr <= 3e-3, f'{name}: prediction {perr:.2e}'
    assert lerr <= 3e-3, f'{name}: loss {lerr:.2e}'
    assert gerr <= 1e-2, f'{name}: parameter gradients {gerr:.2e}'


def test_config4_highres_f16_train_step_vs_oracle(f16):
    """BASELINE configs[4] geometry: Monochrome + Paragraph + Line train step (fwd, Dice, bwd, L2) on
    2 x 1024 x 2048 binary16 pages through the production (fused) kernels against the float64 oracle on the
    binary16-rounded pages; Dice gradients are ~1e-6 here, i.e. only representable through the 2^17 scale."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.nn import ops
    assert ops.f16_grad_scale_log2('seg', 1024 * 2048) == 17
    data = make_page_batch(2, 1024, 2048, 16, seed=41, char_batch=1)
    for name, tag_x, tag_y in (('Monochrome', 'image', 'monochrome'), ('Paragraph', 'monochrome', 'paragraph'),
                               ('Line', 'monochrome', 'line')):
        perr, lerr, gerr = f16_net_step(f16, name, data[tag_x].shape, data[tag_x], data[tag_y], True)
        assert perr <= 3e-3, f'{name}: prediction {perr:.2e}'
        assert lerr <= 3e-3, f'{name}: loss {lerr:.2e}'
        assert gerr <= 1e-2, f'{name}: parameter gradients {gerr:.2e}'


def test_config4_highres_f16_batch8_properties(f16):
    """BASELINE configs[4] at the benchmarked per-GPU batch (8 x 1024 x 2048, float16 storage): the Monochrome net's
    compute_loss_and_gradients through the production kernels.  The Dice loss and every parameter gradient are sums
    over independent pages (plus one L2 term per run), so the batch in two halves must add up (float32 / float64
    summation order only -- the binary16 roundings are per element and identical), predictions must agree bit for bit, and page 0 is checked
    against the float64 oracle on the binary16-rounded page."""
    from univer_ocr_amd.my_model.model import NET_MAKERS
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.nn.optimizers import Momentum
    CP = f16
    data = make_page_batch(8, 1024, 2048, 16, seed=43, char_batch=1)
    name, X, y = 'Monochrome', data['image'], data['monochrome']
    net = O.make_net(name)
    for pn in net.params:
        net.params[pn] = net.params[pn].astype(np.float32).astype(np.float64)

    def run(sl):
        model = NET_MAKERS[name](X[sl].shape, Momentum(lr=0.01, momentum=0))
        model.set_weights(nest(net.params))
        model.enable_fusion(True)
        losses = model.compute_loss_and_gradients(CP.copy(X[sl]), CP.copy(y[sl]))
        assert len(model._pairs_used) == 1
        return (CP.asnumpy(model.layers_outputs[0]), float(losses['output_losses'][0]),
                {pn: CP.asnumpy(p.grad).astype(np.float64) for pn, p in model.params().items()})
    pred, loss, grads = run(slice(0, 8))
    pa, la, ga = run(slice(0, 4))
    pb, lb, gb = run(slice(4, 8))
    _, _, gc = run(slice(0, 2))
    _, _, gd = run(slice(2, 4))
    assert np.array_equal(np.concatenate([pa, pb]), pred)
    assert abs(la + lb - loss) <= 2e-5 * abs(loss)
    for pn in grads:
        # every run adds the same L2 term r to the sum over its pages: r = C + D - A, and whole = A + B - r
        reg = gc[pn] + gd[pn] - ga[pn]
        assert rel_linf(ga[pn] + gb[pn] - reg, grads[pn]) <= 5e-5, pn
    ref_pred = net.forward(r16(X[:1]))
    assert rel_linf(pred[:1].astype(np.float64), ref_pred) <= 3e-3


def test_f16_train_steps_track_f32(f16):
    """Three SGD steps of the Line net in float16 mode stay next to the float32 run from the same weights (loss
    within 1e-2 relative, weights within 1e-3 of their magnitude): the master weights are float32, so the
    optimizer sees full-precision updates."""
    from univer_ocr_amd.my_model.model import make_line
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.nn.optimizers import Momentum
    CP = f16
    data = make_page_batch(2, 64, 128, 16, seed=5)
    start = O.make_net('Line').params                      # analytic, zero-mean weights
    runs = {}
    for dtype in ('float32', 'float16'):
        CP.set_dtype(dtype)
        model = make_line(data['monochrome'].shape, Momentum(lr=0.05, momentum=0))
        model.set_weights(nest(start))
        model.enable_fusion()
        X, y = CP.copy(data['monochrome']), CP.copy(data['line'])
        losses = [float(model.train(X, y)['output_losses'][0]) for _ in range(3)]
        runs[dtype] = (losses, {n: CP.asnumpy(p.value).astype(np.float64) for n, p in model.params().items()})
    for a, b in zip(runs['float32'][0], runs['float16'][0]):
        assert abs(a - b) <= 1e-2 * abs(a)
    moved = max(rel_linf(runs['float32'][1][n], start[n]) for n in start)
    assert moved > 1e-2, 'the steps did not move the weights: the comparison would be vacuous'
    for n, ref in runs['float32'][1].items():
        assert rel_linf(runs['float16'][1][n], ref) <= 1e-3, n
