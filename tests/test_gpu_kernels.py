"""GPU parity of the kernel VARIANTS behind one C-ABI entry point: every specialised kernel
(f32 MFMA implicit GEMM, shape-specialised direct kernels) must agree with the oracle on the same
seeded inputs, and with the generic kernel.  Variants are forced through uocr_ctx_set_option.

float32 tolerance: 1e-5 normalised max error (2e-5 for dw, a sum over all output pixels)."""
import numpy as np
import pytest

from conftest import load_golden, rel_linf
from oracle import nn_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture
def f32():
    from univer_ocr_amd.nn import CP
    CP.set_dtype('float32')
    yield CP
    CP.runtime().set_option('mfma', 1)
    CP.runtime().set_option('fast_paths', 1)
    CP.runtime().set_option('t32', 2)


def force_t32(CP, bits, fallback=None):
    """Force conv_t32 / conv_t32w kernels through the 't32' option.  The default library holds only the forms that are
    on by default (bit 2); the rest needs UOCR_BUILD_EXPERIMENTS=1 ./build.sh -- then `fallback` bits, or skip."""
    from univer_ocr_amd.hip.lib import HipError
    try:
        CP.runtime().set_option('t32', bits)
    except HipError as e:
        if 'built without' not in str(e):
            raise
        if fallback is None:
            pytest.skip('library built without UOCR_BUILD_EXPERIMENTS')
        CP.runtime().set_option('t32', fallback)


def check(a, b, tol, what):
    from univer_ocr_amd.nn import CP
    err = rel_linf(CP.asnumpy(a), b)
    assert err <= tol, f'{what}: rel_linf={err:.3e} > {tol:.1e}'


def run_conv(CP, ops, X, w, b, g, st, pd, pv, bias):
    Xd, wd, bd, gd = CP.copy(X), CP.copy(w), CP.copy(b), CP.copy(g)
    y = ops.conv2d_fwd(Xd, wd, bd, st, pd, pv, bias)
    dx = ops.conv2d_bwd_data(gd, wd, Xd.shape, st, pd)
    dw, db = CP.full(w.shape, 0.5), CP.full(b.shape, 0.25)
    ops.conv2d_bwd_weight(Xd, gd, dw, db, st, pd, pv, bias, accumulate=True)
    return y, dx, dw, db


CONV_SHAPES = [
    # (x shape, kernel, cout, stride, padding, pad_value, bias)
    ((2, 14, 10, 64), (5, 3), 64, (2, 1), (0, 1), 0.0, True),        # Char conv_2
    ((3, 9, 11, 32), (3, 3), 40, (2, 1), (1, 2), 0.25, True),        # N edge (40 of 64), odd stride/pad
    ((2, 7, 9, 64), (3, 2), 32, (1, 1), (1, 0), 0.0, False),         # no bias
    ((2, 20, 24, 32), (1, 1), 64, (1, 1), (0, 0), 0.0, True),        # 1x1
    ((3, 96, 128, 32), (3, 3), 64, (1, 1), (1, 1), 0.0, True),       # 36864 rows -> BM=128 tiles
    # the loaders' quotients are multiply-highs by host constants (FastDiv): divisors that are not powers of two --
    # 3 depth tiles per tap (96 channels), strides 3 and 2, a 4-wide kernel, 13 / 7 / 5-wide images
    ((2, 13, 7, 96), (3, 4), 32, (3, 2), (1, 2), 0.0, True),
    ((3, 10, 5, 32), (4, 3), 96, (2, 3), (2, 1), -0.5, True),
]


@pytest.mark.parametrize('case', range(len(CONV_SHAPES)))
@pytest.mark.parametrize('mode', ['mfma', 'generic'])
def test_conv_variants_against_oracle(case, mode, f32):
    from univer_ocr_amd.nn import ops
    CP = f32
    xs, ks, cout, st, pd, pv, bias = CONV_SHAPES[case]
    rng = np.random.default_rng(40 + case)
    X = rng.standard_normal(xs)
    w = rng.standard_normal((*ks, xs[3], cout)) * 0.1
    b = rng.standard_normal(cout)
    ref_y = O.conv2d_fwd(X, w, b, st, pd, pv, bias)
    g = rng.standard_normal(ref_y.shape)
    ref_dx, ref_dw, ref_db = O.conv2d_bwd(X, w, g, st, pd, pv, bias)
    CP.runtime().set_option('mfma', 2 if mode == 'mfma' else 0)
    CP.runtime().set_option('fast_paths', 0)
    y, dx, dw, db = run_conv(CP, ops, X, w, b, g, st, pd, pv, bias)
    check(y, ref_y, 1e-5, 'y')
    check(dx, ref_dx, 1e-5, 'dx')
    check(dw, ref_dw + 0.5, 2e-5, 'dw (accumulated onto 0.5)')
    check(db, ref_db + 0.25, 2e-5, 'db (accumulated onto 0.25)')


@pytest.mark.parametrize('tag', ['char2', 'char3'])
def test_conv_mfma_golden(tag, f32):
    """The reference's own outputs for the 64-channel Char convs through the MFMA path."""
    from univer_ocr_amd.nn import ops
    CP = f32
    g = load_golden('conv2d')
    kh, kw, sh, sw, ph, pw, pv, bias = g[f'{tag}/cfg']
    CP.runtime().set_option('mfma', 2)
    y, dx, dw, db = run_conv(CP, ops, g[f'{tag}/X'], g[f'{tag}/w'], g[f'{tag}/b'], g[f'{tag}/g'],
                             (int(sh), int(sw)), (int(ph), int(pw)), pv, bool(bias))
    check(y, g[f'{tag}/y'], 1e-5, 'y')
    check(dx, g[f'{tag}/dx'], 1e-5, 'dx')
    check(dw, g[f'{tag}/dw'] + 0.5, 2e-5, 'dw')
    check(db, g[f'{tag}/db'] + 0.25, 2e-5, 'db')


@pytest.mark.parametrize('m,n_in,n_out', [(12, 129, 162), (2048, 512, 1024), (300, 1024, 128), (70, 36, 50),
                                          (1000, 128, 162)])
@pytest.mark.parametrize('mode', ['mfma', 'generic'])
def test_dense_variants_against_oracle(m, n_in, n_out, mode, f32):
    from univer_ocr_amd.nn import ops
    CP = f32
    rng = np.random.default_rng(m + n_in)
    X = rng.standard_normal((m, n_in))
    w = rng.standard_normal((n_in + 1, n_out)) / np.sqrt(n_in)
    g = rng.standard_normal((m, n_out))
    ref_y = O.dense_fwd(X, w)
    ref_dx, ref_dw = O.dense_bwd(X, w, g)
    CP.runtime().set_option('mfma', 2 if mode == 'mfma' else 0)
    Xd, wd, gd = CP.copy(X), CP.copy(w), CP.copy(g)
    check(ops.dense_fwd(Xd, wd), ref_y, 1e-5, 'y')
    dw = CP.full(w.shape, 1.0)
    dx = ops.dense_bwd(Xd, wd, gd, dw, accumulate=True)
    check(dx, ref_dx, 1e-5, 'dx')
    check(dw, ref_dw + 1.0, 2e-5, 'dw')
    ops.dense_bwd(Xd, wd, gd, dw, accumulate=False, need_dx=False)
    check(dw, ref_dw, 2e-5, 'dw overwrite')


def test_mfma_identity_with_asymmetric_b(f32):
    """A = I, B asymmetric: catches a transposed C write or a swapped fragment map exactly."""
    from univer_ocr_amd.nn import ops
    CP = f32
    n = 96
    X = np.eye(n)
    w = np.zeros((n + 1, 80))
    w[:n] = np.arange(n * 80).reshape(n, 80) % 251
    w[n] = 1000.0
    CP.runtime().set_option('mfma', 2)
    y = CP.asnumpy(ops.dense_fwd(CP.copy(X), CP.copy(w)))
    assert np.array_equal(y, (w[:n] + 1000.0).astype(np.float32))


FAST_SHAPES = [
    # (x shape, kernel, cout, stride, padding): the nine instantiated my_model configurations, on
    # odd-sized images (partial 64x4 tiles, several row bands, border taps) with pad_value 0.25
    ((3, 37, 83, 1), (3, 3), 16, (1, 1), (1, 1)),
    ((3, 37, 83, 16), (3, 3), 1, (1, 1), (1, 1)),
    ((2, 41, 77, 1), (5, 5), 1, (2, 2), (2, 2)),
    ((2, 41, 77, 1), (5, 5), 1, (1, 1), (2, 2)),
    ((2, 41, 77, 1), (5, 5), 4, (2, 2), (2, 2)),
    ((2, 41, 77, 4), (5, 5), 4, (2, 2), (2, 2)),
    ((2, 41, 77, 4), (5, 5), 4, (1, 1), (2, 2)),
    ((2, 41, 77, 4), (5, 5), 2, (1, 1), (2, 2)),
    ((3, 32, 70, 1), (5, 3), 64, (2, 1), (0, 1)),
    ((5, 64, 130, 16), (3, 3), 1, (1, 1), (1, 1)),     # more rows than one band per image
    ((2, 40, 76, 1), (5, 5), 1, (2, 2), (2, 2)),       # even sizes: the stride-2 dx kernel stores pixel pairs
    ((2, 40, 140, 1), (5, 5), 4, (2, 2), (2, 2)),
]


@pytest.mark.parametrize('case', range(len(FAST_SHAPES)))
@pytest.mark.parametrize('pad_value,bias', [(0.0, True), (0.25, False)])
def test_fast_conv_kernels_against_oracle(case, pad_value, bias, f32):
    from univer_ocr_amd.nn import ops
    CP = f32
    xs, ks, cout, st, pd = FAST_SHAPES[case]
    rng = np.random.default_rng(90 + case)
    X = rng.standard_normal(xs)
    w = rng.standard_normal((*ks, xs[3], cout)) * 0.2
    b = rng.standard_normal(cout)
    ref_y = O.conv2d_fwd(X, w, b, st, pd, pad_value, bias)
    g = rng.standard_normal(ref_y.shape)
    ref_dx, ref_dw, ref_db = O.conv2d_bwd(X, w, g, st, pd, pad_value, bias)
    results = {}
    for mode in ('fast', 'generic'):
        CP.runtime().set_option('fast_paths', 1 if mode == 'fast' else 0)
        CP.runtime().set_option('mfma', 0)
        results[mode] = run_conv(CP, ops, X, w, b, g, st, pd, pad_value, bias)
        y, dx, dw, db = results[mode]
        check(y, ref_y, 1e-5, f'{mode} y')
        check(dx, ref_dx, 1e-5, f'{mode} dx')
        check(dw, ref_dw + 0.5, 2e-5, f'{mode} dw')
        check(db, ref_db + 0.25, 2e-5, f'{mode} db')
    # the two kernels must have actually been different code paths yet agree closely
    check(results['fast'][0], CP.asnumpy(results['generic'][0]).astype(np.float64), 1e-5, 'fast vs generic y')


TOEPLITZ_SHAPES = [
    # (x shape, cout): 5x5 / stride 1 / padding 2 layers of the page nets
    ((2, 40, 72, 4), 2), ((2, 33, 47, 4), 2), ((1, 70, 130, 4), 2),      # Line end: one tile, ragged, several tiles
    ((2, 37, 66, 4), 4),                                                 # 4 -> 4
    ((2, 40, 72, 1), 1), ((1, 70, 133, 1), 1),                           # Paragraph end
    ((1, 3, 5, 4), 2), ((2, 1, 1, 1), 1),                                # smaller than a tile / a staging unit
]


@pytest.mark.parametrize('case', range(len(TOEPLITZ_SHAPES)))
@pytest.mark.parametrize('pad_value,bias', [(0.0, True), (0.75, False)])
def test_toeplitz_f32_conv_kernels(case, pad_value, bias, f32):
    """conv_t32.hip (vertical-Toeplitz float32 MFMA forward / backward-data, every instantiation forced through the
    't32' option; the default library: the backward-data forms that are on by default) against the oracle, with the
    fused activation and the backward mask of a consumer."""
    from univer_ocr_amd.nn import ops
    CP = f32
    force_t32(CP, 63, fallback=2)
    xs, cout = TOEPLITZ_SHAPES[case]
    rng = np.random.default_rng(700 + case)
    X = rng.standard_normal(xs)
    w = rng.standard_normal((5, 5, xs[3], cout)) * 0.2
    b = rng.standard_normal(cout)
    ref_y = O.conv2d_fwd(X, w, b, 1, 2, pad_value, bias)
    g = rng.standard_normal(ref_y.shape)
    ref_dx, _, _ = O.conv2d_bwd(X, w, g, 1, 2, pad_value, bias)
    Xd, wd, bd, gd = CP.copy(X), CP.copy(w), CP.copy(b), CP.copy(g)
    check(ops.conv2d_fwd(Xd, wd, bd, (1, 1), (2, 2), pad_value, bias), ref_y, 1e-5, 'y')
    check(ops.conv2d_fwd(Xd, wd, bd, (1, 1), (2, 2), pad_value, bias, act='sigmoid'), O.sigmoid_fwd(ref_y), 1e-5, 'sigmoid(y)')
    check(ops.conv2d_bwd_data(gd, wd, Xd.shape, (1, 1), (2, 2)), ref_dx, 1e-5, 'dx')
    slope = np.where(X >= 0, 1.0, 0.01)
    check(ops.conv2d_bwd_data(gd, wd, Xd.shape, (1, 1), (2, 2), x_act=Xd, act='leaky', alpha=0.01), ref_dx * slope, 1e-5,
          'masked dx')


@pytest.mark.parametrize('ch,hl,wl', [(4, 24, 40), (4, 17, 33), (1, 24, 40), (1, 19, 21), (4, 40, 70), (4, 2, 3), (1, 1, 1)])
def test_toeplitz_f32_upconv_dgrad(ch, hl, wl, f32):
    """Upsample2D(2) + conv5x5 backward-data as a stride-2 6x6 window over dy (conv_t32.hip) against the two layers
    of the oracle."""
    from univer_ocr_amd.nn import ops
    CP = f32
    force_t32(CP, 63)
    rng = np.random.default_rng(ch * 10 + hl)
    xl = rng.standard_normal((2, hl, wl, ch))
    w = rng.standard_normal((5, 5, ch, ch)) * 0.2
    g = rng.standard_normal((2, 2 * hl, 2 * wl, ch))
    up = O.upsample2d_fwd(xl, (2, 2))
    dx_hi, _, _ = O.conv2d_bwd(up, w, g, 1, 2, 0.0, True)
    ref_dx = O.upsample2d_bwd(dx_hi, (2, 2))
    xd, wd, gd = CP.copy(xl), CP.copy(w), CP.copy(g)
    check(ops.upconv2x_bwd_data(gd, wd, xd.shape, (2, 2)), ref_dx, 1e-5, 'dx')
    slope = np.where(xl >= 0, 1.0, 0.01)
    check(ops.upconv2x_bwd_data(gd, wd, xd.shape, (2, 2), x_act=xd, act='leaky', alpha=0.01), ref_dx * slope, 1e-5,
          'masked dx')


T32W_SHAPES = [
    # (x shape, cout, stride): the 5x5 / padding 2 convs whose dw has a float32-MFMA form (conv_t32w.hip)
    ((2, 40, 72, 4), 2, 1), ((1, 70, 133, 4), 2, 1), ((2, 33, 47, 1), 1, 1), ((1, 70, 141, 1), 1, 1),
    ((2, 40, 72, 4), 4, 2), ((2, 33, 47, 4), 4, 2), ((1, 70, 140, 4), 4, 2),
    ((2, 40, 72, 1), 4, 2), ((1, 37, 141, 1), 4, 2), ((2, 33, 47, 1), 1, 2), ((1, 70, 140, 1), 1, 2),
    ((1, 3, 5, 4), 2, 1), ((2, 1, 1, 1), 1, 1), ((2, 2, 3, 4), 4, 2), ((1, 2, 7, 1), 4, 2), ((1, 4, 3, 1), 1, 2),   # tiny
]


@pytest.mark.parametrize('case', range(len(T32W_SHAPES)))
@pytest.mark.parametrize('pad_value,bias', [(0.0, True), (0.75, False)])
def test_toeplitz_f32_weight_gradients(case, pad_value, bias, f32):
    """conv_t32w.hip (channel planes + shifted dy operand on float32 MFMAs; off by default, forced through the 't32'
    option) against the oracle, accumulating into non-zero dw / db."""
    from univer_ocr_amd.nn import ops
    CP = f32
    force_t32(CP, 255)
    xs, cout, s = T32W_SHAPES[case]
    rng = np.random.default_rng(900 + case)
    X = rng.standard_normal(xs)
    w = rng.standard_normal((5, 5, xs[3], cout)) * 0.2
    ref_y = O.conv2d_fwd(X, w, np.zeros(cout), s, 2, pad_value, bias)
    g = rng.standard_normal(ref_y.shape)
    _, ref_dw, ref_db = O.conv2d_bwd(X, w, g, s, 2, pad_value, bias)
    Xd, gd = CP.copy(X), CP.copy(g)
    dw, db = CP.full(w.shape, 0.5), CP.full((cout,), 0.25)
    ops.conv2d_bwd_weight(Xd, gd, dw, db, (s, s), (2, 2), pad_value, bias, accumulate=True)
    check(dw, ref_dw + 0.5, 2e-5, 'dw')
    check(db, ref_db + 0.25, 2e-5, 'db')


PAIR_SHAPES = [(3, 37, 83), (2, 16, 32), (1, 1, 1), (2, 5, 200), (1, 70, 33), (4, 64, 96)]


@pytest.mark.parametrize('shape', PAIR_SHAPES)
@pytest.mark.parametrize('sigmoid,bias,pad1,need_dx', [(True, True, 0.0, True), (False, False, 0.25, True),
                                                      (True, True, 0.0, False)])
def test_conv_pair_kernels_against_oracle(shape, sigmoid, bias, pad1, need_dx, f32):
    """uocr_conv_pair_fwd/bwd (conv3x3 1->16 + LeakyReLU + conv3x3 16->1 [+ Sigmoid] with the 16-channel
    tensors recomputed on chip) == the oracle run layer by layer: ragged tiles, images smaller than a
    tile, several row bands per block, both dx variants, padding value of the first conv, no bias."""
    from univer_ocr_amd.hip import lib as hiplib
    from univer_ocr_amd.nn import ops
    CP = f32
    n, h, w = shape
    rng = np.random.default_rng(sum(shape))
    X = rng.standard_normal((n, h, w, 1))
    w1 = rng.standard_normal((3, 3, 1, 16)) * 0.4
    b1 = rng.standard_normal(16) * 0.3
    w2 = rng.standard_normal((3, 3, 16, 1)) * 0.2
    b2 = rng.standard_normal(1)
    alpha = 0.01
    z1 = O.conv2d_fwd(X, w1, b1, 1, 1, pad1, bias)
    a1 = O.leaky_relu_fwd(z1, alpha)
    z2 = O.conv2d_fwd(a1, w2, b2, 1, 1, 0.0, bias)
    ref_y = O.sigmoid_fwd(z2) if sigmoid else z2
    g = rng.standard_normal(ref_y.shape)
    gz2 = O.sigmoid_bwd(z2, g) if sigmoid else g
    ga1, ref_dw2, ref_db2 = O.conv2d_bwd(a1, w2, gz2, 1, 1, 0.0, bias)
    gz1 = O.leaky_relu_bwd(z1, ga1, alpha)
    ref_dx, ref_dw1, ref_db1 = O.conv2d_bwd(X, w1, gz1, 1, 1, pad1, bias)

    act2 = hiplib.ACT_SIGMOID if sigmoid else hiplib.ACT_NONE
    Xd, w1d, b1d, w2d, b2d, gd = (CP.copy(a) for a in (X, w1, b1, w2, b2, g))
    y = ops.conv_pair_fwd(Xd, w1d, b1d, w2d, b2d, pad1, bias, bias, alpha, act2)
    check(y, ref_y, 1e-5, 'y')
    dw1, db1, dw2, db2 = CP.full(w1.shape, 0.5), CP.full(b1.shape, 0.25), CP.full(w2.shape, -0.5), CP.full(b2.shape, 2.0)
    dx = ops.conv_pair_bwd(Xd, y, gd, w1d, b1d, w2d, dw1, db1, dw2, db2, pad1, bias, bias, alpha, act2,
                           need_dx=need_dx, accumulate=True)
    if need_dx:
        check(dx, ref_dx, 2e-5, 'dx')
    else:
        assert dx is None
    check(dw1, ref_dw1 + 0.5, 2e-5, 'dw1')
    check(db1, ref_db1 + 0.25, 2e-5, 'db1')
    check(dw2, ref_dw2 - 0.5, 2e-5, 'dw2')
    check(db2, ref_db2 + 2.0, 2e-5, 'db2')
    # accumulate = 0 overwrites
    ops.conv_pair_bwd(Xd, y, gd, w1d, b1d, w2d, dw1, db1, dw2, db2, pad1, bias, bias, alpha, act2,
                      need_dx=False, accumulate=False)
    check(dw1, ref_dw1, 2e-5, 'dw1 overwrite')
    check(dw2, ref_dw2, 2e-5, 'dw2 overwrite')


def test_conv_pair_rejects_what_it_does_not_implement(f32):
    from univer_ocr_amd.hip.lib import HipError
    from univer_ocr_amd.nn import ops
    CP = f32
    X = CP.copy(np.zeros((1, 8, 8, 1)))
    w1, b1 = CP.copy(np.zeros((3, 3, 1, 8))), CP.copy(np.zeros(8))
    w2, b2 = CP.copy(np.zeros((3, 3, 8, 1))), CP.copy(np.zeros(1))
    with pytest.raises(HipError, match='16 middle channels'):
        ops.conv_pair_fwd(X, w1, b1, w2, b2)


@pytest.mark.parametrize('shape', [(2, 5, 8, 1), (3, 4, 6, 4), (1, 3, 5, 1), (2, 7, 12, 4), (1, 64, 128, 1)])
def test_upsample2_vector_kernels_are_exact(shape, f32):
    """2x2 upsampling: the 16-byte-per-thread kernels (1 / 4 channels, row length a multiple of 4 floats)
    and the generic fallback (third shape) reproduce the oracle bit for bit, forward and backward."""
    from univer_ocr_amd.nn import ops
    CP = f32
    rng = np.random.default_rng(sum(shape))
    X = rng.standard_normal(shape).astype(np.float32)
    y = ops.upsample2d_fwd(CP.copy(X), (2, 2))
    assert np.array_equal(CP.asnumpy(y), O.upsample2d_fwd(X, (2, 2)).astype(np.float32))
    g = rng.standard_normal(y.shape).astype(np.float32)
    dx = ops.upsample2d_bwd(CP.copy(g), X.shape, (2, 2))
    n, h, w, c = shape
    ref = np.zeros(shape, np.float32)
    for j in range(2):                       # the reference's accumulation order (upsample.py:27-39)
        for i in range(2):
            ref = ref + g[:, j::2, i::2, :]
    assert np.array_equal(CP.asnumpy(dx), ref)


UP_SHAPES = [(2, 16, 32), (3, 7, 19), (1, 1, 1), (2, 33, 40), (1, 5, 70), (4, 40, 64)]


@pytest.mark.parametrize('shape', UP_SHAPES)
@pytest.mark.parametrize('ch', [4, 1])
@pytest.mark.parametrize('bias,act', [(True, 'leaky'), (False, None)])
def test_upconv2x_kernels_against_oracle(shape, ch, bias, act, f32):
    """uocr_upconv2x_* (Upsample2D(2) + 5x5 conv 4->4 evaluated on the low-res tensor as a 3x3 conv to 16
    phase channels) == the oracle's upsample followed by its conv, forward, dx (with and without the
    LeakyReLU' epilogue) and dw/db: ragged tiles, images smaller than a tile, several row bands."""
    from univer_ocr_amd.nn import ops
    CP = f32
    n, hl, wl = shape
    rng = np.random.default_rng(sum(shape) + 7)
    xl = rng.standard_normal((n, hl, wl, ch))
    w = rng.standard_normal((5, 5, ch, ch)) * 0.1
    b = rng.standard_normal(ch)
    alpha = 0.01
    xu = O.upsample2d_fwd(xl, (2, 2))
    z = O.conv2d_fwd(xu, w, b, 1, 2, 0.0, bias)
    ref_y = O.leaky_relu_fwd(z, alpha) if act else z
    g = rng.standard_normal(ref_y.shape)
    gz = O.leaky_relu_bwd(z, g, alpha) if act else g
    dxu, ref_dw, ref_db = O.conv2d_bwd(xu, w, gz, 1, 2, 0.0, bias)
    ref_dx = O.upsample2d_bwd(dxu, (2, 2))

    xd, wd, bd = CP.copy(xl), CP.copy(w), CP.copy(b)
    y = ops.upconv2x_fwd(xd, wd, bd, (2, 2), bias, act, alpha)
    check(y, ref_y, 1e-5, 'y')
    gd = CP.copy(gz)
    dx = ops.upconv2x_bwd_data(gd, wd, xl.shape, (2, 2))
    check(dx, ref_dx, 1e-5, 'dx')
    if ch == 4:
        # the per-phase weights written by the forward kernel and handed back: the same bits as computed in the call
        weff = CP.full((576,), np.nan)
        y2 = ops.upconv2x_fwd(xd, wd, bd, (2, 2), bias, act, alpha, weff=weff)
        assert np.array_equal(CP.asnumpy(y2), CP.asnumpy(y))
        assert np.isfinite(CP.asnumpy(weff)).all()
        dx2 = ops.upconv2x_bwd_data(gd, wd, xl.shape, (2, 2), weff=weff)
        assert np.array_equal(CP.asnumpy(dx2), CP.asnumpy(dx))
    # epilogue: dx *= LeakyReLU'(x_act) with x_act = the low-res input seen as an activation output
    dxm = ops.upconv2x_bwd_data(gd, wd, xl.shape, (2, 2), x_act=xd, act='leaky', alpha=alpha)
    check(dxm, ref_dx * ((xl >= 0) + alpha * (xl < 0)), 1e-5, 'dx with mask')
    dw, db = CP.full(w.shape, 0.5), CP.full(b.shape, 0.25)
    ops.upconv2x_bwd_weight(xd, gd, dw, db, (2, 2), bias, accumulate=True)
    check(dw, ref_dw + 0.5, 2e-5, 'dw')
    check(db, ref_db + 0.25, 2e-5, 'db')
    ops.upconv2x_bwd_weight(xd, gd, dw, db, (2, 2), bias, accumulate=False)
    check(dw, ref_dw, 2e-5, 'dw overwrite')


def test_upconv2x_rejects_other_shapes(f32):
    from univer_ocr_amd.hip.lib import HipError
    from univer_ocr_amd.nn import ops
    CP = f32
    x = CP.copy(np.zeros((1, 4, 4, 2)))
    w, b = CP.copy(np.zeros((5, 5, 2, 2))), CP.copy(np.zeros(2))
    with pytest.raises(HipError, match='channels only'):
        ops.upconv2x_fwd(x, w, b, (2, 2))


@pytest.mark.parametrize('kind', ['dice', 'jaccard'])
@pytest.mark.parametrize('shape', [(3, 9, 14, 2), (2, 100, 90, 1), (1, 1, 1, 1)])
def test_seg_loss_with_folded_output_sigmoid(kind, shape, f32):
    """uocr_seg_loss(out_act = sigmoid): the gradient w.r.t. the INPUT of the Sigmoid that produced pred
    == oracle loss gradient times p (1 - p); loss value and plain gradient unchanged (several chunks per
    image in the second shape)."""
    from univer_ocr_amd.nn import ops
    CP = f32
    rng = np.random.default_rng(sum(shape))
    z = rng.standard_normal(shape)
    pred = O.sigmoid_fwd(z)
    gt = (rng.random(shape) > 0.6).astype(np.float64)
    ref_loss, ref_grad = (O.dice_loss if kind == 'dice' else O.jaccard_loss)(pred, gt)
    pd, gd = CP.copy(pred), CP.copy(gt)
    loss, grad = ops.seg_loss(kind, pd, gd, True)
    assert abs(float(loss) - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    check(grad, ref_grad, 1e-5, 'grad')
    loss2, grad2 = ops.seg_loss(kind, pd, gd, True, out_act='sigmoid')
    assert float(loss2) == float(loss)
    p32 = pred.astype(np.float32).astype(np.float64)
    check(grad2, ref_grad * p32 * (1 - p32), 1e-5, 'grad through sigmoid')
    loss3, none = ops.seg_loss(kind, pd, gd, False)
    assert none is None and float(loss3) == float(loss)


@pytest.mark.parametrize('shape,pad_value,act', [((2, 40, 128, 4), 0.0, None), ((1, 37, 130, 4), 0.0, 'sigmoid'),
                                                 ((3, 16, 60, 4), 0.0, None), ((2, 21, 61, 4), 0.25, 'leaky'),
                                                 ((1, 5, 3, 4), 0.0, None), ((2, 64, 512, 4), 0.0, 'sigmoid')])
def test_conv_h3_error_compensated_f16_mfma_forward(shape, pad_value, act, f32):
    """ctx option h3 = 1: the float32 5x5 4 -> 2 convolution (the Line net's output layer) computed as
    x_hi w_hi + x_hi w_lo + x_lo w_hi on binary16 MFMAs (csrc/conv_h3.hip) holds the float32 tolerance against
    the oracle (1e-5 of the tensor's largest value): strips of 60 columns, bands of rows, ragged widths and heights,
    a non-zero padding value, fused activations; values over five orders of magnitude in one tensor."""
    from univer_ocr_amd.nn import ops
    CP = f32
    rt = CP.runtime()
    rng = np.random.default_rng(sum(shape))
    X = rng.standard_normal(shape) * np.exp(rng.uniform(-6, 2, shape))
    w = rng.standard_normal((5, 5, 4, 2)) * 0.2
    b = rng.standard_normal(2)
    ref = O.conv2d_fwd(X, w, b, (1, 1), (2, 2), pad_value, True)
    alpha = 0.0
    if act == 'sigmoid':
        ref = O.sigmoid_fwd(ref)
    elif act == 'leaky':
        alpha = 0.1
        ref = np.where(ref >= 0, ref, alpha * ref)
    Xd, wd, bd = CP.copy(X), CP.copy(w), CP.copy(b)
    from univer_ocr_amd.hip.lib import HipError
    try:
        rt.set_option('h3', 1)
    except HipError as e:
        assert 'built without' in str(e)
        pytest.skip('library built without UOCR_BUILD_EXPERIMENTS (the kernel was measured and not kept: DESIGN.md section 5c)')
    try:
        for band in (0, 16):
            rt.set_option('pair_band', band)
            y = ops.conv2d_fwd(Xd, wd, bd, (1, 1), (2, 2), pad_value, True, act=act, alpha=alpha)
            check(y, ref, 1e-5, f'h3 forward, band {band}')
        rt.set_option('h3', 0)
        y0 = ops.conv2d_fwd(Xd, wd, bd, (1, 1), (2, 2), pad_value, True, act=act, alpha=alpha)
        assert rel_linf(CP.asnumpy(y), CP.asnumpy(y0).astype(np.float64)) <= 5e-6     # against the float32 vector kernel
    finally:
        rt.set_option('h3', 0)
        rt.set_option('pair_band', 0)


def test_deferred_weight_gradients_run_as_one_group(f32):
    """uocr_wgrad_defer_begin / _flush (Runtime.defer_wgrad): the weight gradients of three convolutions (one of them the
    windows + dense layer of the Char net) and two dense layers are recorded and run as ONE grid + one reduction; dw / db
    against the oracle and against the separate launches (2e-5: the depth splits differ), accumulating into non-zero
    buffers; dx of the dense layers is NOT deferred; a flush in the middle (keep_open) and a second group on the same ctx."""
    from univer_ocr_amd.nn import ops
    CP = f32
    rt = CP.runtime()
    rng = np.random.default_rng(77)
    convs = [((4, 14, 64, 64), (5, 3), 64, (2, 1), (0, 1)), ((4, 5, 64, 64), (5, 3), 64, (2, 1), (0, 1)),
             ((3, 9, 20, 32), (3, 3), 48, (1, 1), (1, 1))]
    dense = [(256, 1024, 128), (256, 128, 162)]
    jobs = []
    for xs, ks, cout, st, pd in convs:
        X = rng.standard_normal(xs)
        w = rng.standard_normal((*ks, xs[3], cout)) * 0.1
        y = O.conv2d_fwd(X, w, np.zeros(cout), st, pd, 0.0, True)
        g = rng.standard_normal(y.shape)
        _, ref_dw, ref_db = O.conv2d_bwd(X, w, g, st, pd, 0.0, True)
        jobs.append(('conv', CP.copy(X), CP.copy(g), w.shape, cout, st, pd, ref_dw, ref_db))
    for m, n_in, n_out in dense:
        X = rng.standard_normal((m, n_in))
        w = rng.standard_normal((n_in + 1, n_out)) * 0.1
        g = rng.standard_normal((m, n_out))
        ref_dw = np.concatenate([X, np.ones((m, 1))], axis=1).T @ g
        ref_dx = g @ w[:-1].T
        jobs.append(('dense', CP.copy(X), CP.copy(g), CP.copy(w), ref_dw, ref_dx))

    def run(deferred, flush_after=None):
        outs = []
        scope = rt.defer_wgrad() if deferred else None
        if scope:
            scope.__enter__()
        try:
            for i, job in enumerate(jobs):
                if job[0] == 'conv':
                    _, Xd, gd, wshape, cout, st, pd, _, _ = job
                    dw, db = CP.full(wshape, 0.5), CP.full((cout,), 0.25)
                    ops.conv2d_bwd_weight(Xd, gd, dw, db, st, pd, 0.0, True, accumulate=True)
                    outs.append((dw, db))
                else:
                    _, Xd, gd, wd, _, _ = job
                    dw = CP.full(wd.shape, 0.5)
                    dx = ops.dense_bwd(Xd, wd, gd, dw, accumulate=True)
                    outs.append((dw, dx))
                if deferred and flush_after == i:
                    rt.flush_deferred()
        finally:
            if scope:
                scope.__exit__(None, None, None)
        return [(CP.asnumpy(a).astype(np.float64), CP.asnumpy(b).astype(np.float64)) for a, b in outs]

    plain = run(False)
    for flush_after in (None, 1, None):
        got = run(True, flush_after)
        for job, (a, b), (pa, pb) in zip(jobs, got, plain):
            if job[0] == 'conv':
                assert rel_linf(a, job[7] + 0.5) <= 2e-5 and rel_linf(b, job[8] + 0.25) <= 2e-5
            else:
                assert rel_linf(a, job[4] + 0.5) <= 2e-5 and rel_linf(b, job[5]) <= 1e-5
            assert rel_linf(a, pa) <= 2e-5 and rel_linf(b, pb) <= 2e-5


def test_deferred_finish_kernels_run_as_one_launch(f32):
    """Runtime.defer_wgrad: the finish kernels of the direct weight-gradient producers (the 5x5 convs of the Paragraph /
    Line nets in their four layouts, upsample + conv, the Monochrome pair block) are recorded and run as ONE launch at the
    flush, their block partials kept in a region of their own meanwhile; dw / db equal those of the separate finish kernels
    (float64 column sums in a different order: 1e-6) and the oracle's (2e-5), accumulating into non-zero buffers."""
    from univer_ocr_amd.nn import ops
    CP = f32
    rt = CP.runtime()
    rng = np.random.default_rng(99)
    convs = [((2, 40, 72, 1), 1, (2, 2)), ((2, 40, 72, 4), 2, (1, 1)), ((2, 40, 72, 1), 4, (2, 2)), ((2, 40, 72, 4), 4, (2, 2))]
    jobs = []
    for xs, cout, st in convs:
        X = rng.standard_normal(xs)
        w = rng.standard_normal((5, 5, xs[3], cout)) * 0.2
        y = O.conv2d_fwd(X, w, np.zeros(cout), st, 2, 0.0, True)
        g = rng.standard_normal(y.shape)
        _, ref_dw, ref_db = O.conv2d_bwd(X, w, g, st, 2, 0.0, True)
        jobs.append(('conv', CP.copy(X), CP.copy(g), w.shape, cout, st, ref_dw, ref_db))
    xl = rng.standard_normal((2, 20, 36, 4))
    wu = rng.standard_normal((5, 5, 4, 4)) * 0.2
    gu = rng.standard_normal((2, 40, 72, 4))
    _, ref_dwu, ref_dbu = O.conv2d_bwd(O.upsample2d_fwd(xl, (2, 2)), wu, gu, 1, 2, 0.0, True)
    jobs.append(('up', CP.copy(xl), CP.copy(gu), wu.shape, ref_dwu, ref_dbu))
    Xp = rng.standard_normal((2, 24, 64, 1))
    w1, b1 = rng.standard_normal((3, 3, 1, 16)) * 0.4, rng.standard_normal(16) * 0.3
    w2, b2 = rng.standard_normal((3, 3, 16, 1)) * 0.2, rng.standard_normal(1)
    gp = rng.standard_normal((2, 24, 64, 1))
    pd = [CP.copy(a) for a in (Xp, w1, b1, w2, b2, gp)]
    yp = ops.conv_pair_fwd(pd[0], pd[1], pd[2], pd[3], pd[4], alpha=0.01)

    def run(deferred):
        outs = []
        scope = rt.defer_wgrad() if deferred else None
        if scope:
            scope.__enter__()
        try:
            for job in jobs:
                if job[0] == 'conv':
                    _, Xd, gd, wshape, cout, st, _, _ = job
                    dw, db = CP.full(wshape, 0.5), CP.full((cout,), 0.25)
                    ops.conv2d_bwd_weight(Xd, gd, dw, db, st, (2, 2), 0.0, True, accumulate=True)
                else:
                    _, xd, gd, wshape, _, _ = job
                    dw, db = CP.full(wshape, 0.5), CP.full((4,), 0.25)
                    ops.upconv2x_bwd_weight(xd, gd, dw, db, (2, 2), True, accumulate=True)
                outs += [dw, db]
            grads = [CP.full(a.shape, 0.5) for a in (w1, b1, w2, b2)]
            ops.conv_pair_bwd(pd[0], yp, pd[5], pd[1], pd[2], pd[3], *grads, alpha=0.01, need_dx=False, accumulate=True)
            outs += grads
        finally:
            if scope:
                scope.__exit__(None, None, None)
        return [CP.asnumpy(a).astype(np.float64) for a in outs]

    plain = run(False)
    for _ in range(2):
        got = run(True)
        for a, b in zip(got, plain):
            assert rel_linf(a, b) <= 1e-6
        k = 0
        for job in jobs:
            ref_dw, ref_db = job[-2], job[-1]
            assert rel_linf(got[k], ref_dw + 0.5) <= 2e-5 and rel_linf(got[k + 1], ref_db + 0.25) <= 2e-5
            k += 2


def test_cross_entropy_single_launch_sums(f32):
    """SoftmaxCrossEntropy / SigmoidCrossEntropy add their per-block partials in the last block to arrive (one launch
    each): value against the oracle over many blocks, repeated calls (counter back at zero), odd row counts."""
    from univer_ocr_amd.nn import ops
    CP = f32
    for m, c in ((256, 162), (1001, 37), (3, 162), (4097, 70)):
        rng = np.random.default_rng(m + c)
        z = rng.standard_normal((m, c)) * 3
        gt = np.zeros((m, c))
        gt[np.arange(m), rng.integers(0, c, m)] = 1.0
        ref_loss, ref_grad = O.softmax_ce_loss(z, gt)
        zd, gd = CP.copy(z), CP.copy(gt)
        for _ in range(3):
            loss, grad = ops.softmax_ce(zd, gd)
            assert abs(float(loss) - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (m, c)
            check(grad, ref_grad, 1e-5, f'softmax grad {m}x{c}')
        ref_loss, ref_grad = O.sigmoid_ce_loss(z, gt)
        for _ in range(3):
            loss, grad = ops.sigmoid_ce(zd, gd)
            assert abs(float(loss) - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (m, c)
            check(grad, ref_grad, 1e-5, f'sigmoid grad {m}x{c}')


@pytest.mark.parametrize('shape', [(2, 6, 8, 4), (3, 10, 14, 8), (1, 2, 2, 4), (2, 64, 96, 16)])
def test_maxpool2_vector_kernels_match_generic(shape, f32):
    """2x2 / stride 2 max pooling with c % 4 == 0 (16-byte kernels) == the oracle, with ties inside windows
    (values drawn from a 3-value set) so the mask and the dy / ties split are exercised; the mask must equal
    the generic kernel's byte for byte."""
    from univer_ocr_amd.nn import ops
    CP = f32
    rng = np.random.default_rng(sum(shape))
    X = rng.integers(0, 3, shape).astype(np.float64) - 1.0
    ref_y, ref_mask = O.maxpool2d_fwd(X, (2, 2), (2, 2), (0, 0))
    y, mask = ops.maxpool2d_fwd(CP.copy(X), (2, 2), (2, 2), (0, 0))
    assert np.array_equal(CP.asnumpy(y), ref_y.astype(np.float32))
    assert np.array_equal(CP.asnumpy(mask), ref_mask.astype(np.uint8))
    g = rng.standard_normal(ref_y.shape)
    dx = ops.maxpool2d_bwd(CP.copy(g), mask, X.shape, (2, 2), (2, 2), (0, 0))
    check(dx, O.maxpool2d_bwd(g, ref_mask, X.shape, (2, 2), (2, 2), (0, 0)), 1e-6, 'dx')
    CP.runtime().set_option('fast_paths', 1)


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_fused_momentum_tail_matches_separate_calls(dtype):
    """uocr_momentum_step_fused (regularisers of up to 4 ranges + Momentum update + gradient reset) == the three
    separate entry points in the same order, to one rounding (the compiler contracts multiply-adds differently
    when the regularised gradient stays in a register) and to 1e-14 for the loss."""
    from univer_ocr_amd.nn import CP, ops
    CP.set_dtype(dtype)
    try:
        rng = np.random.default_rng(3)
        n = 5000
        w0, g0, v0 = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n) * 0.1
        ranges = [(('l2', 0.01), 0, 1024), (('l1', 0.02), 2048, 3000), (('l2', 0.005), 4096, 5000)]
        w, g, v = CP.copy(w0), CP.copy(g0), CP.copy(v0)
        slot = CP.zeros((1,), np.float64)
        for (kind, strength), lo, hi in ranges:
            ops.regularize(kind, type(w)(w.t[lo:hi]), type(g)(g.t[lo:hi]), strength, slot, True)
        ops.momentum_step(w, g, v, 0.05, 0.9)
        ref_loss = float(slot.t.item())
        w2, g2, v2 = CP.copy(w0), CP.copy(g0), CP.copy(v0)
        loss = ops.momentum_step_fused(w2, g2, v2, 0.05, 0.9, ranges)
        eps = 4 * float(np.finfo(np.dtype(dtype)).eps)
        assert rel_linf(CP.asnumpy(w2), CP.asnumpy(w).astype(np.float64)) <= eps
        assert rel_linf(CP.asnumpy(v2), CP.asnumpy(v).astype(np.float64)) <= eps
        assert not np.any(CP.asnumpy(g2))
        assert abs(float(loss) - ref_loss) <= 1e-14 * max(1.0, abs(ref_loss))
        # no ranges: plain momentum + reset
        w3, g3, v3 = CP.copy(w0), CP.copy(g0), CP.copy(v0)
        assert ops.momentum_step_fused(w3, g3, v3, 0.05, 0.9, []) == 0
        w4, g4, v4 = CP.copy(w0), CP.copy(g0), CP.copy(v0)
        ops.momentum_step(w4, g4, v4, 0.05, 0.9)
        assert rel_linf(CP.asnumpy(w3), CP.asnumpy(w4).astype(np.float64)) <= eps and not np.any(CP.asnumpy(g3))
    finally:
        CP.set_dtype('float32')


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_fused_adam_tail_matches_separate_calls(dtype):
    """uocr_adam_step_fused == uocr_l2_reg on the range, then uocr_adam_step, then a zeroed gradient."""
    from univer_ocr_amd.nn import CP, ops
    CP.set_dtype(dtype)
    try:
        rng = np.random.default_rng(4)
        n = 4100
        arrs = [rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n) * 0.1, rng.random(n) * 0.1]
        ranges = [(('l2', 0.01), 64, 4096)]
        w, g, v, a = (CP.copy(x) for x in arrs)
        slot = CP.zeros((1,), np.float64)
        ops.regularize('l2', type(w)(w.t[64:4096]), type(g)(g.t[64:4096]), 0.01, slot, True)
        ops.adam_step(w, g, v, a, 0.01, 0.9, 0.999, 1e-8)
        w2, g2, v2, a2 = (CP.copy(x) for x in arrs)
        loss = ops.adam_step_fused(w2, g2, v2, a2, 0.01, 0.9, 0.999, 1e-8, ranges)
        eps = 8 * float(np.finfo(np.dtype(dtype)).eps)
        for got, ref, what in ((w2, w, 'w'), (v2, v, 'velocity'), (a2, a, 'accumulated')):
            assert rel_linf(CP.asnumpy(got), CP.asnumpy(ref).astype(np.float64)) <= eps, what
        assert not np.any(CP.asnumpy(g2))
        assert abs(float(loss) - float(slot.t.item())) <= 1e-14 * max(1.0, abs(float(loss)))
    finally:
        CP.set_dtype('float32')


@pytest.mark.parametrize('shape,n_out', [((3, 1, 70, 64), 96), ((2, 2, 19, 32), 64), ((2, 1, 8, 64), 32)])
@pytest.mark.parametrize('masked', [False, True])
def test_windows_dense_against_the_three_layers(shape, n_out, masked, f32):
    """Conv2DToBatchedFixedWidthed(8) + Flatten + FullyConnected as one implicit GEMM (ops.windows_dense_*)
    against the oracle's three layers, with and without the folded LeakyReLU' of the feature map."""
    from univer_ocr_amd.nn import ops
    CP = f32
    n, h, wd, c = shape
    rng = np.random.default_rng(n_out + wd)
    X = rng.standard_normal(shape)
    w = rng.standard_normal((h * 8 * c + 1, n_out)) * 0.1
    windows = O.fixed_width_fwd(X, 8)
    flat = windows.reshape(n * wd, -1)
    ref_y = O.dense_fwd(flat, w)
    g = rng.standard_normal(ref_y.shape)
    ref_dflat, ref_dw = O.dense_bwd(flat, w, g)
    ref_dx = O.fixed_width_bwd(ref_dflat.reshape(windows.shape), shape, 8)
    if masked:
        ref_dx = ref_dx * np.where(X >= 0, 1.0, 0.01)
    dX, dW = CP.copy(X), CP.copy(w)
    y = ops.windows_dense_fwd(dX, dW, 8)
    check(y, ref_y, 1e-5, 'y')
    dw = CP.copy(np.full(w.shape, 0.5))
    dx = ops.windows_dense_bwd(dX, dW, CP.copy(g), dw, 8, accumulate=True,
                               **({'x_act': dX, 'act': 'leaky', 'alpha': 0.01} if masked else {}))
    check(dx, ref_dx, 1e-5, 'dx')
    check(dw, ref_dw + 0.5, 2e-5, 'dw')


@pytest.mark.parametrize('m,n_in,n_out', [(2048, 512, 1024), (70, 36, 50), (300, 128, 162)])
def test_dense_with_fused_activations(m, n_in, n_out, f32):
    """FullyConnected + LeakyRelu as one GEMM (epilogue of the MFMA kernel, elementwise pass behind the generic
    one) and dx through the fused activation that produced the layer's input (uocr_dense_fwd_act / _bwd_act)."""
    from univer_ocr_amd.nn import ops
    CP = f32
    rng = np.random.default_rng(m)
    X = rng.standard_normal((m, n_in))
    w = rng.standard_normal((n_in + 1, n_out)) * 0.1
    pre = O.dense_fwd(X, w)
    check(ops.dense_fwd(CP.copy(X), CP.copy(w), 'leaky', 0.01), np.where(pre >= 0, pre, 0.01 * pre), 1e-5, 'leaky fwd')
    check(ops.dense_fwd(CP.copy(X), CP.copy(w), 'sigmoid'), 1.0 / (1.0 + np.exp(-pre)), 1e-5, 'sigmoid fwd')
    g = rng.standard_normal(pre.shape)
    ref_dx, ref_dw = O.dense_bwd(X, w, g)
    dw = CP.copy(np.zeros(w.shape))
    dx = ops.dense_bwd(CP.copy(X), CP.copy(w), CP.copy(g), dw, accumulate=False, x_act='leaky', x_alpha=0.01)
    check(dx, ref_dx * np.where(X >= 0, 1.0, 0.01), 1e-5, 'dx through leaky')
    check(dw, ref_dw, 2e-5, 'dw')
