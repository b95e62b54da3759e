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
