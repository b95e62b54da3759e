"""Out-of-bounds guard for the hand-written multi-layer kernels: every output of the C-ABI call lives in the
middle of a larger buffer whose borders hold a sentinel; after the call the borders must be untouched and
the payload fully written (no sentinel left).  Ragged shapes: partial tiles in both directions, images
smaller than a tile, several row bands per block."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GUARD = 2048            # floats on each side
SENTINEL = 777.25


class Guarded:
    """A float32 device buffer of `count` payload elements between two sentinel borders."""

    def __init__(self, CP, count, fill=SENTINEL):
        self.CP, self.count = CP, int(count)
        self.buf = CP.full((self.count + 2 * GUARD,), fill, np.float32)

    @property
    def ptr(self):
        return self.buf.ptr + GUARD * 4

    def check(self, what, expect_written=True):
        host = self.CP.asnumpy(self.buf)
        assert np.all(host[:GUARD] == SENTINEL), f'{what}: wrote BEFORE the buffer'
        assert np.all(host[GUARD + self.count:] == SENTINEL), f'{what}: wrote PAST the buffer'
        payload = host[GUARD:GUARD + self.count]
        if expect_written:
            assert not np.any(payload == SENTINEL), f'{what}: {int(np.sum(payload == SENTINEL))} elements not written'
        return payload


@pytest.fixture
def rt():
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    return CP, CP.runtime()


SHAPES = [(2, 37, 83), (1, 1, 1), (3, 16, 32), (1, 15, 31), (2, 70, 33), (1, 29, 61)]


@pytest.mark.parametrize('shape', SHAPES)
def test_conv_pair_stays_inside_its_buffers(shape, rt):
    from univer_ocr_amd.hip import lib as hiplib
    CP, runtime = rt
    n, h, w = shape
    rng = np.random.default_rng(1)
    x = CP.copy(rng.standard_normal((n, h, w, 1)))
    w1, b1 = CP.copy(rng.standard_normal((3, 3, 1, 16)) * 0.3), CP.copy(rng.standard_normal(16))
    w2, b2 = CP.copy(rng.standard_normal((3, 3, 16, 1)) * 0.2), CP.copy(rng.standard_normal(1))
    y = Guarded(CP, n * h * w)
    runtime.call('uocr_conv_pair_fwd', hiplib.F32, x.ptr, w1.ptr, b1.ptr, w2.ptr, b2.ptr, y.ptr, n, h, w, 16, 0.0, 1, 1,
                 0.01, hiplib.ACT_SIGMOID)
    y.check('pair fwd y')
    g = CP.copy(rng.standard_normal((n, h, w, 1)))
    outs = {'dw1': Guarded(CP, 144), 'db1': Guarded(CP, 16), 'dw2': Guarded(CP, 144), 'db2': Guarded(CP, 1),
            'dx': Guarded(CP, n * h * w)}
    runtime.call('uocr_conv_pair_bwd', hiplib.F32, x.ptr, y.ptr, g.ptr, w1.ptr, b1.ptr, w2.ptr, outs['dw1'].ptr,
                 outs['db1'].ptr, outs['dw2'].ptr, outs['db2'].ptr, outs['dx'].ptr, n, h, w, 16, 0.0, 1, 1, 0.01,
                 hiplib.ACT_SIGMOID, 0)
    for name, out in outs.items():
        out.check(f'pair bwd {name}')


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('ch', [4, 1])
def test_upconv_stays_inside_its_buffers(shape, ch, rt):
    from univer_ocr_amd.hip import lib as hiplib
    CP, runtime = rt
    n, hl, wl = shape
    rng = np.random.default_rng(2)
    xl = CP.copy(rng.standard_normal((n, hl, wl, ch)))
    w, b = CP.copy(rng.standard_normal((5, 5, ch, ch)) * 0.1), CP.copy(rng.standard_normal(ch))
    dims = (n, hl, wl, ch, ch, 5, 5, 2, 2)
    y = Guarded(CP, n * 4 * hl * wl * ch)
    runtime.call('uocr_upconv2x_fwd', hiplib.F32, xl.ptr, w.ptr, b.ptr, y.ptr, *dims, 1, hiplib.ACT_LEAKY, 0.01, None)
    y.check('upconv fwd y')
    g = CP.copy(rng.standard_normal((n, 2 * hl, 2 * wl, ch)))
    dx = Guarded(CP, n * hl * wl * ch)
    runtime.call('uocr_upconv2x_bwd_data', hiplib.F32, g.ptr, w.ptr, dx.ptr, *dims, xl.ptr, hiplib.ACT_LEAKY, 0.01, None)
    dx.check('upconv dx')
    dw, db = Guarded(CP, 25 * ch * ch), Guarded(CP, ch)
    runtime.call('uocr_upconv2x_bwd_weight', hiplib.F32, xl.ptr, g.ptr, dw.ptr, db.ptr, *dims, 1, 0)
    dw.check('upconv dw')
    db.check('upconv db')


@pytest.mark.parametrize('shape', [(2, 5, 8, 1), (3, 4, 6, 4), (1, 33, 20, 4), (2, 7, 12, 1)])
def test_vector_upsample_stays_inside_its_buffers(shape, rt):
    from univer_ocr_amd.hip import lib as hiplib
    CP, runtime = rt
    n, h, w, c = shape
    x = CP.copy(np.random.default_rng(3).standard_normal(shape))
    y = Guarded(CP, 4 * n * h * w * c)
    runtime.call('uocr_upsample2d_fwd', hiplib.F32, x.ptr, y.ptr, n, h, w, c, 2, 2)
    y.check('upsample fwd')
    g = CP.copy(np.random.default_rng(4).standard_normal((n, 2 * h, 2 * w, c)))
    dx = Guarded(CP, n * h * w * c)
    runtime.call('uocr_upsample2d_bwd', hiplib.F32, g.ptr, dx.ptr, n, h, w, c, 2, 2)
    dx.check('upsample bwd')


@pytest.mark.parametrize('shape', [(3, 9, 14, 2), (2, 64, 128, 1), (1, 7, 3, 1)])
def test_seg_loss_gradient_stays_inside_its_buffer(shape, rt):
    from univer_ocr_amd.hip import lib as hiplib
    CP, runtime = rt
    n, h, w, c = shape
    rng = np.random.default_rng(5)
    pred = CP.copy(1.0 / (1.0 + np.exp(-rng.standard_normal(shape))))
    gt = CP.copy((rng.random(shape) > 0.5).astype(np.float64))
    grad = Guarded(CP, n * h * w * c)
    slot = CP.empty((1,), np.float64)
    runtime.call('uocr_seg_loss', hiplib.F32, hiplib.LOSS_DICE, pred.ptr, gt.ptr, grad.ptr, slot.ptr, n, h * w, c,
                 hiplib.ACT_SIGMOID)
    grad.check('seg grad')


CONVS = [
    # x shape (ragged), kernel, cout, stride, padding: every shape-specialised direct kernel, the LDS-tiled
    # forward, the f32 MFMA implicit GEMM (64 -> 64) and the generic kernels (3 -> 5)
    ((3, 37, 83, 1), (3, 3), 16, (1, 1), (1, 1)),
    ((3, 37, 83, 16), (3, 3), 1, (1, 1), (1, 1)),
    ((2, 41, 77, 1), (5, 5), 1, (2, 2), (2, 2)),
    ((2, 41, 77, 1), (5, 5), 1, (1, 1), (2, 2)),
    ((2, 41, 77, 1), (5, 5), 4, (2, 2), (2, 2)),
    ((2, 41, 77, 4), (5, 5), 4, (2, 2), (2, 2)),
    ((2, 41, 77, 4), (5, 5), 4, (1, 1), (2, 2)),
    ((2, 41, 77, 4), (5, 5), 2, (1, 1), (2, 2)),
    ((3, 32, 70, 1), (5, 3), 64, (2, 1), (0, 1)),
    ((3, 14, 70, 64), (5, 3), 64, (2, 1), (0, 1)),
    ((2, 9, 11, 3), (3, 2), 5, (2, 1), (1, 0)),
]


@pytest.mark.parametrize('case', range(len(CONVS)))
def test_conv2d_entry_points_stay_inside_their_buffers(case, rt):
    from univer_ocr_amd.hip import lib as hiplib
    from univer_ocr_amd.nn import ops
    CP, runtime = rt
    xs, ks, cout, st, pd = CONVS[case]
    n, h, w, cin = xs
    oh, ow = ops.conv_out_hw(h, w, ks, st, pd)
    rng = np.random.default_rng(10 + case)
    x = CP.copy(rng.standard_normal(xs))
    wt, b = CP.copy(rng.standard_normal((*ks, cin, cout)) * 0.2), CP.copy(rng.standard_normal(cout))
    dims = (n, h, w, cin, cout, ks[0], ks[1], st[0], st[1], pd[0], pd[1], oh, ow)
    y = Guarded(CP, n * oh * ow * cout)
    runtime.call('uocr_conv2d_fwd', hiplib.F32, x.ptr, wt.ptr, b.ptr, y.ptr, *dims, 0.0, 1, hiplib.ACT_LEAKY, 0.01)
    y.check('conv fwd y')
    g = CP.copy(rng.standard_normal((n, oh, ow, cout)))
    dx = Guarded(CP, n * h * w * cin)
    runtime.call('uocr_conv2d_bwd_data', hiplib.F32, g.ptr, wt.ptr, dx.ptr, *dims, x.ptr, hiplib.ACT_LEAKY, 0.01)
    dx.check('conv dx')
    dw, db = Guarded(CP, wt.size), Guarded(CP, cout)
    runtime.call('uocr_conv2d_bwd_weight', hiplib.F32, x.ptr, g.ptr, dw.ptr, db.ptr, *dims, 0.0, 1, 0)
    dw.check('conv dw')
    db.check('conv db')


@pytest.mark.parametrize('shape', [(2, 6, 8, 4), (1, 2, 2, 4), (3, 10, 14, 8)])
def test_vector_maxpool_stays_inside_its_buffers(shape, rt):
    from univer_ocr_amd.hip import lib as hiplib
    CP, runtime = rt
    n, h, w, c = shape
    x = CP.copy(np.random.default_rng(6).integers(0, 3, shape).astype(np.float64))
    y = Guarded(CP, n * (h // 2) * (w // 2) * c)
    mask = Guarded(CP, n * h * w * c // 4)              # uint8 mask, counted in 4-byte words
    args = (n, h, w, c, 2, 2, 2, 2, 0, 0, h // 2, w // 2)
    runtime.call('uocr_maxpool2d_fwd', hiplib.F32, x.ptr, y.ptr, mask.ptr, *args)
    y.check('maxpool y')
    mask.check('maxpool mask')
    g = CP.copy(np.random.default_rng(7).standard_normal((n, h // 2, w // 2, c)))
    dx = Guarded(CP, n * h * w * c)
    runtime.call('uocr_maxpool2d_bwd', hiplib.F32, g.ptr, mask.ptr, dx.ptr, *args)
    dx.check('maxpool dx')


@pytest.mark.parametrize('m,n_in,n_out', [(70, 36, 50), (2048, 512, 1024), (300, 1024, 128), (129, 129, 162)])
def test_dense_entry_points_stay_inside_their_buffers(m, n_in, n_out, rt):
    """FullyConnected forward / backward (MFMA GEMM with split depth and the generic GEMM for the small shape)."""
    from univer_ocr_amd.hip import lib as hiplib
    CP, runtime = rt
    rng = np.random.default_rng(m + n_in)
    x = CP.copy(rng.standard_normal((m, n_in)))
    w = CP.copy(rng.standard_normal((n_in + 1, n_out)) * 0.1)
    y = Guarded(CP, m * n_out)
    runtime.call('uocr_dense_fwd', hiplib.F32, x.ptr, w.ptr, y.ptr, m, n_in, n_out)
    y.check('dense y')
    g = CP.copy(rng.standard_normal((m, n_out)))
    dx, dw = Guarded(CP, m * n_in), Guarded(CP, (n_in + 1) * n_out)
    runtime.call('uocr_dense_bwd', hiplib.F32, x.ptr, w.ptr, g.ptr, dx.ptr, dw.ptr, m, n_in, n_out, 0)
    dx.check('dense dx')
    dw.check('dense dw')


@pytest.mark.parametrize('shape', [(3, 1, 70, 64), (2, 2, 16, 4)])
def test_fixed_width_and_softmax_stay_inside_their_buffers(shape, rt):
    from univer_ocr_amd.hip import lib as hiplib
    CP, runtime = rt
    n, h, w, c = shape
    rng = np.random.default_rng(w)
    x = CP.copy(rng.standard_normal(shape))
    y = Guarded(CP, n * w * h * 8 * c)
    runtime.call('uocr_fixed_width_fwd', hiplib.F32, x.ptr, y.ptr, n, h, w, c, 8)
    y.check('fixed width y')
    g = CP.copy(rng.standard_normal((n * w, h, 8, c)))
    dx = Guarded(CP, n * h * w * c)
    runtime.call('uocr_fixed_width_bwd', hiplib.F32, g.ptr, dx.ptr, n, h, w, c, 8)
    dx.check('fixed width dx')
    rows, classes = n * w, 162
    logits = CP.copy(rng.standard_normal((rows, classes)))
    onehot = CP.copy(np.eye(classes)[rng.integers(0, classes, rows)])
    grad = Guarded(CP, rows * classes)
    slot = CP.empty((1,), np.float64)
    runtime.call('uocr_softmax_ce', hiplib.F32, logits.ptr, onehot.ptr, grad.ptr, slot.ptr, rows, classes)
    grad.check('softmax grad')


def test_fused_optimizer_tail_stays_inside_its_buffers(rt):
    import ctypes as C
    from univer_ocr_amd.hip import lib as hiplib
    CP, runtime = rt
    n = 5003
    bufs = {k: Guarded(CP, n, fill=SENTINEL) for k in ('w', 'g', 'v', 'a')}
    for b in bufs.values():                      # payload: small numbers; the borders keep the sentinel
        b.buf.t[GUARD:GUARD + n] = 0.5
    lo, hi = (C.c_longlong * 1)(64), (C.c_longlong * 1)(4096)
    kind, strength = (C.c_int * 1)(2), (C.c_double * 1)(0.01)
    slot = CP.empty((1,), np.float64)
    runtime.call('uocr_momentum_step_fused', hiplib.F32, bufs['w'].ptr, bufs['g'].ptr, bufs['v'].ptr, n, 0.01, 0.9, 1, lo,
                 hi, kind, strength, slot.ptr, 1, None)
    runtime.call('uocr_adam_step_fused', hiplib.F32, bufs['w'].ptr, bufs['g'].ptr, bufs['v'].ptr, bufs['a'].ptr, n, 0.01,
                 0.9, 0.999, 1e-8, 1, lo, hi, kind, strength, slot.ptr, 1, None)
    hyper = CP.copy(np.array([0.01, 0.9, 0.999, 1e-8]), np.float64)      # hyper-parameters from device memory
    runtime.call('uocr_adam_step_fused', hiplib.F32, bufs['w'].ptr, bufs['g'].ptr, bufs['v'].ptr, bufs['a'].ptr, n, 0.0,
                 0.0, 0.0, 0.0, 1, lo, hi, kind, strength, slot.ptr, 1, hyper.ptr)
    for name, b in bufs.items():
        b.check(name, expect_written=False)
    assert not np.any(bufs['g'].check('g', expect_written=False))


@pytest.mark.parametrize('shape,n_out', [((3, 1, 70, 64), 96), ((2, 2, 19, 32), 64), ((2, 1, 8, 64), 32)])
def test_windows_dense_stays_inside_its_buffers(shape, n_out, rt):
    """Conv2DToBatchedFixedWidthed + Flatten + FullyConnected through the conv entry points: kernel (h, 8),
    padding (0, 4), the output cut to W columns (fewer than the padding would allow); dw and its bias row are ONE
    array."""
    from univer_ocr_amd.hip import lib as hiplib
    CP, runtime = rt
    n, h, wd, c = shape
    n_in = h * 8 * c
    rng = np.random.default_rng(wd)
    x = CP.copy(rng.standard_normal(shape))
    w = CP.copy(rng.standard_normal((n_in + 1, n_out)) * 0.1)
    dims = (n, h, wd, c, n_out, h, 8, 1, 1, 0, 4, 1, wd)
    bias_off = n_in * n_out * 4
    y = Guarded(CP, n * wd * n_out)
    runtime.call('uocr_conv2d_fwd', hiplib.F32, x.ptr, w.ptr, w.ptr + bias_off, y.ptr, *dims, 0.0, 1, hiplib.ACT_LEAKY, 0.01)
    y.check('windows y')
    g = CP.copy(rng.standard_normal((n * wd, n_out)))
    dx = Guarded(CP, n * h * wd * c)
    runtime.call('uocr_conv2d_bwd_data', hiplib.F32, g.ptr, w.ptr, dx.ptr, *dims, x.ptr, hiplib.ACT_LEAKY, 0.01)
    dx.check('windows dx')
    dw = Guarded(CP, (n_in + 1) * n_out)
    runtime.call('uocr_conv2d_bwd_weight', hiplib.F32, x.ptr, g.ptr, dw.ptr, dw.ptr + bias_off, *dims, 0.0, 1, 0)
    dw.check('windows dw + bias row')
