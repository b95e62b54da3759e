"""Parity at the sizes BASELINE.json names (configs[0..2] and the geometry of configs[4]); the small oracle-vs-HIP cases live in
test_gpu_ops / test_gpu_models, here the FULL sizes:

  configs[0]  my_model forward on one 64x64 glyph            -> HIP vs oracle, f32 and f64
  configs[1]  forward only, batch 8, 256x512 pages           -> HIP vs oracle (the oracle needs ~2 s)
  configs[2]  train step, batch 32, 256x512 pages            -> size-independent properties on the
              production kernels at full size: exact homogeneity (scaling by 2 is exact in binary
              floating point), batch-split invariance, additivity of dw over the batch, plus one
              Monochrome train step against the oracle at batch 8.
float32 tolerance 1e-5 (normalised max error), 2e-5 for dw."""
import numpy as np
import pytest

from conftest import rel_linf
from oracle import nn_oracle as O

pytestmark = pytest.mark.gpu


def nest(flat):
    out = {}
    for key, value in flat.items():
        layer, pname = key.rsplit('/', 1)
        out.setdefault(layer, {})[pname] = value.tolist()
    return out


def build(name, shape, dtype='float32'):
    from univer_ocr_amd.my_model.model import NET_MAKERS
    from univer_ocr_amd.nn import CP
    from univer_ocr_amd.nn.optimizers import Momentum
    CP.set_dtype(dtype)
    net = O.make_net(name)                         # analytic, RNG-free weights
    model = NET_MAKERS[name](shape, Momentum(lr=0.01, momentum=0))
    model.set_weights(nest(net.params))
    model.enable_fusion()
    return net, model


@pytest.mark.parametrize('dtype,tol', [('float32', 1e-5), ('float64', 1e-12)])
def test_config0_single_glyph_forward(dtype, tol):
    from univer_ocr_amd.nn import CP
    rng = np.random.default_rng(3)
    glyph = np.ones((1, 64, 64, 1))
    glyph[0, 20:44, 28:36, 0] = rng.uniform(0, 0.3, (24, 8))
    try:
        for name in ('Monochrome', 'Paragraph', 'Line'):
            net, model = build(name, glyph.shape, dtype)
            got = CP.asnumpy(model.predict(CP.copy(glyph))[0])
            assert rel_linf(got, net.forward(glyph)) <= tol, name
        strip = glyph[:, 16:48, :, :]
        net, model = build('Char', strip.shape, dtype)
        got = CP.asnumpy(model.predict(CP.copy(strip))[0])
        assert got.shape == (64, 162)
        assert rel_linf(got, net.forward(strip)) <= tol
    finally:
        CP.set_dtype('float32')


def test_config1_forward_batch8_full_pages():
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.nn import CP
    data = make_page_batch(8, 256, 512, 64, seed=11)
    for name, tag in (('Monochrome', 'image'), ('Paragraph', 'monochrome'), ('Line', 'monochrome')):
        net, model = build(name, data[tag].shape)
        got = CP.asnumpy(model.predict(CP.copy(data[tag]))[0])
        err = rel_linf(got, net.forward(data[tag]))
        assert err <= 1e-5, f'{name}: {err:.2e}'
    net, model = build('Char', data['char_lines'].shape)
    got = CP.asnumpy(model.predict(CP.copy(data['char_lines']))[0])
    assert rel_linf(got, net.forward(data['char_lines'])) <= 1e-5


def test_config2_monochrome_train_step_batch8_vs_oracle():
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.nn import CP
    data = make_page_batch(8, 256, 512, 64, seed=12)
    net, model = build('Monochrome', data['image'].shape)
    ref_losses, ref_pred = net.train_step(data['image'], data['monochrome'], O.MomentumState(0.01, 0.0))
    losses = model.train(CP.copy(data['image']), CP.copy(data['monochrome']))
    assert abs(float(losses['output_losses'][0]) - ref_losses['output_losses'][0]) <= 1e-5 * ref_losses['output_losses'][0]
    assert abs(float(losses['regularization_loss']) - ref_losses['regularization_loss']) <= 1e-5
    for pn, p in model.params().items():
        err = rel_linf(CP.asnumpy(p.value), net.params[pn])
        assert err <= 5e-5, f'{pn}: {err:.2e}'


@pytest.mark.parametrize('name,tag_x,tag_y', [('Paragraph', 'monochrome', 'paragraph'), ('Line', 'monochrome', 'line'),
                                              ('Char', 'char_lines', 'char_labels')])
def test_config2_net_train_step_batch8_vs_oracle(name, tag_x, tag_y):
    """The production kernels of the other three nets at the BASELINE configs[2] page size against the oracle:
    Paragraph / Line = uocr_upconv2x_* (1 -> 1 and 4 -> 4), the 5x5 stride-2 kernels (conv_dgrad_s2,
    conv_wgrad_s2_tiled), the Line output conv (conv_fwd_t542 / conv_wgrad_t542 / conv_dgrad_px) and the Dice kernel
    with the folded Sigmoid at 8 x 256 x 512; Char = conv_1 (5x3 stride (2,1), 64 channels), the MFMA convs, the
    windows + flatten + dense_1 implicit GEMM (ops.windows_dense_*), the dense block and softmax cross-entropy at
    8 strips of 32 x 64.  One train step with SGD: prediction, losses, every parameter after the update."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.nn import CP
    data = make_page_batch(8, 256, 512, 64, seed=13)
    net, model = build(name, data[tag_x].shape)
    ref_losses, ref_pred = net.train_step(data[tag_x], data[tag_y], O.MomentumState(0.01, 0.0))
    losses = model.train(CP.copy(data[tag_x]), CP.copy(data[tag_y]))
    assert len(model._ups_used) == (2 if name != 'Char' else 0) and len(model._wins_used) == (name == 'Char')
    err = rel_linf(CP.asnumpy(model.layers_outputs[0]), ref_pred)
    assert err <= 1e-5, f'{name}: prediction {err:.2e}'
    ref = ref_losses['output_losses'][0]
    assert abs(float(losses['output_losses'][0]) - ref) <= 1e-5 * abs(ref)
    assert abs(float(losses['regularization_loss']) - ref_losses['regularization_loss']) <= \
        1e-5 * max(1.0, ref_losses['regularization_loss'])
    for pn, p in model.params().items():
        err = rel_linf(CP.asnumpy(p.value), net.params[pn])
        assert err <= 5e-5, f'{pn}: {err:.2e}'


FULL = [  # (cin, cout, kernel, stride, padding): the production kernels at 32 x 256 x 512
    (1, 16, (3, 3), (1, 1), (1, 1)), (16, 1, (3, 3), (1, 1), (1, 1)), (1, 1, (5, 5), (1, 1), (2, 2)),
    (4, 4, (5, 5), (1, 1), (2, 2)), (4, 2, (5, 5), (1, 1), (2, 2)), (1, 4, (5, 5), (2, 2), (2, 2)),
]


@pytest.mark.parametrize('cin,cout,ks,st,pd', FULL)
def test_config2_full_size_properties(cin, cout, ks, st, pd):
    from univer_ocr_amd.nn import CP, ops
    CP.set_dtype('float32')
    rng = np.random.default_rng(cin * 31 + cout)
    n, h, w = 32, 256, 512
    x = CP.copy(rng.standard_normal((n, h, w, cin)).astype(np.float32))
    wt = CP.copy((rng.standard_normal((*ks, cin, cout)) * 0.2).astype(np.float32))
    zero_b = CP.zeros((cout,))
    y = ops.conv2d_fwd(x, wt, zero_b, st, pd)
    g = CP.copy(rng.standard_normal(y.shape).astype(np.float32))
    two = lambda a: CP.copy(2.0 * CP.asnumpy(a))          # noqa: E731  (exact: power of two)
    # 1. homogeneity, bit-exact
    assert np.array_equal(CP.asnumpy(ops.conv2d_fwd(two(x), wt, zero_b, st, pd)), 2.0 * CP.asnumpy(y))
    dx = ops.conv2d_bwd_data(g, wt, x.shape, st, pd)
    assert np.array_equal(CP.asnumpy(ops.conv2d_bwd_data(two(g), wt, x.shape, st, pd)), 2.0 * CP.asnumpy(dx))
    dw, db = CP.zeros(wt.shape), CP.zeros((cout,))
    ops.conv2d_bwd_weight(x, g, dw, db, st, pd, accumulate=False)
    dw2, db2 = CP.zeros(wt.shape), CP.zeros((cout,))
    ops.conv2d_bwd_weight(x, two(g), dw2, db2, st, pd, accumulate=False)
    assert np.array_equal(CP.asnumpy(dw2), 2.0 * CP.asnumpy(dw)) and np.array_equal(CP.asnumpy(db2), 2.0 * CP.asnumpy(db))
    # 2. images are independent: the batch in two halves gives the same pixels, bit-exact
    xa, xb = CP.copy(CP.asnumpy(x)[:16]), CP.copy(CP.asnumpy(x)[16:])
    halves = np.concatenate([CP.asnumpy(ops.conv2d_fwd(xa, wt, zero_b, st, pd)),
                             CP.asnumpy(ops.conv2d_fwd(xb, wt, zero_b, st, pd))])
    assert np.array_equal(halves, CP.asnumpy(y))
    # 3. dw / db are sums over the batch: halves add up (different partial-sum trees: 2e-5)
    ga, gb = CP.copy(CP.asnumpy(g)[:16]), CP.copy(CP.asnumpy(g)[16:])
    dwa, dba = CP.zeros(wt.shape), CP.zeros((cout,))
    ops.conv2d_bwd_weight(xa, ga, dwa, dba, st, pd, accumulate=False)
    ops.conv2d_bwd_weight(xb, gb, dwa, dba, st, pd, accumulate=True)
    assert rel_linf(CP.asnumpy(dwa), CP.asnumpy(dw).astype(np.float64)) <= 2e-5
    assert rel_linf(CP.asnumpy(dba), CP.asnumpy(db).astype(np.float64)) <= 2e-5
    # 4. a column of the full-size result against the oracle (one image, exact reference arithmetic)
    ref = O.conv2d_fwd(CP.asnumpy(x)[:1].astype(np.float64), CP.asnumpy(wt).astype(np.float64), np.zeros(cout), st, pd)
    assert rel_linf(CP.asnumpy(y)[:1], ref) <= 1e-5


# ---- the FUSED production entry points at the benchmarked batch (their grids depend on the batch size) ---------------
def _dev(a):
    from univer_ocr_amd.nn import CP
    return CP.copy(np.ascontiguousarray(a), np.float32)


def _np(a):
    from univer_ocr_amd.nn import CP
    return CP.asnumpy(a)


def _split_check(run, n, exact, summed, exact_tol=0.0):
    """run(slice) -> dict of host arrays for the images of `slice`.  `exact` keys are per-image results: the batch in
    two halves gives the same bits (exact_tol > 0: kernels whose depth split -- hence float32 summation order --
    depends on the number of rows: equal to rounding); `summed` keys are sums over the batch: the halves add up
    (2e-5: different partial-sum trees)."""
    whole, lo, hi = run(slice(0, n)), run(slice(0, n // 2)), run(slice(n // 2, n))
    for key in exact:
        halves = np.concatenate([lo[key], hi[key]])
        if exact_tol:
            err = rel_linf(halves.astype(np.float64), whole[key].astype(np.float64))
            assert err <= exact_tol, f'{key}: batch halves differ from the whole batch by {err:.2e}'
        else:
            assert np.array_equal(halves, whole[key]), f'{key}: batch halves differ from the whole batch'
    for key in summed:
        err = rel_linf(lo[key].astype(np.float64) + hi[key].astype(np.float64), whole[key].astype(np.float64))
        assert err <= 2e-5, f'{key}: halves do not add up ({err:.2e})'
    return whole


def test_config2_pair_kernels_batch32():
    """uocr_conv_pair_fwd / uocr_conv_pair_bwd (the Monochrome block on the column-strip kernels, csrc/conv_pair_strip.hip)
    at 32 x 256 x 512: batch-split invariance (y, dx bit for bit; dw1, db1, dw2, db2 to 2e-5) and image 0 against the
    oracle run layer by layer."""
    from univer_ocr_amd.hip import lib as hiplib
    from univer_ocr_amd.nn import CP, ops
    CP.set_dtype('float32')
    rng = np.random.default_rng(77)
    n, h, w = 32, 256, 512
    X = rng.random((n, h, w, 1)).astype(np.float32)
    G = rng.standard_normal((n, h, w, 1)).astype(np.float32)
    w1, b1 = rng.standard_normal((3, 3, 1, 16)) * 0.4, rng.standard_normal(16) * 0.3
    w2, b2 = rng.standard_normal((3, 3, 16, 1)) * 0.2, rng.standard_normal(1)
    dws = [_dev(a) for a in (w1, b1, w2, b2)]

    def run(sl):
        x, g = _dev(X[sl]), _dev(G[sl])
        y = ops.conv_pair_fwd(x, *dws, act2=hiplib.ACT_SIGMOID)
        grads = [CP.zeros(a.shape, np.float32) for a in (w1, b1, w2, b2)]
        dx = ops.conv_pair_bwd(x, y, g, dws[0], dws[1], dws[2], *grads, act2=hiplib.ACT_SIGMOID, accumulate=False)
        return dict(y=_np(y), dx=_np(dx), dw1=_np(grads[0]), db1=_np(grads[1]), dw2=_np(grads[2]), db2=_np(grads[3]))
    whole = _split_check(run, n, ('y', 'dx'), ('dw1', 'db1', 'dw2', 'db2'))
    x0, g0 = X[:1].astype(np.float64), G[:1].astype(np.float64)
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)          # noqa: E731  (the weights the kernel saw)
    z1 = O.conv2d_fwd(x0, f32(w1), f32(b1), 1, 1, 0.0, True)
    a1 = O.leaky_relu_fwd(z1, 0.01)
    z2 = O.conv2d_fwd(a1, f32(w2), f32(b2), 1, 1, 0.0, True)
    assert rel_linf(whole['y'][:1], O.sigmoid_fwd(z2)) <= 1e-5
    ga1, _, _ = O.conv2d_bwd(a1, f32(w2), O.sigmoid_bwd(z2, g0), 1, 1, 0.0, True)
    ref_dx, _, _ = O.conv2d_bwd(x0, f32(w1), O.leaky_relu_bwd(z1, ga1, 0.01), 1, 1, 0.0, True)
    assert rel_linf(whole['dx'][:1], ref_dx) <= 2e-5


@pytest.mark.parametrize('ch', [1, 4])
def test_config2_upconv_kernels_batch32(ch):
    """uocr_upconv2x_* (Upsample2D(2) + conv5x5 ch -> ch + LeakyReLU on the low-res tensor: the Paragraph (1 -> 1) and
    Line (4 -> 4) decoder blocks) at the benchmarked size, low-res 32 x 128 x 256: batch-split invariance and image 0
    against the oracle's two layers."""
    from univer_ocr_amd.nn import CP, ops
    CP.set_dtype('float32')
    rng = np.random.default_rng(78 + ch)
    n, hl, wl = 32, 128, 256
    X = rng.standard_normal((n, hl, wl, ch)).astype(np.float32)
    G = rng.standard_normal((n, 2 * hl, 2 * wl, ch)).astype(np.float32)
    wt, b = rng.standard_normal((5, 5, ch, ch)) * 0.2, rng.standard_normal(ch) * 0.1
    dwt, dbias = _dev(wt), _dev(b)

    def run(sl):
        x, g = _dev(X[sl]), _dev(G[sl])
        y = ops.upconv2x_fwd(x, dwt, dbias, (2, 2), True, 'leaky', 0.01)
        dx = ops.upconv2x_bwd_data(g, dwt, x.shape, (2, 2))
        dw, db = CP.zeros(wt.shape, np.float32), CP.zeros(b.shape, np.float32)
        ops.upconv2x_bwd_weight(x, g, dw, db, (2, 2), True, accumulate=False)
        return dict(y=_np(y), dx=_np(dx), dw=_np(dw), db=_np(db))
    whole = _split_check(run, n, ('y', 'dx'), ('dw', 'db'))
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)          # noqa: E731
    up = O.upsample2d_fwd(X[:1].astype(np.float64), (2, 2))
    pre = O.conv2d_fwd(up, f32(wt), f32(b), 1, 2, 0.0, True)
    assert rel_linf(whole['y'][:1], O.leaky_relu_fwd(pre, 0.01)) <= 1e-5
    dup, _, _ = O.conv2d_bwd(up, f32(wt), G[:1].astype(np.float64), 1, 2, 0.0, True)
    assert rel_linf(whole['dx'][:1], O.upsample2d_bwd(dup, (2, 2))) <= 2e-5


def test_config2_windows_dense_batch32():
    """Conv2DToBatchedFixedWidthed(8) + Flatten + FullyConnected(513 -> 1024) of the Char net as one implicit GEMM
    (ops.windows_dense_*: split-depth MFMA GEMMs at M = 32 * 64 rows) at the benchmarked batch: batch-split invariance
    and image 0 against the oracle's three layers."""
    from univer_ocr_amd.nn import CP, ops
    CP.set_dtype('float32')
    rng = np.random.default_rng(79)
    n, wd, c, n_out = 32, 64, 64, 1024
    X = rng.standard_normal((n, 1, wd, c)).astype(np.float32)
    G = rng.standard_normal((n * wd, n_out)).astype(np.float32)
    W = (rng.standard_normal((8 * c + 1, n_out)) * 0.1).astype(np.float32)
    dW = _dev(W)

    def run(sl):
        x = _dev(X[sl])
        rows = slice(sl.start * wd, sl.stop * wd)
        g = _dev(G[rows])
        y = ops.windows_dense_fwd(x, dW, 8, 'leaky', 0.01)
        dw = CP.zeros(W.shape, np.float32)
        dx = ops.windows_dense_bwd(x, dW, g, dw, 8, accumulate=False, x_act=x, act='leaky', alpha=0.01)
        return dict(y=_np(y), dx=_np(dx), dw=_np(dw))
    # (the MFMA GEMMs split their depth until the chip is full: 16 and 32 strips get different slab counts)
    whole = _split_check(run, n, ('y', 'dx'), ('dw',), exact_tol=2e-6)
    windows = O.fixed_width_fwd(X[:1].astype(np.float64), 8)
    flat = windows.reshape(wd, -1)
    pre = O.dense_fwd(flat, W.astype(np.float64))
    assert rel_linf(whole['y'][:wd], np.where(pre >= 0, pre, 0.01 * pre)) <= 1e-5
    dflat, _ = O.dense_bwd(flat, W.astype(np.float64), G[:wd].astype(np.float64))
    ref_dx = O.fixed_width_bwd(dflat.reshape(windows.shape), X[:1].shape, 8) * np.where(X[:1] >= 0, 1.0, 0.01)
    assert rel_linf(whole['dx'][:1], ref_dx) <= 1e-5


def test_config2_line_end_dx_batch32():
    """Backward-data of the Line output conv (5x5, 4 <- 2) with the folded LeakyReLU' of its input, i.e. the default
    conv_t32_kernel (float32-MFMA Toeplitz rows, csrc/conv_t32.hip) at 32 x 256 x 512: batch-split invariance and image 0
    against the oracle."""
    from univer_ocr_amd.nn import CP, ops
    CP.set_dtype('float32')
    rng = np.random.default_rng(80)
    n, h, w = 32, 256, 512
    A = rng.standard_normal((n, h, w, 4)).astype(np.float32)           # the conv's input = a LeakyReLU output
    G = rng.standard_normal((n, h, w, 2)).astype(np.float32)
    wt = (rng.standard_normal((5, 5, 4, 2)) * 0.2).astype(np.float32)
    dwt = _dev(wt)

    def run(sl):
        a, g = _dev(A[sl]), _dev(G[sl])
        return dict(dx=_np(ops.conv2d_bwd_data(g, dwt, a.shape, (1, 1), (2, 2), x_act=a, act='leaky', alpha=0.01)))
    whole = _split_check(run, n, ('dx',), ())
    ref, _, _ = O.conv2d_bwd(A[:1].astype(np.float64), wt.astype(np.float64), G[:1].astype(np.float64), 1, 2, 0.0, True)
    assert rel_linf(whole['dx'][:1], ref * np.where(A[:1] >= 0, 1.0, 0.01)) <= 2e-5


def test_config4_high_res_pages_fused_equals_layer_by_layer():
    """configs[4] geometry (1024x2048 full-page scans) in float32 (the float16 storage mode of the same configuration:
    tests/test_gpu_f16.py::test_config4_highres_f16_*):
    the multi-layer kernels (conv pair, upsample+conv on the low-res tensor, loss with the folded Sigmoid)
    against the layer-by-layer kernels on the same weights and pages at full resolution: losses of two train
    steps and the weights after them.  Covers the 64-bit offsets of 2-page batches of 2M-pixel images
    (the unfused Monochrome activation alone is 268 MB)."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    layers = make_page_batch(2, 1024, 2048, 64, seed=21)
    results = []
    for fuse in (True, False):
        trainer = PageTrainer(2, 1024, 2048, 64, optimizer='sgd', lr=0.0015, seed=9, fuse=fuse,
                              nets=('Monochrome', 'Paragraph', 'Line'))
        context = trainer.make_context(layers)
        rows = []
        for _ in range(2):
            losses = trainer.step(context)
            rows.append([float(v) for name in ('Monochrome', 'Paragraph', 'Line') for v in losses[name]['output_losses']])
        weights = {n: p.value.numpy() for m in trainer.models.values() for n, p in m.params().items()}
        results.append((np.array(rows), weights))
        used = {name: (len(m._pairs_used), len(m._ups_used)) for name, m in trainer.models.items()}
        assert used == ({'Monochrome': (1, 0), 'Paragraph': (0, 2), 'Line': (0, 2)} if fuse else
                        {'Monochrome': (0, 0), 'Paragraph': (0, 0), 'Line': (0, 0)})
        del trainer, context
    assert np.all(np.isfinite(results[0][0]))
    assert rel_linf(results[0][0], results[1][0]) <= 1e-5
    for name, w in results[0][1].items():
        assert rel_linf(w, results[1][1][name]) <= 5e-5, name
