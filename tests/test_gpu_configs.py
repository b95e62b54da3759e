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


def test_config4_high_res_pages_fused_equals_layer_by_layer():
    """configs[4] geometry (1024x2048 full-page scans; float32 here -- the float16 storage mode is not built):
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
