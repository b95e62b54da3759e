"""GPU parity of every op of the hot path, called through the C ABI (libuniver_hip.so via
univer_ocr_amd.nn.ops), against the golden vectors produced by the reference (tests/golden) and
against oracle/ on seeded inputs.

Tolerance (normalised max error  max|hip - ref| / max|ref|):
    float64: 1e-12   (same arithmetic as the reference, only the summation order differs)
    float32: 1e-5    (BASELINE.json north_star: "outputs within 1e-5 of the NumPy reference")
"""
import numpy as np
import pytest

from conftest import load_golden, rel_linf

pytestmark = pytest.mark.gpu

TOL = {'float32': 1e-5, 'float64': 1e-12}


@pytest.fixture(params=['float32', 'float64'])
def dt(request):
    from univer_ocr_amd.nn import CP
    CP.set_dtype(request.param)
    yield request.param
    CP.set_dtype('float32')


def close(a, b, dt, scale=1.0):
    from univer_ocr_amd.nn import CP
    err = rel_linf(CP.asnumpy(a), b)
    assert err <= TOL[dt] * scale, f'rel_linf={err:.3e} > {TOL[dt] * scale:.1e}'


def dev(a):
    from univer_ocr_amd.nn import CP
    return CP.copy(a)


def test_library_loaded_is_in_tree():
    from univer_ocr_amd.hip import get_lib, lib_path
    assert get_lib().uocr_abi_version() == 4
    assert lib_path().endswith('univer-ocr_amd/libuniver_hip.so')


def test_graph_capture_and_replay_with_ctypes_only():
    """INTEGRATION.md section 7 as a reference-side binding would write it: ctypes, a context and buffers of the library
    itself (no torch tensor, stream or graph anywhere) -- one SGD step of conv3x3 (1 -> 4) + Dice loss is captured with
    uocr_graph_begin_capture / end_capture and replayed with uocr_graph_launch on new inputs; every replay equals the
    same calls made eagerly, and the oracle."""
    import ctypes as C
    from oracle import nn_oracle as O
    from univer_ocr_amd.hip import lib as hiplib
    lib = hiplib.get_lib()
    ctx = C.c_void_p()
    assert lib.uocr_ctx_create(0, 64 << 20, C.byref(ctx)) == 0
    n, h, w, cout = 2, 24, 40, 4
    rng = np.random.default_rng(5)
    wt = (rng.standard_normal((3, 3, 1, cout)) * 0.3).astype(np.float32)
    bias = (rng.standard_normal(cout) * 0.1).astype(np.float32)

    def buf(nbytes):
        p = C.c_void_p()
        assert lib.uocr_malloc(ctx, nbytes, C.byref(p)) == 0
        return p

    def up(ptr, a):
        a = np.ascontiguousarray(a)
        assert lib.uocr_h2d(ctx, ptr, a.ctypes.data_as(C.c_void_p), a.nbytes) == 0

    def down(ptr, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        assert lib.uocr_d2h_sync(ctx, out.ctypes.data_as(C.c_void_p), ptr, out.nbytes) == 0
        return out
    px, py = n * h * w, n * h * w * cout
    X, G, Y, DY, W, B, DW, DB, V, slot = (buf(4 * px), buf(4 * py), buf(4 * py), buf(4 * py), buf(wt.nbytes), buf(bias.nbytes),
                                          buf(wt.nbytes), buf(bias.nbytes), buf(wt.nbytes), buf(8))
    dims = (n, h, w, 1, cout, 3, 3, 1, 1, 1, 1, h, w)

    def step():
        assert lib.uocr_conv2d_fwd(ctx, hiplib.F32, X, W, B, Y, *dims, 0.0, 1, hiplib.ACT_SIGMOID, 0.0) == 0
        assert lib.uocr_seg_loss(ctx, hiplib.F32, hiplib.LOSS_DICE, Y, G, DY, slot, n, h * w, cout, hiplib.ACT_SIGMOID) == 0
        assert lib.uocr_conv2d_bwd_weight(ctx, hiplib.F32, X, DY, DW, DB, *dims, 0.0, 1, 0) == 0
        assert lib.uocr_momentum_step(ctx, hiplib.F32, W, DW, V, wt.size, 0.05, 0.0) == 0

    def reset():
        up(W, wt), up(B, bias)
        assert lib.uocr_memset_zero(ctx, V, wt.nbytes) == 0
    reset()
    graph = C.c_void_p()
    assert lib.uocr_graph_begin_capture(ctx) == 0
    step()                                               # recorded, not run
    assert lib.uocr_graph_end_capture(ctx, C.byref(graph)) == 0 and graph.value
    batches = [(rng.random((n, h, w, 1)).astype(np.float32), (rng.random((n, h, w, cout)) > 0.6).astype(np.float32))
               for _ in range(3)]
    replayed, eager = [], []
    for x, g in batches:                                 # three replays on three batches: the weights move every time
        up(X, x), up(G, g)
        assert lib.uocr_graph_launch(ctx, graph) == 0
        replayed.append((down(slot, (1,), np.float64)[0], down(W, wt.shape)))
    reset()
    for x, g in batches:
        up(X, x), up(G, g)
        step()
        eager.append((down(slot, (1,), np.float64)[0], down(W, wt.shape)))
    for (lr, wr), (le, we) in zip(replayed, eager):
        assert lr == le and np.array_equal(wr, we)
    # and the first step against the oracle
    x, g = batches[0]
    z = O.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, 1, 0.0, True)
    loss, grad = O.dice_loss(O.sigmoid_fwd(z), g.astype(np.float64))
    assert abs(replayed[0][0] - loss) <= 1e-5 * abs(loss)
    _, dw, _ = O.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), O.sigmoid_bwd(z, grad), 1, 1, 0.0, True)
    assert rel_linf(replayed[0][1].astype(np.float64), wt - 0.05 * dw) <= 1e-5
    assert lib.uocr_graph_destroy(graph) == 0
    for p in (X, G, Y, DY, W, B, DW, DB, V, slot):
        assert lib.uocr_free(ctx, p) == 0
    assert lib.uocr_ctx_destroy(ctx) == 0


@pytest.mark.parametrize('tag', [str(n) for n in load_golden('conv2d')['names']])
def test_conv2d(tag, dt):
    from univer_ocr_amd.nn import CP, ops
    g = load_golden('conv2d')
    kh, kw, sh, sw, ph, pw, pv, bias = g[f'{tag}/cfg']
    st, pd = (int(sh), int(sw)), (int(ph), int(pw))
    X, w, b, gr = dev(g[f'{tag}/X']), dev(g[f'{tag}/w']), dev(g[f'{tag}/b']), dev(g[f'{tag}/g'])
    y = ops.conv2d_fwd(X, w, b, st, pd, pv, bool(bias))
    close(y, g[f'{tag}/y'], dt)
    dx = ops.conv2d_bwd_data(gr, w, X.shape, st, pd)
    close(dx, g[f'{tag}/dx'], dt)
    dw, db = CP.zeros(w.shape), CP.zeros(b.shape)
    ops.conv2d_bwd_weight(X, gr, dw, db, st, pd, pv, bool(bias), accumulate=True)
    close(dw, g[f'{tag}/dw'], dt)
    if bias:
        close(db, g[f'{tag}/db'], dt)
    else:
        assert not CP.asnumpy(db).any()
    # accumulate=True adds on top (self.w.grad += dw_total, convolutional.py:137)
    ops.conv2d_bwd_weight(X, gr, dw, db, st, pd, pv, bool(bias), accumulate=True)
    close(dw, 2 * g[f'{tag}/dw'], dt)
    ops.conv2d_bwd_weight(X, gr, dw, db, st, pd, pv, bool(bias), accumulate=False)
    close(dw, g[f'{tag}/dw'], dt)


@pytest.mark.parametrize('act,alpha', [('relu', 0.0), ('leaky', 0.01), ('sigmoid', 0.0)])
def test_conv2d_fused_activation(act, alpha, dt):
    from univer_ocr_amd.nn import ops
    from oracle import nn_oracle as O
    g = load_golden('conv2d')
    X, w, b = g['ti_pad/X'], g['ti_pad/w'], g['ti_pad/b']
    y = ops.conv2d_fwd(dev(X), dev(w), dev(b), (1, 1), (1, 1), 0.0, True, act=act, alpha=alpha)
    ref = O.conv2d_fwd(X, w, b, 1, 1)
    ref = {'relu': O.relu_fwd, 'leaky': lambda v: O.leaky_relu_fwd(v, alpha), 'sigmoid': O.sigmoid_fwd}[act](ref)
    close(y, ref, dt)


@pytest.mark.parametrize('tag', [str(n) for n in load_golden('maxpool2d')['names']])
def test_maxpool2d(tag, dt):
    from univer_ocr_amd.nn import CP, ops
    g = load_golden('maxpool2d')
    kh, kw, sh, sw, ph, pw, ceil = (int(v) for v in g[f'{tag}/cfg'])
    X = dev(g[f'{tag}/X'])
    y, mask = ops.maxpool2d_fwd(X, (kh, kw), (sh, sw), (ph, pw), bool(ceil))
    ref = g[f'{tag}/y'].astype(dt)
    assert np.array_equal(CP.asnumpy(y), ref)          # a max is exact in any precision
    dx = ops.maxpool2d_bwd(dev(g[f'{tag}/g']), mask, X.shape, (kh, kw), (sh, sw), (ph, pw))
    close(dx, g[f'{tag}/dx'], dt)


@pytest.mark.parametrize('tag', [str(n) for n in load_golden('upsample2d')['names']])
def test_upsample2d(tag, dt):
    from univer_ocr_amd.nn import CP, ops
    g = load_golden('upsample2d')
    sf = tuple(int(v) for v in g[f'{tag}/cfg'])
    X = dev(g[f'{tag}/X'])
    y = ops.upsample2d_fwd(X, sf)
    assert np.array_equal(CP.asnumpy(y), g[f'{tag}/y'].astype(dt))
    close(ops.upsample2d_bwd(dev(g[f'{tag}/g']), X.shape, sf), g[f'{tag}/dx'], dt)


def test_activations(dt):
    from univer_ocr_amd.nn import CP, ops
    g = load_golden('layers')
    X, gr = dev(g['X']), dev(g['g'])
    for tag, kind, alpha in (('relu', 'relu', 0.0), ('leaky', 'leaky', 0.01), ('leaky03', 'leaky', 0.3),
                             ('sigmoid', 'sigmoid', 0.0)):
        close(ops.act_fwd(kind, X, alpha), g[f'{tag}/y'], dt)
        close(ops.act_bwd(kind, X, gr, alpha), g[f'{tag}/dx'], dt)
    # mask semantics at exactly zero: X >= 0 passes the gradient (layers.py:379,396)
    zero_grad = CP.asnumpy(ops.act_bwd('relu', X, gr))[0, 0, 0, :3]
    assert np.array_equal(zero_grad, g['g'][0, 0, 0, :3].astype(dt))
    Xw, gw = dev(g['sigmoid_wide/X']), dev(g['sigmoid_wide/g'])
    close(ops.act_fwd('sigmoid', Xw), g['sigmoid_wide/y'], dt)
    close(ops.act_bwd('sigmoid', Xw, gw), g['sigmoid_wide/dx'], dt)


@pytest.mark.parametrize('tag', ['fc_small', 'fc_mid', 'fc_char'])
def test_dense(tag, dt):
    from univer_ocr_amd.nn import CP, ops
    g = load_golden('layers')
    X, w, gr = dev(g[f'{tag}/X']), dev(g[f'{tag}/w']), dev(g[f'{tag}/g'])
    close(ops.dense_fwd(X, w), g[f'{tag}/y'], dt)
    dw = CP.zeros(w.shape)
    dx = ops.dense_bwd(X, w, gr, dw, accumulate=True)
    close(dx, g[f'{tag}/dx'], dt)
    close(dw, g[f'{tag}/dw'], dt)
    ops.dense_bwd(X, w, gr, dw, accumulate=True, need_dx=False)
    close(dw, 2 * g[f'{tag}/dw'], dt)


@pytest.mark.parametrize('tag', ['fw3', 'fw8', 'fw2'])
def test_fixed_width(tag, dt):
    from univer_ocr_amd.nn import CP, ops
    g = load_golden('layers')
    X, width = dev(g[f'{tag}/X']), int(g[f'{tag}/width'])
    assert np.array_equal(CP.asnumpy(ops.fixed_width_fwd(X, width)), g[f'{tag}/y'].astype(dt))
    close(ops.fixed_width_bwd(dev(g[f'{tag}/g']), X.shape, width), g[f'{tag}/dx'], dt)


def test_concat_split_add(dt):
    from univer_ocr_amd.nn import CP, ops
    g = load_golden('layers')
    parts = [g['concat3/a'], g['concat3/b'], g['concat3/c']]
    y = ops.concat([dev(p) for p in parts])
    assert np.array_equal(CP.asnumpy(y), g['concat3/y'].astype(dt))
    outs = ops.split(dev(g['concat3/g']), [p.shape for p in parts])
    for o, key in zip(outs, ('da', 'db', 'dc')):
        assert np.array_equal(CP.asnumpy(o), g[f'concat3/{key}'].astype(dt))
    a, b = g['concat3/a'].astype(dt), g['concat3/a'].astype(dt)[::-1].copy()
    assert np.array_equal(CP.asnumpy(dev(a) + dev(b)), a + b)
    # axis-0 concat (rows == 1 path)
    y0 = ops.concat([dev(parts[0]), dev(parts[0])], axis=0)
    assert np.array_equal(CP.asnumpy(y0), np.concatenate([parts[0], parts[0]], axis=0).astype(dt))


def test_losses(dt):
    from univer_ocr_amd.nn import ops
    g = load_golden('losses_reg')
    pred, gt = dev(g['seg/pred']), dev(g['seg/gt'])
    for tag, kind, gtt in (('dice', 'dice', gt), ('jaccard', 'jaccard', gt), ('dice_zero', 'dice', dev(g['dice_zero/gt']))):
        loss, grad = ops.seg_loss(kind, pred, gtt)
        assert abs(loss - float(g[f'{tag}/loss'])) <= TOL[dt] * max(1.0, abs(float(g[f'{tag}/loss'])))
        close(grad, g[f'{tag}/grad'], dt)
        loss2, none = ops.seg_loss(kind, pred, gtt, need_grad=False)
        assert none is None and loss2 == loss
    for tag in ('softmax_ce', 'softmax_ce162'):
        loss, grad = ops.softmax_ce(dev(g[f'{tag}/pred']), dev(g[f'{tag}/gt']))
        assert abs(loss - float(g[f'{tag}/loss'])) <= TOL[dt] * max(1.0, abs(float(g[f'{tag}/loss'])))
        close(grad, g[f'{tag}/grad'], dt)
    loss, grad = ops.sigmoid_ce(dev(g['softmax_ce/pred']), dev(g['sigmoid_ce/gt']))
    assert abs(loss - float(g['sigmoid_ce/loss'])) <= TOL[dt] * max(1.0, abs(float(g['sigmoid_ce/loss'])))
    close(grad, g['sigmoid_ce/grad'], dt)


def test_regularizers(dt):
    from univer_ocr_amd.nn import CP
    from univer_ocr_amd.nn.regularizations import L1, L2
    g = load_golden('losses_reg')
    w = dev(g['reg/w'])
    for tag, reg in (('l1', L1(0.1)), ('l2', L2(0.01))):
        loss, grad = reg(w)
        assert abs(loss - float(g[f'{tag}/loss'])) <= TOL[dt] * abs(float(g[f'{tag}/loss']))
        close(grad, g[f'{tag}/grad'], dt)
    # fused form: adds into an existing gradient and an existing loss slot
    grad = CP.full(w.shape, 1.0)
    slot = CP.full((1,), 5.0, np.float64)
    L2(0.01).apply(w, grad, slot, accumulate=True)
    close(grad, 1.0 + g['l2/grad'], dt)
    assert abs(float(slot.numpy()[0]) - (5.0 + float(g['l2/loss']))) <= TOL[dt] * 6


def test_optimizers(dt):
    from univer_ocr_amd.nn import optimizers as opt
    from univer_ocr_amd.nn.layers import Param
    g = load_golden('optimizers')
    for tag, make in (('adam', lambda: opt.Adam(lr=0.0015)),
                      ('adam_b', lambda: opt.Adam(lr=0.01, beta1=0.8, beta2=0.9)),
                      ('sgd', lambda: opt.Momentum(lr=0.05, momentum=0)),
                      ('momentum', lambda: opt.Momentum(lr=0.05, momentum=0.9)),
                      ('rmsprop', lambda: opt.RMSProp(lr=0.01, rho=0.95))):
        o = make()
        p = Param(g['w0'], optimizer=o)
        for k in range(3):
            p.grad = g[f'g{k}']
            p.update_grad()
            close(p.value, g[f'{tag}/w{k + 1}'], dt, scale=4)
    o = opt.Adagrad(lr=0.01)
    p = Param(g['w0'], optimizer=o)
    with pytest.raises(AttributeError):        # the reference's Adagrad raises on first use
        p.update_grad()
    assert str(g['adagrad/error']) == 'AttributeError'


def test_has_nan_fill_scale(dt):
    from univer_ocr_amd.nn import CP, ops
    x = np.arange(1000, dtype=dt).reshape(10, 100)
    d = dev(x)
    assert not ops.has_nan(d)
    x[7, 13] = np.nan
    assert ops.has_nan(dev(x))
    ops.fill_(d, 2.5)
    assert np.all(CP.asnumpy(d) == 2.5)
    ops.scale_(d, 2.0)
    assert np.all(CP.asnumpy(d) == 5.0)
    y = dev(np.ones((10, 100)))
    ops.axpy(3.0, d, y)
    assert np.all(CP.asnumpy(y) == 16.0)
    u8 = CP.copy(np.arange(256, dtype=np.uint8), np.uint8)
    f = ops.u8_to_float(u8)
    assert np.allclose(CP.asnumpy(f), np.arange(256) / 255.0, rtol=1e-6)


def test_error_paths_fail_loudly():
    """Wrong shapes / dtypes must raise, never fall back (help_func.py:24-29, layers.py:58)."""
    from univer_ocr_amd.hip import HipError
    from univer_ocr_amd.nn import CP, ops
    from univer_ocr_amd.nn.layers import Convolutional2D
    with pytest.raises(ValueError):
        Convolutional2D((3, 3), 1, 1, padding=-1)
    with pytest.raises(TypeError):
        Convolutional2D('3x3', 1, 1)
    x = CP.zeros((1, 4, 4, 2), 'float32')
    w = CP.zeros((3, 3, 3, 1), 'float32')
    with pytest.raises(AssertionError):
        ops.conv2d_fwd(x, w, CP.zeros((1,), 'float32'), (1, 1), (0, 0))
    with pytest.raises(HipError):
        ops.conv2d_fwd(x, CP.zeros((3, 3, 2, 1), 'float64'), CP.zeros((1,), 'float32'), (1, 1), (0, 0))
    with pytest.raises(HipError):
        ops.conv2d_fwd(x, CP.zeros((5, 5, 2, 1), 'float32'), CP.zeros((1,), 'float32'), (1, 1), (0, 0))
    with pytest.raises(NotImplementedError):
        CP.use_cpu()
