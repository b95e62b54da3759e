#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference).  It imports the
reference's NumPy CPU path (web_app/components/nn, my_model/model.py) with the
local stand-ins listed in SURVEY.md section 8(c) for modules the CPU path never
executes (cupy, numba.cuda.jit, faker) and for aliases that newer
Python/NumPy removed (collections.Iterable, np.float, np.bool, np.product).
Nothing of the reference's source is copied: the script calls the reference's
classes on seeded inputs and stores inputs + outputs as float64 .npz data.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The fixtures are what pins oracle/ (tests/test_oracle_golden.py) and what the
GPU parity tests compare against on the GPU box, where /root/reference does
not exist.
"""
import collections
import collections.abc
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get('UNIVER_REFERENCE', '/root/reference')
OUT_DIR = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------
# shims
# ----------------------------------------------------------------------------
def install_shims():
    if not hasattr(collections, 'Iterable'):
        collections.Iterable = collections.abc.Iterable
    for alias, target in (('float', float), ('bool', bool), ('product', np.prod)):
        if not hasattr(np, alias):
            setattr(np, alias, target)

    cupy = types.ModuleType('cupy')
    cupy.asarray = np.asarray
    cupy.asnumpy = np.asarray
    cupy.ndarray = np.ndarray
    sys.modules.setdefault('cupy', cupy)

    numba = types.ModuleType('numba')
    cuda = types.ModuleType('numba.cuda')

    def jit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn
    cuda.jit = jit
    numba.cuda = cuda
    sys.modules.setdefault('numba', numba)
    sys.modules.setdefault('numba.cuda', cuda)

    faker = types.ModuleType('faker')

    class Faker:  # never instantiated on the nn path
        def __init__(self, *a, **k):
            pass
    faker.Faker = Faker
    sys.modules.setdefault('faker', faker)

    sys.path.insert(0, os.path.join(REF_ROOT, 'web_app'))


install_shims()

from components.nn import losses as ref_losses  # noqa: E402
from components.nn import optimizers as ref_opt  # noqa: E402
from components.nn import regularizations as ref_reg  # noqa: E402
from components.nn.layers import (  # noqa: E402
    Concat, Conv2DToBatchedFixedWidthed, Convolutional2D, Flatten, FullyConnected, LeakyRelu,
    MaxPool2D, Noop, Relu, Sigmoid, Upsample2D)
from components.nn.models import Model, Sequential  # noqa: E402


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    size = os.path.getsize(path)
    print(f'{name}.npz: {len(arrays)} arrays, {size / 1024:.1f} KiB')


def rng(seed):
    return np.random.default_rng(seed)


# ----------------------------------------------------------------------------
# single layers
# ----------------------------------------------------------------------------
def conv_cases():
    """(tag, X shape, kernel, cin, cout, kwargs).  First five = test_gradients.py:130-165
    shapes, next five = test_identity.py:9-25 at reduced H,W, the rest = my_model shapes."""
    cases = []
    variants = [
        ('plain', {}),
        ('pad', dict(padding=1)),
        ('padval', dict(padding=1, padding_value=0.5)),
        ('stride', dict(stride=2)),
        ('padstride', dict(padding=1, stride=2)),
    ]
    for tag, kw in variants:
        cases.append((f'tg_{tag}', (3, 5, 5, 6), (4, 4), 6, 7, kw))
    for tag, kw in variants:
        cases.append((f'ti_{tag}', (2, 12, 16, 6), (3, 3), 6, 7, kw))
    cases += [
        ('mono1', (2, 8, 12, 1), (3, 3), 1, 16, dict(padding=1)),
        ('mono2', (2, 8, 12, 16), (3, 3), 16, 1, dict(padding=1)),
        ('para_down', (2, 16, 24, 1), (5, 5), 1, 1, dict(padding=2, stride=2)),
        ('para_same', (2, 8, 12, 1), (5, 5), 1, 1, dict(padding=2)),
        ('line_down', (2, 16, 24, 4), (5, 5), 4, 4, dict(padding=2, stride=2)),
        ('line_end', (2, 8, 12, 4), (5, 5), 4, 2, dict(padding=2)),
        ('char1', (2, 32, 10, 1), (5, 3), 1, 64, dict(padding=(0, 1), stride=(2, 1))),
        ('char2', (2, 14, 10, 64), (5, 3), 64, 64, dict(padding=(0, 1), stride=(2, 1))),
        ('char3', (2, 5, 10, 64), (5, 3), 64, 64, dict(padding=(0, 1), stride=(2, 1))),
        ('nobias', (2, 6, 7, 3), (3, 2), 3, 5, dict(padding=(1, 0), bias=False)),
        ('odd_stride', (1, 9, 11, 2), (3, 3), 2, 3, dict(padding=(2, 1), stride=(3, 2),
                                                        padding_value=-0.25)),
        ('k1', (2, 4, 5, 3), (1, 1), 3, 4, {}),
    ]
    return cases


def gen_conv():
    out = {}
    names = []
    for i, (tag, xs, ks, cin, cout, kw) in enumerate(conv_cases()):
        r = rng(100 + i)
        w = r.standard_normal((*ks, cin, cout)) * 0.3
        b = r.standard_normal((cout,)) * 0.3
        layer = Convolutional2D(ks, cin, cout, w=w.copy(), b=b.copy(), **kw)
        X = r.standard_normal(xs)
        y = layer.forward(X)[0]
        g = r.standard_normal(y.shape)
        layer.clear_grads()
        y = layer.forward(X)[0]
        dx = layer.backward(g)[0]
        names.append(tag)
        out[f'{tag}/X'] = X
        out[f'{tag}/w'] = w
        out[f'{tag}/b'] = b
        out[f'{tag}/y'] = y
        out[f'{tag}/g'] = g
        out[f'{tag}/dx'] = dx
        out[f'{tag}/dw'] = layer.w.grad
        out[f'{tag}/db'] = layer.b.grad
        cfg = dict(padding=0, padding_value=0, stride=1, bias=True)
        cfg.update(kw)
        pad = cfg['padding'] if isinstance(cfg['padding'], tuple) else (cfg['padding'],) * 2
        st = cfg['stride'] if isinstance(cfg['stride'], tuple) else (cfg['stride'],) * 2
        out[f'{tag}/cfg'] = np.array(
            [ks[0], ks[1], st[0], st[1], pad[0], pad[1], float(cfg['padding_value']),
             float(bool(cfg['bias']))], dtype=np.float64)
    out['names'] = np.array(names)
    save('conv2d', **out)


def gen_pool_upsample():
    out = {}
    names = []
    pool_variants = [
        ('plain', (2, 2), {}),
        ('pad', (2, 2), dict(padding=1)),
        ('stride1', (2, 2), dict(stride=1)),
        ('padstride1', (2, 2), dict(padding=1, stride=1)),
        ('k3', (3, 3), {}),
        ('k3s2ceil', (3, 3), dict(stride=2, ceil_mode=True)),
        ('k32', (3, 2), dict(padding=(1, 0))),
    ]
    for i, (tag, ks, kw) in enumerate(pool_variants):
        r = rng(200 + i)
        X = r.standard_normal((2, 7, 9, 3))
        if tag in ('plain', 'stride1'):
            # force ties inside windows (tie-splitting branch, maxpool.py:52,80-83)
            X[0, 0:2, 0:2, 0] = 1.5
            X[1, 2:4, 4:6, 1] = 2.5
            X[1, 2, 4, 1] = 0.5
        layer = MaxPool2D(ks, **kw)
        y = layer.forward(X)[0]
        g = r.standard_normal(y.shape)
        y = layer.forward(X)[0]
        dx = layer.backward(g)[0]
        names.append(tag)
        out[f'{tag}/X'], out[f'{tag}/y'], out[f'{tag}/g'], out[f'{tag}/dx'] = X, y, g, dx
        pad = kw.get('padding', 0)
        pad = pad if isinstance(pad, tuple) else (pad, pad)
        st = kw.get('stride', None)
        st = ks if st is None else (st if isinstance(st, tuple) else (st, st))
        out[f'{tag}/cfg'] = np.array(
            [ks[0], ks[1], st[0], st[1], pad[0], pad[1], float(kw.get('ceil_mode', False))])
    # known answer of test_gradients.py:171-177
    X = np.array([[1, 0, 1, 2], [0, -1, -1, -1], [-1, -1, 1, -2]], dtype=float).reshape(1, 3, 4, 1)
    layer = MaxPool2D(2, ceil_mode=True)
    y = layer.forward(X)[0]
    g = np.arange(1, 1 + y.size, dtype=float).reshape(y.shape)
    dx = layer.backward(g)[0]
    names.append('known_ceil')
    out['known_ceil/X'], out['known_ceil/y'] = X, y
    out['known_ceil/g'], out['known_ceil/dx'] = g, dx
    out['known_ceil/cfg'] = np.array([2, 2, 2, 2, 0, 0, 1.0])
    out['names'] = np.array(names)
    save('maxpool2d', **out)

    out = {}
    names = []
    for i, (tag, sf, xs) in enumerate([
            ('s2', (2, 2), (2, 5, 6, 3)), ('s23', (2, 3), (2, 4, 3, 2)), ('s5', (5, 5), (1, 2, 2, 3)),
            ('s31', (3, 1), (2, 3, 4, 1))]):
        r = rng(300 + i)
        X = r.standard_normal(xs)
        layer = Upsample2D(sf)
        y = layer.forward(X)[0]
        g = r.standard_normal(y.shape)
        dx = layer.backward(g)[0]
        names.append(tag)
        out[f'{tag}/X'], out[f'{tag}/y'], out[f'{tag}/g'], out[f'{tag}/dx'] = X, y, g, dx
        out[f'{tag}/cfg'] = np.array(sf, dtype=float)
    # known answer of test_gradients.py:181-188
    X = np.array([[0.1, 0.2], [0.3, 0.4]]).reshape(1, 2, 2, 1).repeat(4, axis=0).repeat(3, axis=-1)
    layer = Upsample2D((2, 3))
    y = layer.forward(X)[0]
    dx = layer.backward(y)[0]
    names.append('known')
    out['known/X'], out['known/y'], out['known/g'], out['known/dx'] = X, y, y, dx
    out['known/cfg'] = np.array([2.0, 3.0])
    out['names'] = np.array(names)
    save('upsample2d', **out)


def gen_simple_layers():
    out = {}
    r = rng(400)
    X = r.standard_normal((3, 4, 5, 6))
    X[0, 0, 0, :3] = 0.0          # >= 0 branch at exactly zero (layers.py:379,396)
    X[1, 1, 1, 0] = -0.0
    g = r.standard_normal(X.shape)
    for tag, layer in (('relu', Relu()), ('leaky', LeakyRelu(0.01)), ('leaky03', LeakyRelu(0.3)),
                       ('sigmoid', Sigmoid()), ('noop', Noop())):
        y = layer.forward(X)[0]
        dx = layer.backward(g)[0]
        out[f'{tag}/y'], out[f'{tag}/dx'] = y, dx
    out['X'], out['g'] = X, g
    Xs = r.standard_normal((4, 7)) * 8.0   # sigmoid over a wide range
    gs = r.standard_normal(Xs.shape)
    layer = Sigmoid()
    out['sigmoid_wide/X'], out['sigmoid_wide/g'] = Xs, gs
    out['sigmoid_wide/y'] = layer.forward(Xs)[0]
    out['sigmoid_wide/dx'] = layer.backward(gs)[0]

    # FullyConnected (layers.py:307-363): bias is the last row of w
    for tag, (m, nin, nout) in (('fc_small', (3, 2, 5)), ('fc_mid', (9, 33, 17)),
                                ('fc_char', (12, 129, 162))):
        w = r.standard_normal((nin + 1, nout)) * 0.2
        layer = FullyConnected(nin, nout, w=w.copy())
        Xf = r.standard_normal((m, nin))
        y = layer.forward(Xf)[0]
        gf = r.standard_normal(y.shape)
        layer.clear_grads()
        y = layer.forward(Xf)[0]
        dx = layer.backward(gf)[0]
        out[f'{tag}/X'], out[f'{tag}/w'], out[f'{tag}/y'] = Xf, w, y
        out[f'{tag}/g'], out[f'{tag}/dx'], out[f'{tag}/dw'] = gf, dx, layer.w.grad

    # Flatten (layers.py:287-304)
    layer = Flatten()
    y = layer.forward(X)[0]
    out['flatten/y'] = y
    out['flatten/dx'] = layer.backward(y * 2.0)[0]

    # Conv2DToBatchedFixedWidthed (convolutional.py:330-373)
    for tag, width, xs in (('fw3', 3, (3, 5, 5, 6)), ('fw8', 8, (2, 1, 11, 4)), ('fw2', 2, (2, 2, 2, 3))):
        Xw = r.standard_normal(xs)
        layer = Conv2DToBatchedFixedWidthed(width)
        y = layer.forward(Xw)[0]
        gw = r.standard_normal(y.shape)
        y = layer.forward(Xw)[0]
        dx = layer.backward(gw)[0]
        out[f'{tag}/X'], out[f'{tag}/y'], out[f'{tag}/g'], out[f'{tag}/dx'] = Xw, y, gw, dx
        out[f'{tag}/width'] = np.array(width)

    # Concat (layers.py:240-284) incl. the known answer of test_gradients.py:216-222
    a, b = np.array([[[1., 2., 3.]]]), np.array([[[4., 5., 6.]]])
    layer = Concat()
    y = layer.forward([a, b])[0]
    gs_ = layer.backward([y])
    out['concat_known/a'], out['concat_known/b'], out['concat_known/y'] = a, b, y
    out['concat_known/da'], out['concat_known/db'] = gs_[0], gs_[1]
    a, b, c = (r.standard_normal((2, 3, 4, n)) for n in (2, 5, 1))
    layer = Concat()
    y = layer.forward([a, b, c])[0]
    gc = r.standard_normal(y.shape)
    da, db, dc = layer.backward([gc])
    out['concat3/a'], out['concat3/b'], out['concat3/c'], out['concat3/y'] = a, b, c, y
    out['concat3/g'], out['concat3/da'], out['concat3/db'], out['concat3/dc'] = gc, da, db, dc
    save('layers', **out)


def gen_losses_reg_opt():
    out = {}
    r = rng(500)
    # segmentation losses (losses.py:9-42) on sigmoid-like predictions
    pred = 1 / (1 + np.exp(-r.standard_normal((3, 6, 7, 2))))
    gt = (r.random((3, 6, 7, 2)) > 0.6).astype(float)
    for tag, fn in (('dice', ref_losses.SegmentationDice2D()),
                    ('jaccard', ref_losses.SegmentationJaccard2D())):
        loss, grad = fn(pred, gt)
        out[f'{tag}/loss'], out[f'{tag}/grad'] = loss, grad
    out['seg/pred'], out['seg/gt'] = pred, gt
    # all-zero ground truth channel: exercises eps (losses.py:19-21)
    gt0 = gt.copy()
    gt0[..., 1] = 0
    loss, grad = ref_losses.SegmentationDice2D()(pred, gt0)
    out['dice_zero/gt'], out['dice_zero/loss'], out['dice_zero/grad'] = gt0, loss, grad

    logits = r.standard_normal((6, 9)) * 3
    onehot = np.zeros((6, 9))
    onehot[np.arange(6), r.integers(0, 9, 6)] = 1
    loss, grad = ref_losses.SoftmaxCrossEntropy()(logits, onehot)
    out['softmax_ce/pred'], out['softmax_ce/gt'] = logits, onehot
    out['softmax_ce/loss'], out['softmax_ce/grad'] = loss, grad
    big = r.standard_normal((5, 162)) * 6
    oh = np.zeros((5, 162))
    oh[np.arange(5), r.integers(0, 162, 5)] = 1
    loss, grad = ref_losses.SoftmaxCrossEntropy()(big, oh)
    out['softmax_ce162/pred'], out['softmax_ce162/gt'] = big, oh
    out['softmax_ce162/loss'], out['softmax_ce162/grad'] = loss, grad
    multi = (r.random((6, 9)) > 0.5).astype(float)
    loss, grad = ref_losses.SigmoidCrossEntropy()(logits, multi)
    out['sigmoid_ce/gt'], out['sigmoid_ce/loss'], out['sigmoid_ce/grad'] = multi, loss, grad

    # regularizers (regularizations.py:15-26)
    w = r.standard_normal((5, 4, 3))
    w[0, 0, 0] = 0.0
    for tag, fn in (('l1', ref_reg.L1(0.1)), ('l2', ref_reg.L2(0.01))):
        loss, grad = fn(w)
        out[f'{tag}/loss'], out[f'{tag}/grad'] = loss, grad
    out['reg/w'] = w
    save('losses_reg', **out)

    # optimizers (optimizers.py:31-98): three updates with fresh gradients
    out = {}

    class P:
        pass
    grads = [r.standard_normal((7, 5)) for _ in range(3)]
    w0 = r.standard_normal((7, 5))
    out['w0'] = w0
    for k, g in enumerate(grads):
        out[f'g{k}'] = g
    for tag, opt in (('adam', ref_opt.Adam(lr=0.0015)),
                     ('adam_b', ref_opt.Adam(lr=0.01, beta1=0.8, beta2=0.9)),
                     ('sgd', ref_opt.Momentum(lr=0.05, momentum=0)),
                     ('momentum', ref_opt.Momentum(lr=0.05, momentum=0.9)),
                     ('rmsprop', ref_opt.RMSProp(lr=0.01, rho=0.95))):
        p = P()
        p.value = w0.copy()
        opt.add_param(p)
        for k, g in enumerate(grads):
            p.grad = g.copy()
            opt.update(p)
            out[f'{tag}/w{k + 1}'] = p.value.copy()
    # Adagrad.update reads state.lr which is never set (optimizers.py:40): record the error type
    p = P()
    p.value = w0.copy()
    opt = ref_opt.Adagrad(lr=0.01)
    opt.add_param(p)
    p.grad = grads[0].copy()
    try:
        opt.update(p)
        err = 'none'
    except Exception as e:   # noqa: BLE001
        err = type(e).__name__
    out['adagrad/error'] = np.array(err)
    save('optimizers', **out)


# ----------------------------------------------------------------------------
# models
# ----------------------------------------------------------------------------
def analytic_weights(shape, salt):
    """Deterministic, RNG-free initial weights: the tests rebuild them from this formula."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.float64)
    fan_in = max(1, n // shape[-1])
    vals = np.sin(idx * 0.618 + salt * 1.37) * np.cos(idx * 0.0173 + salt) / np.sqrt(fan_in)
    return vals.reshape(shape)


def set_analytic_weights(model):
    weights = {}
    for salt, (lname, layer) in enumerate(sorted(model.layers.items())):
        ps = layer.params()
        if not ps:
            continue
        weights[lname] = {pn: analytic_weights(p.value.shape, salt + 0.5 * j).tolist()
                          for j, (pn, p) in enumerate(sorted(ps.items()))}
    model.set_weights(weights)


def input_grad(model, i):
    v = model.input_grads[i]
    return v[0] if isinstance(v, list) else v


def sample_param(name, arr, store):
    """Full tensor when small; strided sample + moments when large (Char dense layers)."""
    arr = np.asarray(arr)
    if arr.size <= 20000:
        store[name] = arr
    else:
        flat = arr.reshape(-1)
        store[name + '@stride97'] = flat[::97].copy()
        store[name + '@moments'] = np.array([flat.sum(), (flat ** 2).sum(), flat.min(), flat.max()])


def gen_my_model():
    from components.my_model import model as mm

    r = rng(600)
    specs = {
        'Monochrome': (mm.make_monochrome, (2, 16, 32, 1), 1),
        'Paragraph': (mm.make_paragraph, (2, 16, 32, 1), 1),
        'Line': (mm.make_line, (2, 16, 32, 1), 2),
        'Char': (mm.make_char, (2, 32, 12, 1), None),
    }
    for name, (maker, in_shape, out_ch) in specs.items():
        out = {}
        for opt_tag, make_opt in (('adam', lambda: ref_opt.Adam(lr=0.0015)),
                                  ('sgd', lambda: ref_opt.Momentum(lr=0.01, momentum=0))):
            np.random.seed(7)
            model = maker(in_shape, make_opt())
            set_analytic_weights(model)
            X = r.random(in_shape)
            if name == 'Char':
                n = in_shape[0] * in_shape[2]
                y = np.zeros((n, 162))
                y[np.arange(n), r.integers(0, 162, n)] = 1
            else:
                y = (r.random((*in_shape[:3], out_ch)) > 0.7).astype(float)
            out[f'{opt_tag}/X'], out[f'{opt_tag}/y'] = X, y
            pred = model.predict(X)[0]
            out[f'{opt_tag}/pred0'] = pred
            # gradients of the first step (before the optimizer touches them)
            losses = model.compute_loss_and_gradients(X, y)
            out[f'{opt_tag}/grad_loss'] = np.array(
                [*losses['output_losses'], losses['regularization_loss']])
            for pname, p in model.params().items():
                sample_param(f'{opt_tag}/grad/{pname}', p.grad, out)
            out[f'{opt_tag}/input_grad'] = input_grad(model, 0)
            model.clear_grads()
            step_losses = []
            for step in range(3):
                losses = model.train(X, y)
                step_losses.append([*losses['output_losses'], losses['regularization_loss']])
            out[f'{opt_tag}/step_losses'] = np.array(step_losses)
            for pname, p in model.params().items():
                sample_param(f'{opt_tag}/w3/{pname}', p.value, out)
            out[f'{opt_tag}/pred3'] = model.predict(X)[0]
            tl = model.test(X, y)
            out[f'{opt_tag}/test_loss3'] = np.array(tl['output_losses'])
        out['layer_names'] = np.array(sorted(model.layers.keys()))
        out['param_names'] = np.array(sorted(model.params().keys()))
        out['in_shape'] = np.array(in_shape)
        save(f'my_model_{name.lower()}', **out)


def gen_graph_models():
    """The composite models of test_gradients.py:191-308 (FCN, multi-IO DAG, nested)."""
    out = {}
    r = rng(700)
    # 9-layer FCN (test_gradients.py:191-214)
    X = r.random((4, 4, 8, 3)) * 0.98 + 0.01
    gt = r.integers(0, 2, size=(4, 11, 16, 5)).astype(float)

    def fcn_layers():
        np.random.seed(11)
        return [
            Convolutional2D((3, 3), 3, 2, padding=1),
            Convolutional2D((3, 3), 2, 3, padding=1),
            MaxPool2D(3),
            Convolutional2D((2, 2), 3, 4, padding=1),
            Upsample2D(5),
            Noop(),
            Relu(),
            Convolutional2D((2, 2), 4, 5, padding=1),
            Sigmoid(),
        ]
    for tag, loss in (('dice', ref_losses.SegmentationDice2D()),
                      ('jaccard', ref_losses.SegmentationJaccard2D())):
        model = Sequential(fcn_layers(), loss=loss)
        model.initialize_from_X(X)
        set_analytic_weights(model)
        losses = model.compute_loss_and_gradients(X, gt)
        out[f'fcn_{tag}/loss'] = np.array(losses['output_losses'])
        out[f'fcn_{tag}/pred'] = model.layers_outputs[0]
        out[f'fcn_{tag}/input_grad'] = input_grad(model, 0)
        for pname, p in model.params().items():
            out[f'fcn_{tag}/grad/{pname}'] = p.grad
        out[f'fcn_{tag}/param_names'] = np.array(sorted(model.params().keys()))
    out['fcn/X'], out['fcn/gt'] = X, gt

    # multi-input / multi-output DAG (test_gradients.py:225-259)
    np.random.seed(12)
    Xs = [r.standard_normal((5, 5, 5, 1)) for _ in range(3)]
    ys = [r.integers(0, 2, size=(5, 3)).astype(float) for _ in range(2)]
    layers = {
        'conv1': Convolutional2D((2, 2), out_channels=3),
        'conv2': Convolutional2D((2, 2), out_channels=3),
        'conv3': Convolutional2D((2, 2), out_channels=3),
        'concat': Concat(),
        'pool': MaxPool2D(2),
        'flatten': Flatten(),
        'dense1': FullyConnected(n_output=3),
        'dense2': FullyConnected(n_output=3),
    }
    relations = {
        'conv1': 0, 'conv2': 1, 'conv3': 2,
        'concat': ['conv1', 'conv2', 'conv3'],
        'pool': 'concat', 'flatten': 'pool', 'dense1': 'flatten', 'dense2': 'dense1',
        0: 'dense1', 1: 'dense2',
    }
    model = Model(layers, relations, loss=ref_losses.SigmoidCrossEntropy())
    model.initialize_from_X(Xs)
    set_analytic_weights(model)
    losses = model.compute_loss_and_gradients(Xs, ys)
    out['dag/loss'] = np.array(losses['output_losses'])
    for i in range(3):
        out[f'dag/X{i}'] = Xs[i]
        out[f'dag/input_grad{i}'] = input_grad(model, i)
    for i in range(2):
        out[f'dag/y{i}'] = ys[i]
        out[f'dag/pred{i}'] = model.layers_outputs[i]
    for pname, p in model.params().items():
        out[f'dag/grad/{pname}'] = p.grad
    out['dag/param_names'] = np.array(sorted(model.params().keys()))

    # nested model with L1/L2 (test_gradients.py:261-308)
    np.random.seed(13)
    Xn = [r.standard_normal((3, 18, 18, 3)), r.standard_normal((3, 18, 18, 3))]
    yn = r.integers(0, 2, size=(3, 1, 1, 3)).astype(float)

    def sub(out_ch):
        return Sequential([
            Convolutional2D((2, 2), out_channels=out_ch, regularizer=ref_reg.L2(0.1)),
            Convolutional2D((2, 2), out_channels=out_ch, regularizer=ref_reg.L1(0.1)),
            MaxPool2D((2, 2)),
        ])
    layers = {
        'row_1': sub(2), 'row_2': sub(3), 'concat_rows': Concat(),
        'concat_inputs': Concat(), 'row_inputs': sub(2), 'concat_all': Concat(),
        'pool_1': MaxPool2D((2, 2)), 'pool_2': MaxPool2D((2, 2)),
        'conv_end': Convolutional2D((2, 2), out_channels=3),
    }
    relations = {
        'row_1': 0, 'row_2': 1, 'concat_rows': ['row_1', 'row_2'],
        'concat_inputs': [0, 1], 'row_inputs': 'concat_inputs',
        'concat_all': ['concat_rows', 'row_inputs'],
        'pool_1': 'concat_all', 'pool_2': 'pool_1', 'conv_end': 'pool_2', 0: 'conv_end',
    }
    model = Model(layers, relations, loss=ref_losses.SegmentationDice2D())
    model.initialize_from_X(Xn)
    set_analytic_weights(model)
    losses = model.compute_loss_and_gradients(Xn, yn)
    out['nested/loss'] = np.array([*losses['output_losses'], losses['regularization_loss']])
    out['nested/X0'], out['nested/X1'], out['nested/y'] = Xn[0], Xn[1], yn
    out['nested/pred'] = model.layers_outputs[0]
    out['nested/input_grad0'] = input_grad(model, 0)
    out['nested/input_grad1'] = input_grad(model, 1)
    for pname, p in model.params().items():
        out[f'nested/grad/{pname}'] = p.grad
    out['nested/param_names'] = np.array(sorted(model.params().keys()))
    out['nested/layer_names'] = np.array(sorted(model.layers.keys()))
    shapes, _ = model.get_all_output_shapes([x.shape for x in Xn])
    out['nested/out_shape'] = np.array(shapes[0][0])
    save('graph_models', **out)


def gen_model_system():
    """ModelSystem / ModelComponent / IterableSelector (nn/model_system.py:76-167) over LISTS of differently sized
    crops, the way the reference feeds its Line and Char nets (my_model/model.py:353-400): every list entry is its
    own train step (weights move between the entries), the per-component losses are accumulated
    (`output_losses` lists concatenate, `regularization_loss` adds up, model_system.py:104-118), every prediction
    is appended to the selector's list."""
    from components.my_model import model as mm
    from components.nn.model_system import IterableSelector, ModelComponent, ModelSystem

    r = rng(900)
    np.random.seed(11)
    opt = ref_opt.Momentum(lr=0.01, momentum=0)
    line = mm.make_line((1, 32, 48, 1), opt)
    char = mm.make_char((1, 32, 16, 1), opt)
    set_analytic_weights(line)
    set_analytic_weights(char)
    crop_shapes = [(1, 32, 48, 1), (1, 48, 64, 1), (2, 64, 32, 1)]
    strip_shapes = [(1, 32, 16, 1), (2, 32, 24, 1), (1, 32, 40, 1)]
    out = {'crop_shapes': np.array(crop_shapes), 'strip_shapes': np.array(strip_shapes)}
    data = {'line_X': [], 'line_y': [], 'char_X': [], 'char_y': []}
    for i, shp in enumerate(crop_shapes):
        data['line_X'].append(r.random(shp))
        data['line_y'].append((r.random((*shp[:3], 2)) > 0.7).astype(float))
        out[f'line_X{i}'], out[f'line_y{i}'] = data['line_X'][-1], data['line_y'][-1]
    for i, shp in enumerate(strip_shapes):
        n = shp[0] * shp[2]
        y = np.zeros((n, 162))
        y[np.arange(n), r.integers(0, 162, n)] = 1
        data['char_X'].append(r.random(shp))
        data['char_y'].append(y)
        out[f'char_X{i}'], out[f'char_y{i}'] = data['char_X'][-1], y
    system = ModelSystem([
        ModelComponent('Line', line, IterableSelector('line_X', 'line_y', 'line_pred'), delist_result=True),
        ModelComponent('Char', char, IterableSelector('char_X', 'char_y', 'char_pred'), delist_result=True)])
    for mode in ('train1', 'train2', 'test', 'predict'):
        context = {k: list(v) for k, v in data.items()}
        getattr(system, mode.rstrip('12'))(context)
        if mode != 'predict':
            for name in ('Line', 'Char'):
                entry = context['losses'][name]
                out[f'{mode}/{name}/output_losses'] = np.array(entry['output_losses'])
                if 'regularization_loss' in entry:
                    out[f'{mode}/{name}/regularization_loss'] = np.array(entry['regularization_loss'])
        for key in ('line_pred', 'char_pred'):
            assert len(context[key]) == 3
            for i, pred in enumerate(context[key]):
                out[f'{mode}/{key}{i}'] = pred
    for model in (line, char):
        for pname, p in model.params().items():
            sample_param(f'final/{pname}', p.value, out)
    save('model_system_lists', **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'model_system':      # (adds the one fixture, leaves the others alone)
        return gen_model_system()
    gen_model_system()
    gen_conv()
    gen_pool_upsample()
    gen_simple_layers()
    gen_losses_reg_opt()
    gen_my_model()
    gen_graph_models()


if __name__ == '__main__':
    main()
