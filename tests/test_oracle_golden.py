"""Pins oracle/nn_oracle.py against the reference's own outputs (tests/golden/*.npz, produced by
tests/golden/make_golden.py importing the reference NumPy path).  float64, tolerance 1e-12
(normalised max error) -- the only difference is summation order."""
import numpy as np
import pytest

from conftest import load_golden, rel_linf
from oracle import nn_oracle as O

TOL = 1e-12


def close(a, b, tol=TOL):
    err = rel_linf(a, b)
    assert err <= tol, f'rel_linf={err:.3e} > {tol}'


def conv_names():
    return [str(n) for n in load_golden('conv2d')['names']]


@pytest.mark.parametrize('tag', conv_names())
def test_conv2d(tag):
    g = load_golden('conv2d')
    kh, kw, sh, sw, ph, pw, pv, bias = g[f'{tag}/cfg']
    kw_ = dict(stride=(int(sh), int(sw)), padding=(int(ph), int(pw)), padding_value=pv,
               bias=bool(bias))
    X, w, b = g[f'{tag}/X'], g[f'{tag}/w'], g[f'{tag}/b']
    assert w.shape[:2] == (int(kh), int(kw))
    close(O.conv2d_fwd(X, w, b, **kw_), g[f'{tag}/y'])
    dx, dw, db = O.conv2d_bwd(X, w, g[f'{tag}/g'], **kw_)
    close(dx, g[f'{tag}/dx'])
    close(dw, g[f'{tag}/dw'])
    if bias:
        close(db, g[f'{tag}/db'])
    else:
        assert np.all(db == 0) and np.all(g[f'{tag}/db'] == 0)


@pytest.mark.parametrize('tag', conv_names())
def test_conv2d_loop_formulation(tag):
    """The per-output-pixel loops of the reference's NumPy path (convolutional.py:90-96, 121-134), restated in
    oracle.conv2d_fwd_loop / conv2d_bwd_loop for bench.py's `reference-cost-model` baseline, against the
    reference's outputs (skipped for the one large identity-test shape: 76 800 iterations per call)."""
    g = load_golden('conv2d')
    X = g[f'{tag}/X']
    if X.shape[1] * X.shape[2] > 64 * 64:
        pytest.skip('large shape: covered by the vectorised restatement')
    kh, kw, sh, sw, ph, pw, pv, bias = g[f'{tag}/cfg']
    kw_ = dict(stride=(int(sh), int(sw)), padding=(int(ph), int(pw)), padding_value=pv, bias=bool(bias))
    w, b = g[f'{tag}/w'], g[f'{tag}/b']
    close(O.conv2d_fwd_loop(X, w, b, **kw_), g[f'{tag}/y'])
    dx, dw, db = O.conv2d_bwd_loop(X, w, g[f'{tag}/g'], **kw_)
    close(dx, g[f'{tag}/dx'])
    close(dw, g[f'{tag}/dw'])
    if bias:
        close(db, g[f'{tag}/db'])
    else:
        assert np.all(db == 0)


@pytest.mark.parametrize('tag', [str(n) for n in load_golden('maxpool2d')['names']])
def test_maxpool2d(tag):
    g = load_golden('maxpool2d')
    kh, kw, sh, sw, ph, pw, ceil = (int(v) for v in g[f'{tag}/cfg'])
    X = g[f'{tag}/X']
    y, mask = O.maxpool2d_fwd(X, (kh, kw), (sh, sw), (ph, pw), bool(ceil))
    assert np.array_equal(y, g[f'{tag}/y'])
    dx = O.maxpool2d_bwd(g[f'{tag}/g'], mask, X.shape, (kh, kw), (sh, sw), (ph, pw))
    close(dx, g[f'{tag}/dx'])


def test_maxpool_known_answer():
    """test_gradients.py:171-177."""
    X = np.array([[1, 0, 1, 2], [0, -1, -1, -1], [-1, -1, 1, -2]], dtype=float).reshape(1, 3, 4, 1)
    y, _ = O.maxpool2d_fwd(X, 2, ceil_mode=True)
    assert np.array_equal(y[0, :, :, 0], [[1, 2], [-1, 1]])


@pytest.mark.parametrize('tag', [str(n) for n in load_golden('upsample2d')['names']])
def test_upsample2d(tag):
    g = load_golden('upsample2d')
    sf = tuple(int(v) for v in g[f'{tag}/cfg'])
    assert np.array_equal(O.upsample2d_fwd(g[f'{tag}/X'], sf), g[f'{tag}/y'])
    close(O.upsample2d_bwd(g[f'{tag}/g'], sf), g[f'{tag}/dx'])


def test_upsample_known_answer():
    """test_gradients.py:181-188: bwd(fwd(x)) of [[.1,.2],[.3,.4]] with scale (2,3)."""
    X = np.array([[0.1, 0.2], [0.3, 0.4]]).reshape(1, 2, 2, 1)
    y = O.upsample2d_fwd(X, (2, 3))
    close(O.upsample2d_bwd(y, (2, 3))[0, :, :, 0], [[0.6, 1.2], [1.8, 2.4]])


def test_activations():
    g = load_golden('layers')
    X, gr = g['X'], g['g']
    assert np.array_equal(O.relu_fwd(X), g['relu/y'])
    assert np.array_equal(O.relu_bwd(X, gr), g['relu/dx'])
    assert np.array_equal(O.leaky_relu_fwd(X, 0.01), g['leaky/y'])
    assert np.array_equal(O.leaky_relu_bwd(X, gr, 0.01), g['leaky/dx'])
    assert np.array_equal(O.leaky_relu_fwd(X, 0.3), g['leaky03/y'])
    close(O.sigmoid_fwd(X), g['sigmoid/y'])
    close(O.sigmoid_bwd(X, gr), g['sigmoid/dx'])
    close(O.sigmoid_fwd(g['sigmoid_wide/X']), g['sigmoid_wide/y'])
    close(O.sigmoid_bwd(g['sigmoid_wide/X'], g['sigmoid_wide/g']), g['sigmoid_wide/dx'])


@pytest.mark.parametrize('tag', ['fc_small', 'fc_mid', 'fc_char'])
def test_dense(tag):
    g = load_golden('layers')
    X, w = g[f'{tag}/X'], g[f'{tag}/w']
    close(O.dense_fwd(X, w), g[f'{tag}/y'])
    dx, dw = O.dense_bwd(X, w, g[f'{tag}/g'])
    close(dx, g[f'{tag}/dx'])
    close(dw, g[f'{tag}/dw'])


@pytest.mark.parametrize('tag', ['fw3', 'fw8', 'fw2'])
def test_fixed_width(tag):
    g = load_golden('layers')
    X, width = g[f'{tag}/X'], int(g[f'{tag}/width'])
    assert np.array_equal(O.fixed_width_fwd(X, width), g[f'{tag}/y'])
    close(O.fixed_width_bwd(g[f'{tag}/g'], X.shape, width), g[f'{tag}/dx'])


def test_concat_flatten():
    g = load_golden('layers')
    y = O.concat_fwd([g['concat_known/a'], g['concat_known/b']])
    assert np.array_equal(y, g['concat_known/y'])          # test_gradients.py:216-222
    parts = [g['concat3/a'], g['concat3/b'], g['concat3/c']]
    assert np.array_equal(O.concat_fwd(parts), g['concat3/y'])
    da, db, dc = O.concat_bwd(g['concat3/g'], [p.shape for p in parts])
    assert np.array_equal(da, g['concat3/da'])
    assert np.array_equal(db, g['concat3/db'])
    assert np.array_equal(dc, g['concat3/dc'])
    assert np.array_equal(g['X'].reshape(3, -1), g['flatten/y'])


def test_losses_and_regularizers():
    g = load_golden('losses_reg')
    pred, gt = g['seg/pred'], g['seg/gt']
    for tag, fn, gtt in (('dice', O.dice_loss, gt), ('jaccard', O.jaccard_loss, gt),
                         ('dice_zero', O.dice_loss, g['dice_zero/gt'])):
        loss, grad = fn(pred, gtt)
        close(loss, g[f'{tag}/loss'])
        close(grad, g[f'{tag}/grad'])
    for tag in ('softmax_ce', 'softmax_ce162'):
        loss, grad = O.softmax_ce_loss(g[f'{tag}/pred'], g[f'{tag}/gt'])
        close(loss, g[f'{tag}/loss'])
        close(grad, g[f'{tag}/grad'])
    loss, grad = O.sigmoid_ce_loss(g['softmax_ce/pred'], g['sigmoid_ce/gt'])
    close(loss, g['sigmoid_ce/loss'])
    close(grad, g['sigmoid_ce/grad'])
    w = g['reg/w']
    loss, grad = O.l1_reg(w, 0.1)
    close(loss, g['l1/loss'])
    assert np.array_equal(grad, g['l1/grad'])
    loss, grad = O.l2_reg(w, 0.01)
    close(loss, g['l2/loss'])
    close(grad, g['l2/grad'])


def test_optimizers():
    g = load_golden('optimizers')
    grads = [g[f'g{k}'] for k in range(3)]
    for tag, opt in (('adam', O.AdamState(lr=0.0015)),
                     ('adam_b', O.AdamState(lr=0.01, beta1=0.8, beta2=0.9)),
                     ('sgd', O.MomentumState(lr=0.05, momentum=0)),
                     ('momentum', O.MomentumState(lr=0.05, momentum=0.9)),
                     ('rmsprop', O.RMSPropState(lr=0.01, rho=0.95))):
        w = g['w0'].copy()
        for k, gr in enumerate(grads):
            w = opt.update('p', w, gr)
            close(w, g[f'{tag}/w{k + 1}'])
    assert str(g['adagrad/error']) == 'AttributeError'     # optimizers.py:40 reads an unset attr


def check_sampled(name, arr, g, prefix, tol=TOL):
    key = f'{prefix}/{name}'
    if key in g.files:
        close(arr, g[key], tol)
    else:
        flat = np.asarray(arr).reshape(-1)
        close(flat[::97], g[key + '@stride97'], tol)
        m = g[key + '@moments']
        close(np.array([flat.sum(), (flat ** 2).sum(), flat.min(), flat.max()]), m, 1e-10)


@pytest.mark.parametrize('net_name', ['Monochrome', 'Paragraph', 'Line', 'Char'])
@pytest.mark.parametrize('opt_tag', ['adam', 'sgd'])
def test_my_model_nets(net_name, opt_tag):
    """Forward, first-step gradients, three train steps and the post-step weights of the four
    my_model nets (model.py:108-304) against the reference's Model.train (models.py:232-254)."""
    g = load_golden(f'my_model_{net_name.lower()}')
    net = O.make_net(net_name)
    assert net.param_names() == [str(s) for s in g['param_names']]
    X, y = g[f'{opt_tag}/X'], g[f'{opt_tag}/y']
    close(net.forward(X), g[f'{opt_tag}/pred0'])
    losses, _, dx = net.loss_and_grads(X, y)
    close(np.array([*losses['output_losses'], losses['regularization_loss']]),
          g[f'{opt_tag}/grad_loss'])
    close(dx, g[f'{opt_tag}/input_grad'], 1e-11)
    for pn in net.param_names():
        check_sampled(pn, net.grads[pn], g, f'{opt_tag}/grad', 1e-11)
    opt = O.AdamState(lr=0.0015) if opt_tag == 'adam' else O.MomentumState(lr=0.01, momentum=0)
    rows = []
    for _ in range(3):
        losses, _ = net.train_step(X, y, opt)
        rows.append([*losses['output_losses'], losses['regularization_loss']])
    close(np.array(rows), g[f'{opt_tag}/step_losses'], 1e-10)
    for pn in net.param_names():
        check_sampled(pn, net.params[pn], g, f'{opt_tag}/w3', 1e-10)
    close(net.forward(X), g[f'{opt_tag}/pred3'], 1e-10)
    tl, _ = net.test(X, y)
    close(np.array(tl['output_losses']), g[f'{opt_tag}/test_loss3'], 1e-10)


def test_fcn_chain_with_pool_and_upsample():
    """The 9-layer FCN of test_gradients.py:191-214 (conv, conv, pool3, conv, upsample5, noop,
    relu, conv, sigmoid) with Dice and Jaccard, run as an oracle chain."""
    g = load_golden('graph_models')
    X, gt = g['fcn/X'], g['fcn/gt']
    names = ['0_Convolutional2D', '1_Convolutional2D', '2_MaxPool2D', '3_Convolutional2D',
             '4_Upsample2D', '5_Noop', '6_Relu', '7_Convolutional2D', '8_Sigmoid']
    cfgs = [dict(ks=(3, 3), cin=3, cout=2, stride=1, padding=1),
            dict(ks=(3, 3), cin=2, cout=3, stride=1, padding=1),
            dict(ks=(3, 3)),
            dict(ks=(2, 2), cin=3, cout=4, stride=1, padding=1),
            dict(scale=5), {}, {},
            dict(ks=(2, 2), cin=4, cout=5, stride=1, padding=1), {}]
    kinds = ['conv', 'conv', 'maxpool', 'conv', 'upsample', 'noop', 'relu', 'conv', 'sigmoid']
    spec = list(zip(names, kinds, cfgs))
    for tag, loss in (('dice', 'dice'), ('jaccard', 'jaccard')):
        net = O.Net(spec, loss)
        net.params = O.analytic_net_weights(spec)
        losses, pred, dx = net.loss_and_grads(X, gt)
        close(pred, g[f'fcn_{tag}/pred'])
        close(np.array(losses['output_losses']), g[f'fcn_{tag}/loss'])
        close(dx, g[f'fcn_{tag}/input_grad'], 1e-11)
        for pn in net.param_names():
            close(net.grads[pn], g[f'fcn_{tag}/grad/{pn}'], 1e-11)


def test_model_system_list_path_oracle():
    """nn/model_system.py:104-118 over lists of differently sized crops (golden model_system_lists.npz, made by the
    reference's ModelSystem + IterableSelector): the oracle's chains, stepped entry by entry, reproduce the
    accumulated losses, every prediction and the final weights."""
    g = load_golden('model_system_lists')
    nets = {'Line': O.make_net('Line'), 'Char': O.make_net('Char')}
    opts = {n: O.MomentumState(0.01, 0.0) for n in nets}
    feeds = {'Line': ('line', len(g['crop_shapes'])), 'Char': ('char', len(g['strip_shapes']))}
    for mode in ('train1', 'train2', 'test'):
        for name, (tag, count) in feeds.items():
            out_losses, reg = [], 0.0
            for i in range(count):
                X, y = g[f'{tag}_X{i}'], g[f'{tag}_y{i}']
                if mode == 'test':
                    losses, pred = nets[name].test(X, y)
                else:
                    losses, pred = nets[name].train_step(X, y, opts[name])
                    reg += losses['regularization_loss']
                out_losses += losses['output_losses']
                close(pred, g[f'{mode}/{tag}_pred{i}'], 1e-10)
            close(np.array(out_losses), g[f'{mode}/{name}/output_losses'], 1e-10)
            if mode != 'test':
                close(np.array(reg), g[f'{mode}/{name}/regularization_loss'], 1e-10)
    for name, net in nets.items():
        for pn, value in net.params.items():
            key = f'final/{pn}'
            if key in g.files:
                close(value, g[key], 1e-10)
            else:
                close(value.reshape(-1)[::97], g[key + '@stride97'], 1e-10)
