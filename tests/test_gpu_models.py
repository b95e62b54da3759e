"""GPU parity of whole models driven through the framework API (Model / Sequential / my_model
builders -> layer classes -> C ABI -> HIP kernels) against the reference's outputs in tests/golden:
forward, loss, input gradient, every parameter gradient, three optimizer steps, post-step weights.

Tolerance (normalised max error): float64 1e-10 (three chained steps); float32 1e-5 for single
passes (2e-5 for gradients), 5e-5 for weights after three SGD steps and 5e-4 after three Adam
steps.  Adam without bias correction (optimizers.py:56-61) moves every weight by ~lr*3.2*sign(g)
on the first step, so a gradient element whose float32 value is rounding noise (|g| ~ 1e-7 of the
layer's largest) gets a different step than in float64; measured worst cases on MI355X are
1.3e-4 (Char conv_1/w), 3.4e-5 (Paragraph), <3e-6 elsewhere, SGD <= 1.4e-6, float64 <= 5e-14
(printed with -s)."""
import numpy as np
import pytest

from conftest import load_golden, rel_linf
from oracle.nn_oracle import analytic_weights

pytestmark = pytest.mark.gpu

PASS_TOL = {'float32': 1e-5, 'float64': 1e-11}
STEP_TOL = {'float32': 5e-5, 'float64': 1e-10}
ADAM_F32_WEIGHT_TOL = 5e-4


@pytest.fixture(params=['float32', 'float64'])
def dt(request):
    from univer_ocr_amd.nn import CP
    CP.set_dtype(request.param)
    yield request.param
    CP.set_dtype('float32')


def set_analytic_weights(model):
    weights = {}
    for salt, lname in enumerate(sorted(model.layers)):
        ps = model.layers[lname].params()
        if ps:
            weights[lname] = {pn: analytic_weights(p.value.shape, salt + 0.5 * j).tolist()
                              for j, (pn, p) in enumerate(sorted(ps.items()))}
    model.set_weights(weights)


def close(a, b, tol, what=''):
    from univer_ocr_amd.nn import CP
    err = rel_linf(CP.asnumpy(a), b)
    assert err <= tol, f'{what}: rel_linf={err:.3e} > {tol:.1e}'
    return err


def check_sampled(name, arr, g, prefix, tol):
    from univer_ocr_amd.nn import CP
    key = f'{prefix}/{name}'
    arr = CP.asnumpy(arr)
    if key in g.files:
        return close(arr, g[key], tol, key)
    return close(arr.reshape(-1)[::97], g[key + '@stride97'], tol, key)


def losses_row(losses):
    return np.array([float(v) for v in losses['output_losses']] + [float(losses['regularization_loss'])])


@pytest.mark.parametrize('net_name', ['Monochrome', 'Paragraph', 'Line', 'Char'])
@pytest.mark.parametrize('opt_tag,fused', [('adam', False), ('sgd', False), ('adam', True)])
def test_my_model_net(net_name, opt_tag, fused, dt):
    """fused=True: the production graph (Model.enable_fusion: pair / upsample+conv / windows kernels, folded
    activations, fused optimizer tail) with Adam, the reference trainer's optimizer, against the same golden
    vectors (the SGD leg of the fused graph: test_fused_activation_graph_matches_reference)."""
    from univer_ocr_amd.my_model.model import NET_MAKERS
    from univer_ocr_amd.nn import CP
    from univer_ocr_amd.nn.optimizers import Adam, Momentum
    g = load_golden(f'my_model_{net_name.lower()}')
    opt = Adam(lr=0.0015) if opt_tag == 'adam' else Momentum(lr=0.01, momentum=0)
    model = NET_MAKERS[net_name](tuple(int(v) for v in g['in_shape']), opt)
    assert sorted(model.layers) == [str(s) for s in g['layer_names']]
    assert sorted(model.params()) == [str(s) for s in g['param_names']]
    if fused:
        model.enable_fusion()
    set_analytic_weights(model)
    X, y = CP.copy(g[f'{opt_tag}/X']), CP.copy(g[f'{opt_tag}/y'])
    close(model.predict(X)[0], g[f'{opt_tag}/pred0'], PASS_TOL[dt], 'pred0')
    losses = model.compute_loss_and_gradients(X, y)
    close(losses_row(losses), g[f'{opt_tag}/grad_loss'], PASS_TOL[dt], 'loss')
    close(model.input_grads[0], g[f'{opt_tag}/input_grad'], PASS_TOL[dt] * 2, 'input_grad')
    for pn, p in model.params().items():
        check_sampled(pn, p.grad, g, f'{opt_tag}/grad', PASS_TOL[dt] * 2)
    model.clear_grads()
    rows = [losses_row(model.train(X, y)) for _ in range(3)]
    adam32 = opt_tag == 'adam' and dt == 'float32'
    close(np.array(rows), g[f'{opt_tag}/step_losses'], ADAM_F32_WEIGHT_TOL if adam32 else STEP_TOL[dt], 'step_losses')
    if adam32:
        # Adam normalises the step: an element whose float32 gradient is rounding noise moves by
        # +-lr*3.2 per step in a direction float64 may not share (the MFMA path sums dw in float32).
        # Typical weights must agree to float32 precision (median), the noisy minority stays small
        # (90th percentile) and nothing can exceed the distance between two Adam trajectories.
        worst, wtol = 0.0, ADAM_F32_WEIGHT_TOL
        for pn, p in model.params().items():
            key = f'{opt_tag}/w3/{pn}'
            got = CP.asnumpy(p.value).astype(np.float64)
            ref = g[key] if key in g.files else g[key + '@stride97']
            got = got if key in g.files else got.reshape(-1)[::97]
            diff = np.abs(got - ref).reshape(-1)
            scale = np.max(np.abs(ref))
            q50, q90 = np.quantile(diff, [0.5, 0.9])
            assert q50 <= (1e-5 if diff.size >= 64 else ADAM_F32_WEIGHT_TOL) * scale, f'{pn}: median error {q50 / scale:.2e}'
            assert q90 <= ADAM_F32_WEIGHT_TOL * scale, f'{pn}: 90th percentile error {q90 / scale:.2e}'
            assert diff.max() <= 2 * 3 * 0.0015 * 3.2, f'{pn}: beyond any Adam trajectory'
            worst = max(worst, diff.max() / scale)
    else:
        wtol = STEP_TOL[dt]
        worst = max(check_sampled(pn, p.value, g, f'{opt_tag}/w3', wtol) for pn, p in model.params().items())
    close(model.predict(X)[0], g[f'{opt_tag}/pred3'], wtol * (20 if opt_tag == 'adam' and dt == 'float32' else 1), 'pred3')
    test_losses = model.test(X, y)
    close(np.array([float(v) for v in test_losses['output_losses']]), g[f'{opt_tag}/test_loss3'],
          ADAM_F32_WEIGHT_TOL if adam32 else STEP_TOL[dt])
    assert not model.nan_weights()
    print(f'{net_name}/{opt_tag}/{dt}: worst post-step weight error {worst:.2e}')


@pytest.mark.parametrize('loss_tag', ['dice', 'jaccard'])
def test_fcn_sequential(loss_tag, dt):
    """test_gradients.py:191-214: conv conv pool3 conv upsample5 noop relu conv sigmoid."""
    from univer_ocr_amd.nn import CP
    from univer_ocr_amd.nn.layers import Convolutional2D, MaxPool2D, Noop, Relu, Sigmoid, Upsample2D
    from univer_ocr_amd.nn.losses import SegmentationDice2D, SegmentationJaccard2D
    from univer_ocr_amd.nn.models import Sequential
    g = load_golden('graph_models')
    layers = [Convolutional2D((3, 3), 3, 2, padding=1), Convolutional2D((3, 3), 2, 3, padding=1), MaxPool2D(3),
              Convolutional2D((2, 2), 3, 4, padding=1), Upsample2D(5), Noop(), Relu(),
              Convolutional2D((2, 2), 4, 5, padding=1), Sigmoid()]
    loss = SegmentationDice2D() if loss_tag == 'dice' else SegmentationJaccard2D()
    model = Sequential(layers, loss=loss)
    X, gt = CP.copy(g['fcn/X']), CP.copy(g['fcn/gt'])
    model.initialize_from_X(X)
    set_analytic_weights(model)
    losses = model.compute_loss_and_gradients(X, gt)
    tol = PASS_TOL[dt] * 2
    close(model.layers_outputs[0], g[f'fcn_{loss_tag}/pred'], tol, 'pred')
    close(np.array([float(v) for v in losses['output_losses']]), g[f'fcn_{loss_tag}/loss'], tol, 'loss')
    close(model.input_grads[0], g[f'fcn_{loss_tag}/input_grad'], tol, 'input_grad')
    assert sorted(model.params()) == [str(s) for s in g[f'fcn_{loss_tag}/param_names']]
    for pn, p in model.params().items():
        close(p.grad, g[f'fcn_{loss_tag}/grad/{pn}'], tol, pn)


def test_multi_io_dag(dt):
    """test_gradients.py:225-259: three conv branches -> Concat -> MaxPool -> Flatten -> two dense heads."""
    from univer_ocr_amd.nn import CP
    from univer_ocr_amd.nn.layers import Concat, Convolutional2D, Flatten, FullyConnected, MaxPool2D
    from univer_ocr_amd.nn.losses import SigmoidCrossEntropy
    from univer_ocr_amd.nn.models import Model
    g = load_golden('graph_models')
    layers = {'conv1': Convolutional2D((2, 2), out_channels=3), 'conv2': Convolutional2D((2, 2), out_channels=3),
              'conv3': Convolutional2D((2, 2), out_channels=3), 'concat': Concat(), 'pool': MaxPool2D(2),
              'flatten': Flatten(), 'dense1': FullyConnected(n_output=3), 'dense2': FullyConnected(n_output=3)}
    relations = {'conv1': 0, 'conv2': 1, 'conv3': 2, 'concat': ['conv1', 'conv2', 'conv3'], 'pool': 'concat',
                 'flatten': 'pool', 'dense1': 'flatten', 'dense2': 'dense1', 0: 'dense1', 1: 'dense2'}
    model = Model(layers, relations, loss=SigmoidCrossEntropy())
    Xs = [CP.copy(g[f'dag/X{i}']) for i in range(3)]
    ys = [CP.copy(g[f'dag/y{i}']) for i in range(2)]
    model.initialize_from_X(Xs)
    set_analytic_weights(model)
    losses = model.compute_loss_and_gradients(Xs, ys)
    tol = PASS_TOL[dt] * 2
    close(np.array([float(v) for v in losses['output_losses']]), g['dag/loss'], tol, 'loss')
    for i in range(2):
        close(model.layers_outputs[i], g[f'dag/pred{i}'], tol, f'pred{i}')
    for i in range(3):
        close(model.input_grads[i], g[f'dag/input_grad{i}'], tol, f'input_grad{i}')
    for pn, p in model.params().items():
        close(p.grad, g[f'dag/grad/{pn}'], tol, pn)


def test_nested_model_with_l1_l2(dt):
    """test_gradients.py:261-308: nested Sequential sub-models, Concat of model inputs, L1/L2, Dice."""
    from univer_ocr_amd.nn import CP
    from univer_ocr_amd.nn.layers import Concat, Convolutional2D, MaxPool2D
    from univer_ocr_amd.nn.losses import SegmentationDice2D
    from univer_ocr_amd.nn.models import Model, Sequential
    from univer_ocr_amd.nn.regularizations import L1, L2
    g = load_golden('graph_models')

    def sub(out_ch):
        return Sequential([Convolutional2D((2, 2), out_channels=out_ch, regularizer=L2(0.1)),
                           Convolutional2D((2, 2), out_channels=out_ch, regularizer=L1(0.1)), MaxPool2D((2, 2))])
    layers = {'row_1': sub(2), 'row_2': sub(3), 'concat_rows': Concat(), 'concat_inputs': Concat(),
              'row_inputs': sub(2), 'concat_all': Concat(), 'pool_1': MaxPool2D((2, 2)),
              'pool_2': MaxPool2D((2, 2)), 'conv_end': Convolutional2D((2, 2), out_channels=3)}
    relations = {'row_1': 0, 'row_2': 1, 'concat_rows': ['row_1', 'row_2'], 'concat_inputs': [0, 1],
                 'row_inputs': 'concat_inputs', 'concat_all': ['concat_rows', 'row_inputs'],
                 'pool_1': 'concat_all', 'pool_2': 'pool_1', 'conv_end': 'pool_2', 0: 'conv_end'}
    model = Model(layers, relations, loss=SegmentationDice2D())
    Xs = [CP.copy(g['nested/X0']), CP.copy(g['nested/X1'])]
    model.initialize_from_X(Xs)
    assert sorted(model.layers) == [str(s) for s in g['nested/layer_names']]
    set_analytic_weights(model)
    losses = model.compute_loss_and_gradients(Xs, CP.copy(g['nested/y']))
    tol = PASS_TOL[dt] * 2
    close(losses_row(losses), g['nested/loss'], tol, 'loss')
    close(model.layers_outputs[0], g['nested/pred'], tol, 'pred')
    close(model.input_grads[0], g['nested/input_grad0'], tol)
    close(model.input_grads[1], g['nested/input_grad1'], tol)
    for pn, p in model.params().items():
        close(p.grad, g[f'nested/grad/{pn}'], tol, pn)


@pytest.mark.parametrize('net_name', ['Monochrome', 'Paragraph', 'Line', 'Char'])
def test_fused_activation_graph_matches_reference(net_name, dt):
    """Model.enable_fusion(): conv + LeakyReLU / Sigmoid as one kernel, activation gradient taken from
    the output -- must reproduce the reference's losses, gradients and post-step weights."""
    from univer_ocr_amd.my_model.model import NET_MAKERS
    from univer_ocr_amd.nn import CP
    from univer_ocr_amd.nn.optimizers import Momentum
    g = load_golden(f'my_model_{net_name.lower()}')
    model = NET_MAKERS[net_name](tuple(int(v) for v in g['in_shape']), Momentum(lr=0.01, momentum=0))
    model.enable_fusion()
    fused_conv, fused_act = model._fusion_maps()
    n_convs = sum(1 for name in model.layers if name.rsplit('/', 1)[-1].startswith('conv_'))
    n_dense = 2 if net_name == 'Char' else 0             # dense_1, dense_2 + LeakyRelu (dense_3 feeds the loss)
    assert len(fused_conv) == n_convs + n_dense and len(fused_act) == n_convs + n_dense   # every conv is followed by one
    set_analytic_weights(model)
    X, y = CP.copy(g['sgd/X']), CP.copy(g['sgd/y'])
    close(model.predict(X)[0], g['sgd/pred0'], PASS_TOL[dt], 'pred0')
    # the multi-layer kernels exist in float32 only: Monochrome = one conv pair (csrc/conv_pair.hip), Line =
    # Line / Paragraph =
    # two upsample+conv blocks on the low-res tensor (csrc/conv_up.hip); float64 runs layer by layer
    f32 = np.dtype(dt) == np.float32
    assert len(model._pairs_used) == (1 if f32 and net_name == 'Monochrome' else 0)
    assert len(model._ups_used) == (2 if f32 and net_name in ('Line', 'Paragraph') else 0)
    # Char: windows + flatten + dense_1 as one implicit GEMM on the conv feature map (ops.windows_dense_fwd)
    assert len(model._wins_used) == (1 if f32 and net_name == 'Char' else 0)
    losses = model.compute_loss_and_gradients(X, y)
    close(losses_row(losses), g['sgd/grad_loss'], PASS_TOL[dt], 'loss')
    close(model.input_grads[0], g['sgd/input_grad'], PASS_TOL[dt] * 2, 'input_grad')
    for pn, p in model.params().items():
        check_sampled(pn, p.grad, g, 'sgd/grad', PASS_TOL[dt] * 2)
    model.clear_grads()
    rows = [losses_row(model.train(X, y)) for _ in range(3)]
    close(np.array(rows), g['sgd/step_losses'], STEP_TOL[dt], 'step_losses')
    for pn, p in model.params().items():
        check_sampled(pn, p.value, g, 'sgd/w3', STEP_TOL[dt])


def test_skip_input_grads_leaves_losses_and_updates_unchanged():
    """Model.skip_input_grads drops only the dX of the first convs: losses and weights after two SGD
    steps are bit-identical to the default run (which computes dX like models.py:226-230)."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    layers = make_page_batch(2, 32, 64, 16, seed=5)
    results = []
    for input_grads in (True, False):
        trainer = PageTrainer(2, 32, 64, 16, optimizer='sgd', lr=0.01, seed=3, input_grads=input_grads)
        context = trainer.make_context(layers)
        for _ in range(2):
            losses = trainer.step(context)
        weights = {}
        for model in trainer.models.values():
            weights.update(model.get_weights())
        results.append(({n: [float(v) for v in l['output_losses']] for n, l in losses.items()}, weights))
        if not input_grads:
            assert all(m.input_grads == {} for m in trainer.models.values())
    assert results[0][0] == results[1][0]
    for name, w in results[0][1].items():
        assert np.array_equal(np.asarray(w), np.asarray(results[1][1][name])), name


@pytest.mark.parametrize('graphs', [False, True], ids=['eager', 'graphs'])
def test_weight_gradients_on_side_streams_change_nothing(graphs):
    """PageTrainer(side_wgrad=...): the weight-gradient kernels of a net run on a side stream of its lane
    (Runtime.side) and rejoin before anything reads a parameter gradient -- the same kernels on the same data, so
    losses and weights after six steps are bit-identical to the single-stream backward pass."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    batches = [make_page_batch(2, 32, 64, 16, seed=s) for s in (7, 8)]
    results = []
    lazy = CP.lazy_losses
    CP.lazy_losses = True
    try:
        for side in ((), ('all',)):
            trainer = PageTrainer(2, 32, 64, 16, optimizer='adam', lr=0.001, seed=3, graphs=graphs, side_wgrad=side,
                                  group_wgrad=())
            assert all(m.side_wgrad == bool(side) for m in trainer.models.values())
            history = []
            for i in range(6):
                losses = trainer.step(trainer.make_context(batches[i % 2]))
                history.append({n: [float(v) for v in l['output_losses']] + [float(l['regularization_loss'])]
                                for n, l in losses.items()})
            weights = {}
            for model in trainer.models.values():
                weights.update(model.get_weights())
            results.append((history, weights))
            if side:
                assert len(CP.runtime()._sides) >= 3           # one side stream per lane was really used
    finally:
        CP.lazy_losses = lazy
    assert results[0][0] == results[1][0]
    for name, w in results[0][1].items():
        assert np.array_equal(np.asarray(w), np.asarray(results[1][1][name])), name


@pytest.mark.parametrize('graphs', [False, True], ids=['eager', 'graphs'])
@pytest.mark.parametrize('grouped', [('Char',), ('all',)], ids=['char', 'all'])
def test_grouped_weight_gradients_train_like_separate_launches(graphs, grouped):
    """PageTrainer(group_wgrad=...): the backward pass inside Runtime.defer_wgrad -- the Char net's five weight-gradient
    GEMMs as one launch, and with 'all' every net's finish kernels as one launch per net -- against separate launches:
    same losses and weights to float32 / float64 summation order over four steps; ungrouped nets bit-identical."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    batches = [make_page_batch(4, 32, 64, 64, seed=s) for s in (7, 8)]
    results = []
    lazy = CP.lazy_losses
    CP.lazy_losses = True
    try:
        for group in ((), grouped):
            trainer = PageTrainer(4, 32, 64, 64, optimizer='sgd', lr=0.01, seed=3, graphs=graphs, group_wgrad=group)
            assert trainer.models['Char'].group_wgrad == bool(group)
            history = []
            for i in range(4):
                losses = trainer.step(trainer.make_context(batches[i % 2]))
                history.append({n: [float(v) for v in l['output_losses']] + [float(l['regularization_loss'])]
                                for n, l in losses.items()})
            results.append((history, {n: m.get_weights() for n, m in trainer.models.items()}))
    finally:
        CP.lazy_losses = lazy
    (h0, w0), (h1, w1) = results
    for a, b in zip(h0, h1):
        for name in a:
            if name == 'Char' or 'all' in grouped:
                assert np.allclose(a[name], b[name], rtol=1e-5, atol=1e-7), (a[name], b[name])
            else:
                assert a[name] == b[name]
    for name in w0:
        for layer, params in w0[name].items():
            for key, w in params.items():
                a, b = np.asarray(w, np.float64), np.asarray(w1[name][layer][key], np.float64)
                if name == 'Char' or 'all' in grouped:
                    assert np.allclose(a, b, rtol=1e-4, atol=1e-6), (layer, key)
                else:
                    assert np.array_equal(a, b), (layer, key)


def test_graph_replay_matches_eager_steps():
    """PageTrainer(graphs=True): per-net HIP graphs replayed == the eager multi-stream step, bit for bit
    (same kernels, same order per stream), including a change of the input batch after capture."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    batches = [make_page_batch(2, 32, 64, 16, seed=s) for s in (5, 6)]
    results = []
    lazy = CP.lazy_losses
    CP.lazy_losses = True
    try:
        for graphs in (False, True):
            trainer = PageTrainer(2, 32, 64, 16, optimizer='adam', lr=0.001, seed=3, graphs=graphs)
            eager_maker = PageTrainer.make_context.__get__(PageTrainer.__new__(PageTrainer))
            eager_maker.__self__.graphs = False
            history = []
            for i in range(8):                     # steps 0-1 eager, capture at step 2, replay after
                if i < 5:                          # caller-owned arrays: copied into the graphs' statics
                    context = eager_maker(batches[i % 2])
                    keep = context['monochrome_X']
                else:                              # trainer-owned input buffers, refilled in place
                    context = trainer.make_context(batches[i % 2])
                losses = trainer.step(context)
                if i < 5:
                    assert np.array_equal(CP.asnumpy(keep), batches[i % 2]['image'].astype(np.float32))
                history.append({n: [float(v) for v in l['output_losses']] + [float(l['regularization_loss'])]
                                for n, l in losses.items()})
            assert (trainer._captured is not None) == graphs
            weights = {}
            for model in trainer.models.values():
                weights.update(model.get_weights())
            pred = CP.asnumpy(context['monochrome_pred'])
            results.append((history, weights, pred))
    finally:
        CP.lazy_losses = lazy
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][2], results[1][2])
    for name, w in results[0][1].items():
        assert np.array_equal(np.asarray(w), np.asarray(results[1][1][name])), name


@pytest.mark.parametrize('graphs', [False, True])
def test_pipelined_lanes_match_joined_steps(graphs):
    """PageTrainer(pipelined=True): no per-step join of the four streams (a lane starts its next step when
    ITS previous one is done).  Losses of every step (read later through DeviceScalar.ready), the published
    prediction after join() and the final weights are bit-identical to the joined step, also when every step
    gets a fresh batch whose arrays the caller drops right away."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    batches = [make_page_batch(2, 32, 64, 16, seed=s) for s in (11, 12, 13)]
    lazy = CP.lazy_losses
    CP.lazy_losses = True
    results = []
    try:
        for pipelined in (False, True):
            trainer = PageTrainer(2, 32, 64, 16, optimizer='sgd', lr=0.01, seed=4, graphs=graphs, pipelined=pipelined)
            kept = []
            for i in range(7):
                context = trainer.make_context(batches[i % 3])
                kept.append(trainer.step(context))        # lazy losses, read only at the end
                last = context
            trainer.join()
            pred = CP.asnumpy(last['line_pred'])
            # every step's losses are read only now: with graphs they are per-step snapshots of the static slots
            history = [{n: [float(v) for v in l['output_losses']] + [float(l['regularization_loss'])]
                        for n, l in losses.items()} for losses in kept]
            weights = {}
            for model in trainer.models.values():
                weights.update(model.get_weights())
            results.append((history, pred, weights))
    finally:
        CP.lazy_losses = lazy
    assert results[0][0] == results[1][0]
    # the snapshots are distinct values per step (fresh batches, moving weights), not seven aliases of the last one
    assert len({tuple(step['Line']) for step in results[0][0]}) == 7
    assert np.array_equal(results[0][1], results[1][1])
    for name, w in results[0][2].items():
        assert np.array_equal(np.asarray(w), np.asarray(results[1][2][name])), name


def test_get_weights_single_transfer_equals_per_parameter_lists():
    """Model.get_weights(): one D2H copy of the flat ParamPack, sliced on the host == layer.get_weights()
    (one tolist() per parameter, layers.py:120-121); JSON layout unchanged."""
    from univer_ocr_amd.my_model.model import NET_MAKERS
    from univer_ocr_amd.nn import CP
    from univer_ocr_amd.nn.optimizers import Momentum
    CP.use_gpu(0)
    CP.set_dtype('float32')
    model = NET_MAKERS['Line']((1, 32, 64, 1), Momentum(lr=0.01, momentum=0))
    assert model.pack is not None
    packed = model.get_weights()
    per_layer = {name: layer.get_weights() for name, layer in model.layers.items() if layer.params()}
    assert packed == per_layer and 'Line/end/conv_1' in packed
    assert np.array(packed['Line/end/conv_1']['w']).shape == (5, 5, 4, 2)


@pytest.mark.parametrize('optimizer', ['sgd', 'adam'])
def test_graph_replay_follows_learning_rate_changes(optimizer):
    """The reference's trainer changes optimizer.lr every epoch (my_model/trainer.py:260).  The fused optimizer
    kernels read lr / momentum / betas from a device array (uocr_*_step_fused, hyper_dev), so a step captured in a
    HIP graph BEFORE the change must train exactly like eager steps issued after it."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    lazy, CP.lazy_losses = CP.lazy_losses, True
    try:
        layers = make_page_batch(2, 32, 64, 16, seed=3)
        weights = {}
        for graphs in (False, True):
            trainer = PageTrainer(2, 32, 64, 16, optimizer=optimizer, lr=0.01, seed=4, graphs=graphs, pipelined=True)
            context = trainer.make_context(layers)
            for step in range(8):
                if step == 4:
                    trainer.join()
                    assert (trainer._captured is not None) == graphs       # captured with lr = 0.01 ...
                    trainer.optimizer.lr *= 0.25                          # ... and now it changes
                if step == 6:
                    trainer.optimizer.lr = 0.02
                trainer.step(context)
            trainer.join()
            CP.runtime().synchronize()
            weights[graphs] = {n: p.value.numpy() for m in trainer.models.values() for n, p in m.params().items()}
        changed = 0
        for name, ref in weights[False].items():
            assert np.array_equal(ref, weights[True][name]), name
            changed += 1
        assert changed > 10
    finally:
        CP.lazy_losses = lazy


@pytest.mark.parametrize('fused', [False, True])
def test_model_system_lists_of_differently_sized_crops(fused, dt):
    """ModelSystem.train / test / predict with IterableSelector over LISTS of crops of different sizes -- the way
    the reference feeds its Line and Char nets (nn/model_system.py:76-167, my_model/model.py:353-400) -- against the
    reference's own run (golden model_system_lists.npz): per-entry train steps, accumulated losses
    (`output_losses` concatenated, `regularization_loss` summed), one prediction per entry, final weights.
    Every entry has another shape, so every kernel is launched with several geometries on one model."""
    from univer_ocr_amd.my_model.model import make_char, make_line
    from univer_ocr_amd.nn import CP
    from univer_ocr_amd.nn.model_system import IterableSelector, ModelComponent, ModelSystem
    from univer_ocr_amd.nn.optimizers import Momentum
    g = load_golden('model_system_lists')
    opt = Momentum(lr=0.01, momentum=0)
    line = make_line(tuple(int(v) for v in g['crop_shapes'][0]), opt)
    char = make_char(tuple(int(v) for v in g['strip_shapes'][0]), opt)
    for model in (line, char):
        if fused:
            model.enable_fusion()
        set_analytic_weights(model)
    system = ModelSystem([
        ModelComponent('Line', line, IterableSelector('line_X', 'line_y', 'line_pred'), delist_result=True),
        ModelComponent('Char', char, IterableSelector('char_X', 'char_y', 'char_pred'), delist_result=True)])
    counts = {'line': len(g['crop_shapes']), 'char': len(g['strip_shapes'])}

    def fresh_context():
        return {f'{tag}_{k}': [CP.copy(g[f'{tag}_{k}{i}']) for i in range(n)] for tag, n in counts.items() for k in 'Xy'}
    tol = STEP_TOL[dt]
    for mode in ('train1', 'train2', 'test', 'predict'):
        context = fresh_context()
        getattr(system, mode.rstrip('12'))(context)
        if mode != 'predict':
            for name in ('Line', 'Char'):
                entry = context['losses'][name]
                assert len(entry['output_losses']) == 3
                close(np.array([float(v) for v in entry['output_losses']]), g[f'{mode}/{name}/output_losses'], tol)
                if mode != 'test':
                    close(np.array(float(entry['regularization_loss'])), g[f'{mode}/{name}/regularization_loss'], tol)
        else:
            assert set(context['prediction']) == {'Line', 'Char'}
        for tag, n in counts.items():
            assert len(context[f'{tag}_pred']) == n
            for i in range(n):
                close(context[f'{tag}_pred'][i], g[f'{mode}/{tag}_pred{i}'], tol, f'{mode}/{tag}_pred{i}')
    for model in (line, char):
        for pn, p in model.params().items():
            check_sampled(pn, p.value, g, 'final', tol)


@pytest.mark.parametrize('graphs', [False, True])
def test_page_trainer_forward_lanes_and_graphs(graphs):
    """PageTrainer.forward (configs[1]: forward only): every net on its lane, optionally replayed from HIP graphs --
    the predictions equal Model.predict on the main stream, also after the inputs changed."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    lazy, CP.lazy_losses = CP.lazy_losses, True
    try:
        trainer = PageTrainer(2, 32, 64, 16, optimizer='sgd', lr=0.01, seed=4, graphs=graphs)
        for seed in (1, 2):
            layers = make_page_batch(2, 32, 64, 16, seed=seed)
            context = trainer.make_context(layers)
            out = trainer.forward(dict(context))
            CP.runtime().synchronize()
            import torch
            torch.cuda.synchronize()
            for comp in trainer.model_system.components:
                ref = comp.model.predict(CP.copy(layers[{'Monochrome': 'image', 'Char': 'char_lines'}.get(comp.name, 'monochrome')]))[0]
                assert np.array_equal(CP.asnumpy(out[comp.selector.pred_label]), CP.asnumpy(ref)), comp.name
    finally:
        CP.lazy_losses = lazy
