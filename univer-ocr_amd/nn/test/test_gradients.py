"""`python test_nn.py test_gradients True` -- numeric-vs-analytic gradient checks of every layer,
loss, regularizer and of composite models THROUGH THE HIP KERNELS (reference script:
nn/test/test_gradients.py:60-310, same shapes and tolerances: delta 1e-5, rtol 1e-4).

Runs in float64 (CP.set_dtype): a 1e-5 finite-difference step is below float32 resolution.  Prints
`Passed`/`Error` per check and `Correct: n/N`; returns (correct, total)."""
import datetime
from itertools import cycle

import numpy as np

from .. import gradient_check as gc
from ..gpu import CP
from ..layers import (Concat, Conv2DToBatchedFixedWidthed, Convolutional2D, Flatten, FullyConnected, LeakyRelu,
                      MaxPool2D, Noop, Relu, Sigmoid, Upsample2D)
from ..losses import SegmentationDice2D, SegmentationJaccard2D, SigmoidCrossEntropy, SoftmaxCrossEntropy
from ..models import Model, Sequential
from ..regularizations import L1, L2


class Score:
    def __init__(self):
        self.correct = self.total = 0
        self.time = datetime.timedelta(0)

    def run(self, label, fn, *args, **kwargs):
        print(label)
        t0 = datetime.datetime.now()
        ok = bool(fn(*args, **kwargs))
        dt = datetime.datetime.now() - t0
        print(f'{"Passed" if ok else "Error"} {dt}\n')
        self.correct += ok
        self.total += 1
        self.time += dt


def main(use_gpu=True):
    if not use_gpu:
        CP.use_cpu()
    CP.use_gpu()
    previous = CP.dtype
    CP.set_dtype('float64')
    print('Using GPU (HIP kernels, float64)')
    np.random.seed(0)
    score = Score()
    try:
        _run_all(score)
    finally:
        CP.set_dtype(previous)
    print(f'Correct: {score.correct}/{score.total}\nTotal time: {score.time}')
    return score.correct, score.total


def _run_all(score):
    batch, n_in, n_out = 3, 2, 5
    X_fc, X_fl = np.random.randn(batch, n_in), np.random.randn(batch, n_in, n_out)
    score.run('Fully Connected Layer', gc.check_layer_gradient, FullyConnected(n_in, n_out), X_fc)
    score.run('Fully Connected Layer - Param w', gc.check_layer_param_gradient, FullyConnected(n_in, n_out), X_fc, 'w')
    score.run('Flatten Layer', gc.check_layer_gradient, Flatten(), X_fl)
    score.run('ReLU Layer', gc.check_layer_gradient, Relu(), X_fl)
    score.run('Leaky ReLU Layer', gc.check_layer_gradient, LeakyRelu(), X_fl)
    score.run('L1 Regularization', gc.check_gradient, L1(0.1), X_fc)
    score.run('L2 Regularization', gc.check_gradient, L2(0.1), X_fc)
    score.run('Sigmoid Activation Function Layer', gc.check_layer_gradient, Sigmoid(), X_fc)

    sizes = [4, 7, 5, 3]
    X = np.random.randn(batch, sizes[0])
    regs = [reg(float(np.random.rand())) for _, reg in zip(range(3), cycle([L1, L2]))]

    def dense_stack(with_reg):
        return [FullyConnected(sizes[i], sizes[i + 1], regularizer=regs[i] if with_reg else None) for i in range(3)]
    y = np.zeros((batch, sizes[-1]))
    y[np.arange(batch), np.random.randint(sizes[-1], size=batch)] = 1
    for label, with_reg in (('Sequential model with Softmax CE Loss', False),
                            ('Sequential model with Softmax CE Loss and regularization', True)):
        model = Sequential(dense_stack(with_reg), loss=SoftmaxCrossEntropy())
        model.initialize_from_X(CP.copy(X))
        score.run(label, gc.check_model_gradient, model, X, y, check_inputs=True)
    y = np.random.choice([0.0, 1.0], size=(batch, sizes[-1]))
    for label, with_reg in (('Sequential model with Sigmoid CE Loss', False),
                            ('Sequential model with Sigmoid CE Loss and regularization', True)):
        model = Sequential(dense_stack(with_reg), loss=SigmoidCrossEntropy())
        model.initialize_from_X(CP.copy(X))
        score.run(label, gc.check_model_gradient, model, X, y, check_inputs=True)

    X_conv = np.random.randn(batch, 5, 5, 6)
    for label, kw in (('Convolutional 2D Layer', {}), ('Convolutional 2D Layer with Padding', dict(padding=1)),
                      ('Convolutional 2D Layer with non-zero Padding', dict(padding=1, padding_value=0.5)),
                      ('Convolutional 2D Layer with Stride', dict(stride=2)),
                      ('Convolutional 2D Layer with Padding and Stride', dict(padding=1, stride=2))):
        layer = Convolutional2D((4, 4), 6, 7, **kw)
        score.run(label, gc.check_layer_gradient, layer, X_conv)
        score.run(f'{label} - Param w', gc.check_layer_param_gradient, layer, X_conv, 'w')
        score.run(f'{label} - Param b', gc.check_layer_param_gradient, layer, X_conv, 'b')
    score.run('Conv2D to Batched Fixed Widthed', gc.check_layer_gradient, Conv2DToBatchedFixedWidthed(3), X_conv)

    known = np.array([[1, 0, 1, 2], [0, -1, -1, -1], [-1, -1, 1, -2]], dtype=float).reshape(1, 3, 4, 1)
    pooled = CP.asnumpy(MaxPool2D(2, ceil_mode=True).forward(CP.copy(known))[0])[0, :, :, 0]
    print('Max Pooling 2D Layer (ceil_mode known answer)\n', pooled)
    assert np.array_equal(pooled, [[1, 2], [-1, 1]])
    score.run('Max Pooling 2D Layer', gc.check_layer_gradient, MaxPool2D(2), X_conv)

    X_up = np.array([[0.1, 0.2], [0.3, 0.4]]).reshape(1, 2, 2, 1).repeat(4, axis=0).repeat(3, axis=-1)
    up = Upsample2D((2, 3))
    res = up.forward(CP.copy(X_up))[0]
    back = CP.asnumpy(up.backward(res)[0])[0, :, :, 0]
    print('Upsampling 2D Layer (known answer)\n', back)
    assert np.allclose(back, [[0.6, 1.2], [1.8, 2.4]])
    score.run('Upsampling 2D Layer', gc.check_layer_gradient, Upsample2D(5), X_up)

    X_dice = np.random.rand(4, 4, 8, 3) * 0.98 + 0.01
    gt = np.random.randint(0, 2, size=(4, 11, 16, 5)).astype(float)

    def fcn():
        return [Convolutional2D((3, 3), 3, 2, padding=1), Convolutional2D((3, 3), 2, 3, padding=1), MaxPool2D(3),
                Convolutional2D((2, 2), 3, 4, padding=1), Upsample2D(5), Noop(), Relu(),
                Convolutional2D((2, 2), 4, 5, padding=1), Sigmoid()]
    for label, loss in (('Sequential FCN with Segmentation Dice 2D Loss', SegmentationDice2D()),
                        ('Sequential FCN with Segmentation Jaccard 2D Loss', SegmentationJaccard2D())):
        model = Sequential(fcn(), loss=loss)
        model.initialize_from_X(CP.copy(X_dice))
        score.run(label, gc.check_model_gradient, model, X_dice, gt, check_inputs=True)

    concat = Concat()
    parts = [np.array([[[1., 2., 3.]]]), np.array([[[4., 5., 6.]]])]
    joined = concat.forward([CP.copy(p) for p in parts])
    print('Concat Layer\n', CP.asnumpy(joined[0]), [CP.asnumpy(g) for g in concat.backward(joined)])
    score.run('Concat Layer', gc.check_layer_gradient, Concat(), parts[0])

    Xs = [np.random.randn(5, 5, 5, 1) for _ in range(3)]
    ys = [np.random.randint(2, size=(5, 3)).astype(float) for _ in range(2)]
    layers = {'conv1': Convolutional2D((2, 2), out_channels=3), 'conv2': Convolutional2D((2, 2), out_channels=3),
              'conv3': Convolutional2D((2, 2), out_channels=3), 'concat': Concat(), 'pool': MaxPool2D(2),
              'flatten': Flatten(), 'dense1': FullyConnected(n_output=3), 'dense2': FullyConnected(n_output=3)}
    relations = {'conv1': 0, 'conv2': 1, 'conv3': 2, 'concat': ['conv1', 'conv2', 'conv3'], 'pool': 'concat',
                 'flatten': 'pool', 'dense1': 'flatten', 'dense2': 'dense1', 0: 'dense1', 1: 'dense2'}
    model = Model(layers, relations, loss=SigmoidCrossEntropy())
    model.initialize_from_X([CP.copy(x) for x in Xs])
    print(model.get_all_output_shapes([x.shape for x in Xs])[0])
    score.run(f'Small non-sequential model with multiple inputs and outputs: {model.count_parameters()} parameters',
              gc.check_model_gradient, model, Xs, ys, check_inputs=True)

    Xn = [np.random.randn(3, 18, 18, 3), np.random.randn(3, 18, 18, 3)]
    yn = np.random.randint(2, size=(3, 1, 1, 3)).astype(float)

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
    model.initialize_from_X([CP.copy(x) for x in Xn])
    score.run(f'Big non-sequential model: {model.count_parameters()} parameters',
              gc.check_model_gradient, model, Xn, yn)
