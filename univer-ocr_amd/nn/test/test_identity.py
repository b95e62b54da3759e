"""`python test_nn.py test_identity True` -- the reference's CPU<->GPU identity test
(nn/test/test_identity.py:76-134: 5 conv + 4 pool + 1 upsample configs on randn(5,240,320,6),
y and dX compared with np.isclose) restated for this backend: the float64 GENERIC kernels stand
in for the reference's NumPy side and the float32 production kernels (specialised / MFMA paths)
are compared against them -- plus dw/db, which the reference never compared.  rtol 1e-4, atol 1e-5
on float32 (np.isclose defaults are float64-sized)."""
from datetime import datetime as dt

import numpy as np

from .. import ops
from ..gpu import CP
from ..layers import Convolutional2D, MaxPool2D, Upsample2D


def make_layers():
    ks, cin, cout = (3, 3), 6, 7
    convs = {'Convolutional 2D Layer': {}, 'Convolutional 2D Layer with Padding': dict(padding=1),
             'Convolutional 2D Layer with non-zero Padding': dict(padding=1, padding_value=0.5),
             'Convolutional 2D Layer with Stride': dict(stride=2),
             'Convolutional 2D Layer with Padding and Stride': dict(padding=1, stride=2)}
    pools = {'Max Pooling 2D Layer': {}, 'Max Pooling 2D Layer with Padding': dict(padding=1),
             'Max Pooling 2D Layer with 1-Stride': dict(stride=1),
             'Max Pooling 2D Layer with Padding and 1-Stride': dict(padding=1, stride=1)}
    layers = {name: Convolutional2D(ks, cin, cout, **kw) for name, kw in convs.items()}
    layers.update({name: MaxPool2D((2, 2), **kw) for name, kw in pools.items()})
    layers['Upsampling 2D Layer'] = Upsample2D((2, 2))
    return layers


def run(layers, inputs, dtype, generic):
    CP.set_dtype(dtype)
    rt = CP.runtime()
    rt.set_option('fast_paths', 0 if generic else 1)
    rt.set_option('mfma', 0 if generic else 1)
    out = {}
    for name, layer in layers.items():
        X, grad = inputs[name]
        for p in layer.params().values():
            p._value = CP.copy(p.value, dtype)
            p._grad = CP.zeros(p.value.shape, dtype)
        t0 = dt.now()
        y = layer.forward(CP.copy(X))[0]
        dX = layer.backward(CP.copy(grad))[0]
        rt.synchronize()
        print(f'  {name}... done in {dt.now() - t0}')
        grads = {pn: CP.asnumpy(p.grad) for pn, p in layer.params().items()}
        out[name] = (CP.asnumpy(y), CP.asnumpy(dX), grads)
    rt.set_option('fast_paths', 1)
    rt.set_option('mfma', 1)
    return out


def main(use_gpu=True, *args, **kwargs):
    if not use_gpu:
        CP.use_cpu()
    CP.use_gpu()
    previous = CP.dtype
    np.random.seed(0)
    layers = make_layers()
    inputs = {}
    for name, layer in layers.items():
        X = np.random.randn(5, 240, 320, 6)
        shape = layer.get_output_shapes(X.shape)[0]
        inputs[name] = (X, np.random.randn(*shape))
    try:
        print('Running float64 generic kernels')
        ref = run(layers, inputs, 'float64', generic=True)
        print('\nRunning float32 production kernels')
        got = run(layers, inputs, 'float32', generic=False)
    finally:
        CP.set_dtype(previous)
    print('\nComparing results')
    correct = 0
    for name in layers:
        ok = all(np.isclose(a, b, rtol=1e-4, atol=1e-5).all() for a, b in zip(got[name][:2], ref[name][:2]))
        ok = ok and all(np.isclose(got[name][2][k], ref[name][2][k], rtol=1e-4, atol=1e-3).all() for k in ref[name][2])
        print(f'  {name}: {ok}')
        correct += bool(ok)
    print(f'\nCorrect: {correct}/{len(layers)}')
    return correct, len(layers)
