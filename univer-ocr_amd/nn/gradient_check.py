"""Numeric gradient checker (reference: nn/gradient_check.py:8-181, same function names).

Two-point formula per element, delta=1e-5, isclose with rtol=tol.  The point being perturbed lives
on the host; every evaluation copies it to the device and runs the HIP kernels, so this exercises
exactly the product path.  Needs CP.set_dtype('float64') to be meaningful (float32 cannot resolve
a 1e-5 step)."""
import numpy as np

from .gpu import CP


def _host(a):
    return np.array(CP.asnumpy(a[0] if isinstance(a, list) else a), dtype=np.float64)


def check_gradient(f, x, delta=1e-5, tol=1e-4):
    """f(host array) -> (loss, analytic gradient).  Returns True when every element matches."""
    x = _host(x)
    fx, analytic = f(x)
    analytic = _host(analytic)
    assert analytic.shape == x.shape, f'{analytic.shape} != {x.shape}'
    for ix in np.ndindex(*x.shape):
        step = np.zeros_like(x)
        step[ix] = delta
        numeric = (float(f(x + step)[0]) - float(f(x - step)[0])) / (2 * delta)
        if not np.isclose(numeric, analytic[ix], tol):
            print(f'Gradients are different at {ix}.\n  Analytic: {analytic[ix]},\n  Numeric: {numeric},\n'
                  f'  diff: {abs(analytic[ix] - numeric)},\n  ratio: {analytic[ix] / numeric if numeric else "inf"}')
            return False
    return True


def _weighted_sum(output, weight):
    return float(np.sum(_host(output) * weight))


def check_layer_gradient(layer, x, delta=1e-5, tol=1e-4):
    """Input gradient of a layer: loss = sum(forward(x) * random weights)."""
    x = _host(x)
    out = layer.forward(CP.copy(x))
    weight = np.random.randn(*_host(out).shape)
    d_out = CP.copy(weight)

    def helper(point):
        out = layer.forward(CP.copy(point))
        loss = _weighted_sum(out, weight)
        grad = layer.backward(d_out)
        return loss, grad
    return check_gradient(helper, x, delta, tol)


def check_layer_param_gradient(layer, x, param_name, delta=1e-5, tol=1e-4):
    param = layer.params()[param_name]
    initial = _host(param.value)
    x_dev = CP.copy(_host(x))
    layer.clear_grads()
    out = layer.forward(x_dev)
    weight = np.random.randn(*_host(out).shape)
    d_out = CP.copy(weight)

    def helper(w):
        param.value = w
        layer.clear_grads()
        out = layer.forward(x_dev)
        loss = _weighted_sum(out, weight)
        layer.backward(d_out)
        return loss, param.grad
    ok = check_gradient(helper, initial, delta, tol)
    param.value = initial
    return ok


def check_model_gradient(model, X, y, delta=1e-5, tol=1e-4, check_inputs=False):
    X = [_host(x) for x in (X if isinstance(X, list) else [X])]
    y_dev = [CP.copy(_host(t)) for t in y] if isinstance(y, list) else CP.copy(_host(y))

    def total(losses):
        return float(np.sum([float(v) for v in losses['output_losses']])) + float(losses['regularization_loss'])

    if check_inputs:
        for key in range(len(X)):
            print(f'Checking gradient for model input #{key}')

            def helper(offset, key=key):
                this_X = [CP.copy(x + offset) if i == key else CP.copy(x) for i, x in enumerate(X)]
                loss = total(model.compute_loss_and_gradients(this_X, y_dev))
                return loss, model.input_grads[key]
            if not check_gradient(helper, np.zeros_like(X[key]), delta, tol):
                return False

    X_dev = [CP.copy(x) for x in X]
    for name, param in model.params().items():
        print(f'Checking gradient for {name}')
        initial = _host(param.value)

        def helper(w, param=param):
            param.value = w
            loss = total(model.compute_loss_and_gradients(X_dev, y_dev))
            return loss, param.grad
        ok = check_gradient(helper, initial, delta, tol)
        param.value = initial
        if not ok:
            return False
    return True
