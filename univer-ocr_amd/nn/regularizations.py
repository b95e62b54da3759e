"""L1 / L2 regularizers (reference: nn/regularizations.py:4-26): __call__(weights) -> (loss, grad).

`apply(param, slot)` is the fused form used by BaseLayer.regularize: one kernel adds the gradient
into param.grad and the loss into a device slot (the reference does a float() sync per parameter).
"""
from . import ops
from .gpu import CP


class BaseRegularizer:
    kind = None

    def __init__(self, reg_strength):
        self.reg_strength = float(reg_strength)

    def __call__(self, weights):
        weights = ops.as_device(weights)
        grad = CP.zeros(weights.shape, weights.dtype)
        loss = ops.regularize(self.kind, weights, grad, self.reg_strength)
        return loss, grad

    def apply(self, value, grad, slot=None, accumulate=False):
        return ops.regularize(self.kind, value, grad, self.reg_strength, slot, accumulate)

    def __repr__(self):
        return f'{type(self).__name__}({self.reg_strength})'


class L1(BaseRegularizer):
    kind = 'l1'


class L2(BaseRegularizer):
    kind = 'l2'
