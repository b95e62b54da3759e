"""MaxPool2D (reference: nn/layers/maxpool.py:9-239).  Semantics = the NumPy path (:24-90): zero
padding takes part in the max, ties share the gradient equally, ceil_mode windows that overrun the
padded input shrink.  The mask is kept window-major on the device as uint8."""
from .. import ops
from ..help_func import make_list_if_not, tuplize
from .layers import BaseLayerGPU


class MaxPool2D(BaseLayerGPU):
    def __init__(self, kernel_size, padding=0, stride=None, ceil_mode=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.kernel_size = tuplize('kernel_size', kernel_size, 2)
        self.padding = tuplize('padding', padding, 2)
        self.stride = self.kernel_size if stride is None else tuplize('stride', stride, 2)
        self.ceil_mode = ceil_mode

    def _forward(self, X, mem_id=0):
        y, mask = ops.maxpool2d_fwd(X, self.kernel_size, self.stride, self.padding, self.ceil_mode)
        self._mem[mem_id] = (mask, X.shape)
        return y

    def _backward(self, grad, mem_id=0):
        mask, input_shape = self._mem[mem_id]
        return ops.maxpool2d_bwd(grad, mask, input_shape, self.kernel_size, self.stride, self.padding)

    def get_output_shapes(self, input_shapes):
        batch, height, width, channels = make_list_if_not(input_shapes)[0]
        oh, ow = ops.pool_out_hw(height, width, self.kernel_size, self.stride, self.padding, self.ceil_mode)
        return [(batch, oh, ow, channels)]

    def changes_receptive_field(self):
        return True

    def _get_receptive_field(self, axis, position, output_id):
        assert 0 <= axis < 2, f'MaxPool2D has two axis, found {axis}'
        assert output_id < self.get_outputs_count(), f'This layer has only {self.get_outputs_count()} outputs'
        key = (axis, position, output_id)
        if key not in self._receptive_fields:
            start = position * self.stride[axis] - self.padding[axis]
            self._receptive_fields[key] = {0: set(range(start, start + self.kernel_size[axis]))}
        return self._receptive_fields[key]
