"""Upsample2D (reference: nn/layers/upsample.py:10-135): nearest-neighbour repeat, backward = sum
over each (sy, sx) block."""
from .. import ops
from ..help_func import make_list_if_not, tuplize
from .layers import BaseLayerGPU


class Upsample2D(BaseLayerGPU):
    def __init__(self, scale_factor, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.scale_factor = tuplize('scale_factor', scale_factor, 2)

    def _forward(self, X, mem_id=0):
        self._mem[mem_id] = X.shape
        return ops.upsample2d_fwd(X, self.scale_factor)

    def _backward(self, grad, mem_id=0):
        return ops.upsample2d_bwd(grad, self._mem[mem_id], self.scale_factor)

    def get_output_shapes(self, input_shapes):
        batch, height, width, channels = make_list_if_not(input_shapes)[0]
        return [(batch, height * self.scale_factor[0], width * self.scale_factor[1], channels)]

    def changes_receptive_field(self):
        return True

    def _get_receptive_field(self, axis, position, output_id):
        assert 0 <= axis < 2, f'Upsample2D has two axis, found {axis}'
        assert output_id < self.get_outputs_count(), f'This layer has only {self.get_outputs_count()} outputs'
        key = (axis, position, output_id)
        if key not in self._receptive_fields:
            self._receptive_fields[key] = {0: {position // self.scale_factor[axis]}}
        return self._receptive_fields[key]
