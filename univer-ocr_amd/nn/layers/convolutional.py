"""Convolutional2D and Conv2DToBatchedFixedWidthed (reference: nn/layers/convolutional.py:12-373).

The reference has a NumPy path (Python loop over output pixels, :62-145) and a numba.cuda path
(:147-288).  Here both directions are single calls into libuniver_hip.so (uocr_conv2d_*).  The
semantics follow the NumPy path: the `bias` flag is honoured in forward and in db (the numba path
always adds the bias, :168-169), `padding_value` fills the border and contributes to dw.
"""
import numpy as np

from .. import ops
from ..gpu import CP
from ..help_func import make_list_if_not, tuplize
from ..progress_tracker import track_method
from .layers import BaseLayer, BaseLayerGPU, Param


class Convolutional2D(BaseLayerGPU):
    def __init__(self, kernel_size, in_channels=None, out_channels=None, padding=0, padding_value=0,
                 stride=1, w=None, b=None, bias=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.kernel_size = tuplize('kernel_size', kernel_size, 2)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.padding = tuplize('padding', padding, 2)
        self.padding_value = padding_value
        self.stride = tuplize('stride', stride, 2)
        self.w, self.b, self.bias = w, b, bias
        self.fused_activation = None          # set by Model when the next layer is fused in
        if self.input_shapes is None and in_channels is not None:
            self.input_shapes = [(None, None, None, in_channels)]
        if self.input_shapes is not None:
            self.initialize(self.input_shapes)
        else:
            self.is_initialized = False

    def initialize(self, input_shapes):
        """convolutional.py:33-55: w and b come from ONE (kh*kw*cin + 1, cout) draw."""
        self.input_shapes = input_shapes
        self.in_channels = input_shapes[0][3]
        if self.out_channels is None:
            self.out_channels = self.in_channels
        kh, kw = self.kernel_size
        w_shape = (kh, kw, self.in_channels, self.out_channels)
        b_shape = (self.out_channels,)
        wb = self.initializer(kh * kw * self.in_channels + 1, self.out_channels)
        w = wb[:-1, :].reshape(w_shape) if self.w is None else self.w
        b = wb[-1, :].reshape(b_shape) if self.b is None else self.b
        w = w.value if isinstance(w, Param) else w
        b = b.value if isinstance(b, Param) else b
        assert tuple(w.shape) == w_shape, f'{tuple(w.shape)} != {w_shape}'
        assert tuple(b.shape) == b_shape, f'{tuple(b.shape)} != {b_shape}'
        self.w = Param(w, optimizer=self.optimizer)
        self.b = Param(b, optimizer=self.optimizer)
        self._init_optimizer()
        self.is_initialized = True

    def _forward(self, X, mem_id=0):
        assert X.shape[3] == self.in_channels
        self._mem[mem_id] = X
        return ops.conv2d_fwd(X, self.w.value, self.b.value, self.stride, self.padding, self.padding_value,
                              self.bias)

    def _backward(self, grad, mem_id=0):
        X = self._mem[mem_id]
        ops.conv2d_bwd_weight(X, grad, self.w.grad, self.b.grad, self.stride, self.padding, self.padding_value,
                              self.bias, accumulate=True)
        if not self.needs_input_grad:                 # Model.skip_input_grads: nobody reads dX of this layer
            return None
        return ops.conv2d_bwd_data(grad, self.w.value, X.shape, self.stride, self.padding)

    # conv + following activation as ONE forward kernel (Model.enable_fusion): the pre-activation
    # tensor is never written; backward recovers the activation gradient from the output
    @track_method('forward')
    def forward_fused(self, inputs, activation):
        X = ops.as_device(make_list_if_not(inputs)[0])
        assert X.shape[3] == self.in_channels
        self._mem[0] = X
        y = ops.conv2d_fwd(X, self.w.value, self.b.value, self.stride, self.padding, self.padding_value,
                           self.bias, act=activation.kind, alpha=activation.alpha)
        self._fused_out = y
        return [y]

    @track_method('backward')
    def backward_fused(self, grads, activation=None, act_grad_applied=False, input_activation=None):
        """Backward of the fused graph.  `activation`: the layer fused into this conv's forward (its
        gradient is taken from the stored output unless the consumer already applied it:
        `act_grad_applied`).  `input_activation`: this conv's input is the output of a fused
        activation whose only consumer is this conv -- dx is stored already multiplied by that
        activation's derivative (one pass less over the tensor)."""
        grad = ops.as_device(make_list_if_not(grads)[0])
        if activation is not None and not act_grad_applied:
            grad = ops.act_bwd_from_output(activation.kind, self._fused_out, grad, activation.alpha)
        X = self._mem[0]
        ops.conv2d_bwd_weight(X, grad, self.w.grad, self.b.grad, self.stride, self.padding, self.padding_value,
                              self.bias, accumulate=True)
        if not self.needs_input_grad:
            dx = None
        elif input_activation is None:
            dx = ops.conv2d_bwd_data(grad, self.w.value, X.shape, self.stride, self.padding)
        else:
            dx = ops.conv2d_bwd_data(grad, self.w.value, X.shape, self.stride, self.padding, x_act=X,
                                     act=input_activation.kind, alpha=input_activation.alpha)
        self._fused_out = None
        self.clear_memory()
        return [dx]

    # Upsample2D(2) + this conv evaluated on the low-res tensor (Model._find_ups; csrc/conv_up.hip)
    @track_method('forward')
    def forward_up(self, x_low, activation=None):
        x_low = ops.as_device(x_low)
        assert x_low.shape[3] == self.in_channels
        self._mem[0] = x_low
        # float32, 4 channels: the forward kernel leaves the per-phase weights for this step's backward_up
        self._weff = None
        if self.in_channels == 4 and x_low.dtype == np.float32:
            if getattr(self, '_weff_buf', None) is None:
                self._weff_buf = CP.empty((576,), np.float32)
            self._weff = self._weff_buf
        y = ops.upconv2x_fwd(x_low, self.w.value, self.b.value, self.padding, self.bias,
                             None if activation is None else activation.kind,
                             0.0 if activation is None else activation.alpha, weff=self._weff)
        self._fused_out = y if activation is not None else None
        return y

    @track_method('backward')
    def backward_up(self, grads, activation=None, act_grad_applied=False, input_activation=None):
        """Returns the gradient w.r.t. the LOW-RES input (upsample backward included); the arguments as
        backward_fused."""
        grad = ops.as_device(make_list_if_not(grads)[0])
        if activation is not None and not act_grad_applied:
            grad = ops.act_bwd_from_output(activation.kind, self._fused_out, grad, activation.alpha)
        x_low = self._mem[0]
        ops.upconv2x_bwd_weight(x_low, grad, self.w.grad, self.b.grad, self.padding, self.bias, accumulate=True)
        weff = getattr(self, '_weff', None)               # written by forward_up of this step (same weights)
        self._weff = None
        if input_activation is None:
            dx = ops.upconv2x_bwd_data(grad, self.w.value, x_low.shape, self.padding, weff=weff)
        else:
            dx = ops.upconv2x_bwd_data(grad, self.w.value, x_low.shape, self.padding, x_act=x_low,
                                       act=input_activation.kind, alpha=input_activation.alpha, weff=weff)
        self._fused_out = None
        self.clear_memory()
        return dx

    # this conv as the SECOND of conv3x3(1->16)+LeakyReLU+conv3x3(16->1)[+Sigmoid]: one kernel each way
    # (Model._find_pairs; csrc/conv_pair.hip); `first` is the 1->16 conv whose output is never stored
    @track_method('forward')
    def forward_pair(self, X, first, activation, out_activation):
        X = ops.as_device(X)
        first._mem[0] = X
        y = ops.conv_pair_fwd(X, first.w.value, first.b.value, self.w.value, self.b.value, first.padding_value,
                              first.bias, self.bias, activation.alpha,
                              ops.ACT_CODES[None if out_activation is None else out_activation.kind])
        self._fused_out = y
        return y

    @track_method('backward')
    def backward_pair(self, grads, first, activation, out_activation):
        grad = ops.as_device(make_list_if_not(grads)[0])
        X = first._mem[0]
        dx = ops.conv_pair_bwd(X, self._fused_out, grad, first.w.value, first.b.value, self.w.value,
                               first.w.grad, first.b.grad, self.w.grad, self.b.grad, first.padding_value,
                               first.bias, self.bias, activation.alpha,
                               ops.ACT_CODES[None if out_activation is None else out_activation.kind],
                               need_dx=first.needs_input_grad, accumulate=True)
        self._fused_out = None
        first.clear_memory()
        self.clear_memory()
        return dx

    def get_output_shapes(self, input_shapes):
        batch, height, width, _ = make_list_if_not(input_shapes)[0]
        oh, ow = ops.conv_out_hw(height, width, self.kernel_size, self.stride, self.padding)
        return [(batch, oh, ow, self.out_channels)]

    def changes_receptive_field(self):
        return True

    def _get_receptive_field(self, axis, position, output_id):
        assert 0 <= axis < 2, f'Convolutional2D has two axis, found {axis}'
        assert output_id < self.get_outputs_count(), f'This layer has only {self.get_outputs_count()} outputs'
        key = (axis, position, output_id)
        if key not in self._receptive_fields:
            start = position * self.stride[axis] - self.padding[axis]
            self._receptive_fields[key] = {0: set(range(start, start + self.kernel_size[axis]))}
        return self._receptive_fields[key]

    def params(self):
        return {'w': self.w, 'b': self.b}


class Conv2DToBatchedFixedWidthed(BaseLayer):
    """convolutional.py:330-373: sliding window of `width` columns over the W axis (zero padded by
    width//2 on the left), every window becomes one batch entry: (B,H,W,C) -> (B*W,H,width,C).
    The reference loops over B*W slices in Python even on the GPU; here it is one gather kernel
    forward and one gather kernel backward."""

    def __init__(self, width, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.width = width

    def _forward(self, X, mem_id=0):
        self.get_output_shapes(X.shape)
        self._mem[mem_id] = X.shape
        return ops.fixed_width_fwd(X, self.width)

    def _backward(self, grad, mem_id=0):
        return ops.fixed_width_bwd(grad, self._mem[mem_id], self.width)

    def get_output_shapes(self, input_shapes):
        result = []
        for bs, h, w, ch in make_list_if_not(input_shapes):
            assert w >= self.width, f'Input width must be >= than output width, found: {w} < {self.width}'
            result.append((bs * w, h, self.width, ch))
        return result
