"""Layer base classes and the simple layers (reference: nn/layers/layers.py:10-418).

Same names, constructor arguments and list-in / list-out `forward` / `backward` protocol.  The
bodies call libuniver_hip.so through nn/ops.py; nothing is computed on the host.

Differences that are deliberate (MI355X-first) and invisible to callers:
  * `Param.value` / `Param.grad` keep their storage: assigning a new array copies INTO it, so a
    model's parameters can live as views of one flat buffer (`ParamPack`) -- one optimizer launch,
    one gradient all-reduce (RCCL) and one memset per step instead of one per parameter;
  * no per-layer stream synchronisation (the reference calls cuda.synchronize() after every kernel).
"""
import numpy as np

from .. import ops
from ..gpu import CP, DeviceArray
from ..help_func import make_list_if_not
from ..initializers import kaiming_uniform
from ..optimizers import Adam
from ..progress_tracker import BaseProgressTracker, track_method

PACK_ALIGN = 64      # elements; keeps every parameter view 256-byte aligned in float32


class Param:
    """layers.py:10-21."""

    def __init__(self, value, optimizer=None):
        self._value = CP.copy(value, CP.param_dtype())
        self._grad = CP.zeros(self._value.shape, self._value.dtype)
        self._pack = None                 # set by ParamPack: any access to .grad marks the flat buffer dirty
        self.optimizer = optimizer
        if optimizer is not None:
            optimizer.add_param(self)

    @staticmethod
    def _assign(dst, src):
        if isinstance(src, DeviceArray):
            if src.shape != dst.shape:
                return None
            if src.t is not dst.t:
                dst.t.copy_(src.t)
            return dst
        host = np.asarray(src)
        if host.shape != dst.shape:
            return None
        return dst.set(host)

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, new):
        if self._assign(self._value, new) is None:
            self._value = CP.copy(new, self._value.dtype)       # shape changed: new storage

    @property
    def grad(self):
        if self._pack is not None:
            self._pack.grad_dirty = True
        return self._grad

    @grad.setter
    def grad(self, new):
        if self._assign(self._grad, new) is None:
            self._grad = CP.copy(new, self._grad.dtype)

    def update_grad(self):
        self.optimizer.update(self)

    def clear_grad(self):
        ops.zero_(self._grad)


class ParamPack:
    """All parameters of a model as views of ONE flat value buffer and ONE flat grad buffer."""

    def __init__(self, params):
        self.params = list(params)
        self.dtype = self.params[0].value.dtype if self.params else CP.param_dtype()
        self.entries = []
        off = 0
        for p in self.params:
            size = p.value.size
            self.entries.append((p, off, size))
            off += -(-size // PACK_ALIGN) * PACK_ALIGN
        self.total = max(off, PACK_ALIGN)
        self.value = CP.zeros((self.total,), self.dtype)
        self._grad = CP.zeros((self.total,), self.dtype)
        for p, off, size in self.entries:
            v = self.view_of(self.value, off, size, p.value.shape)
            g = self.view_of(self._grad, off, size, p.value.shape)
            v.t.copy_(p._value.t)
            g.t.copy_(p._grad.t)
            p._value, p._grad, p._pack = v, g, self
        self.grad_dirty = True            # False only right after zero_grad(): the next zero_grad() is then free

    @property
    def grad(self):
        self.grad_dirty = True
        return self._grad

    @staticmethod
    def view_of(flat, off, size, shape):
        return DeviceArray(flat.t[off:off + size].view(*shape))

    def rebase(self, value_flat, grad_flat, base):
        """Move the pack into [base, base + total) of two larger flat buffers (parallel.DataParallel puts the
        packs of all models of a step next to each other so that neighbouring gradients can go out as one
        collective).  Parameter views are re-pointed; must happen before any HIP graph is captured."""
        new_value = DeviceArray(value_flat.t[base:base + self.total])
        new_grad = DeviceArray(grad_flat.t[base:base + self.total])
        new_value.t.copy_(self.value.t)
        new_grad.t.copy_(self._grad.t)
        self.value, self._grad = new_value, new_grad
        for p, off, size in self.entries:
            p._value = self.view_of(self.value, off, size, p._value.shape)
            p._grad = self.view_of(self._grad, off, size, p._grad.shape)
        self.grad_dirty = True

    def zero_grad(self):
        """The reference zeroes every gradient at the start of forward AND after the update
        (models.py:188,282); the second memset of the pair finds the buffer untouched and is skipped."""
        if self.grad_dirty:
            ops.zero_(self._grad)
            self.grad_dirty = False

    def same_optimizer(self):
        opts = {id(p.optimizer) for p in self.params}
        return self.params[0].optimizer if len(opts) == 1 and self.params else None


class BaseLayer:
    """layers.py:24-166."""

    def __init__(self, name=None, input_shapes=None, trainable=True, initializer=kaiming_uniform,
                 regularizer=None, optimizer=None):
        self.name = name
        self.input_shapes = input_shapes
        self.inputs_count = len(input_shapes) if input_shapes is not None else None
        self.trainable = trainable
        self.needs_input_grad = True
        self.initializer = initializer
        self.regularizer = regularizer
        self.optimizer = Adam() if optimizer is None else optimizer
        self.is_initialized = True
        self._mem = {}
        self._receptive_fields = {}
        self.progress_tracker = BaseProgressTracker()

    # -- initialisation ---------------------------------------------------------------------
    def initialize_from_X(self, X):
        self.initialize([x.shape for x in make_list_if_not(X)])

    def initialize(self, input_shapes):
        self.input_shapes = input_shapes
        self.inputs_count = len(input_shapes)
        self.is_initialized = True

    # -- forward / backward protocol ----------------------------------------------------------
    @track_method('forward')
    def forward(self, inputs):
        assert self.is_initialized, 'You must initialize() layer before calling forward() method'
        return [self._forward(ops.as_device(X), mem_id) for mem_id, X in enumerate(make_list_if_not(inputs))]

    @track_method('backward')
    def backward(self, grads):
        result = [self._backward(ops.as_device(g), mem_id) for mem_id, g in enumerate(make_list_if_not(grads))]
        self.clear_memory()
        return result

    def _forward(self, X, mem_id=0):
        raise NotImplementedError()

    def _backward(self, grad, mem_id=0):
        raise NotImplementedError()

    # -- parameters ------------------------------------------------------------------------------
    def params(self):
        return {}

    def update_grads(self):
        if self.trainable:
            for param in self.params().values():
                param.update_grad()

    def clear_grads(self):
        for param in self.params().values():
            param.clear_grad()

    def clear_memory(self):
        self._mem = {}

    def get_weights(self):
        return {name: param.value.tolist() for name, param in self.params().items()}

    def set_weights(self, weights):
        """layers.py:123-137: tensors with NaN or a wrong shape are skipped with a message."""
        for name, param in self.params().items():
            new = weights.get(name)
            if new is None:
                continue
            new = np.array(new)
            problem = None
            if np.any(np.isnan(new)):
                problem = 'NaN found in loaded weights'
            elif new.shape != param.value.shape:
                problem = f'Shapes don`t match: {new.shape} != {param.value.shape}'
            if problem:
                print(f'{self.name}/{name}: {problem}, skipping')
                continue
            param.value = new

    def nan_weights(self):
        return any(ops.has_nan(param.value) for param in self.params().values())

    def count_parameters(self, param=None):
        if param is not None:
            return self.params()[param].value.size
        return sum(p.value.size for p in self.params().values())

    def regularize(self, slot=None):
        """layers.py:147-155.  With a device `slot` the loss is accumulated there (no host sync)."""
        if self.regularizer is None:
            return 0
        total = 0
        for param in self.params().values():
            if slot is not None:
                self.regularizer.apply(param.value, param.grad, slot, accumulate=True)
            else:
                total += self.regularizer.apply(param.value, param.grad)
        return total

    # -- shapes / graph queries --------------------------------------------------------------------
    def get_all_output_shapes(self, input_shapes):
        return self.get_output_shapes(input_shapes), {}

    def get_output_shapes(self, input_shapes):
        raise NotImplementedError()

    def get_outputs_count(self):
        return 1

    def is_fully_convolutional(self):
        return True

    def changes_receptive_field(self):
        return False

    def _get_receptive_field(self, axis, position, output_id):
        assert output_id < self.get_outputs_count(), f'This layer has only {self.get_outputs_count()} outputs'
        return {0: {position}}

    def _clear_receptive_fields_info(self):
        self._receptive_fields = {}

    def _set_name(self, name):
        self.name = name

    def _init_optimizer(self):
        for param in self.params().values():
            if id(param) not in self.optimizer.groups:
                self.optimizer.add_param(param)

    def init_progress_tracker(self, progress_tracker, set_names_recursively=False):
        self.progress_tracker = progress_tracker
        self.progress_tracker.register_layer(self.name)


class BaseLayerGPU(BaseLayer):
    """The reference's BaseLayerGPU (layers.py:169-237) picks a CPU or a numba closure per call;
    here there is one path (HIP), so it only keeps the name for isinstance checks."""

    def _init_forward_backward(self):
        pass


class Concat(BaseLayer):
    """layers.py:240-284."""

    def __init__(self, axis=-1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.axis = axis
        self.is_initialized = self.inputs_count is not None

    @track_method('forward')
    def forward(self, inputs):
        if not isinstance(inputs, list):
            self._mem = [inputs.shape]
            return inputs
        inputs = [ops.as_device(x) for x in inputs]
        self._mem = [x.shape for x in inputs]
        return [ops.concat(inputs, self.axis)]

    @track_method('backward')
    def backward(self, grads):
        grad = ops.as_device(make_list_if_not(grads)[0])
        result = ops.split(grad, self._mem, self.axis)
        self.clear_memory()
        return result

    def get_output_shapes(self, input_shapes):
        input_shapes = make_list_if_not(input_shapes)
        out = list(input_shapes[0])
        axis = self.axis % len(out)
        out[axis] = sum(s[axis] for s in input_shapes)
        return [tuple(out)]

    def changes_receptive_field(self):
        return True

    def _get_receptive_field(self, axis, position, output_id):
        assert output_id < self.get_outputs_count(), f'This layer has only {self.get_outputs_count()} outputs'
        return {in_key: {position} for in_key in range(self.inputs_count)}


class Flatten(BaseLayer):
    """layers.py:287-304."""

    def _forward(self, X, mem_id=0):
        self._mem[mem_id] = X.shape
        return X.reshape(self.get_output_shapes(X.shape)[0])

    def _backward(self, grad, mem_id=0):
        return grad.reshape(self._mem[mem_id])

    def get_output_shapes(self, input_shapes):
        shape = make_list_if_not(input_shapes)[0]
        return [(shape[0], int(np.prod(shape[1:])))]

    def is_fully_convolutional(self):
        return False

    def _get_receptive_field(self, axis, position, output_id):
        raise NotImplementedError('The method is not supported by Flatten Layer')


class FullyConnected(BaseLayer):
    """layers.py:307-363.  w has shape (n_input + 1, n_output); its last row is the bias."""

    def __init__(self, n_input=None, n_output=None, w=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_input, self.n_output, self.w = n_input, n_output, w
        if self.input_shapes is None and n_input is not None:
            self.input_shapes = [(None, n_input)]
        if self.input_shapes is not None:
            self.initialize(self.input_shapes)
        else:
            self.is_initialized = False

    def initialize(self, input_shapes):
        self.input_shapes = input_shapes
        self.n_input = input_shapes[0][1]
        if self.n_output is None:
            self.n_output = self.n_input
        if self.w is None:
            w = self.initializer(self.n_input + 1, self.n_output)
        else:
            w = self.w.value if isinstance(self.w, Param) else self.w
            assert tuple(w.shape) == (self.n_input + 1, self.n_output)
        self.w = Param(w, optimizer=self.optimizer)
        self._init_optimizer()
        self.is_initialized = True

    def _forward(self, X, mem_id=0):
        self._mem[mem_id] = X
        return ops.dense_fwd(X, self.w.value)

    def _backward(self, grad, mem_id=0):
        return ops.dense_bwd(self._mem[mem_id], self.w.value, grad, self.w.grad, accumulate=True)

    # this layer + the LeakyRelu / Sigmoid that follows it as one GEMM (Model.enable_fusion); same contract as
    # Convolutional2D.forward_fused / backward_fused
    @track_method('forward')
    def forward_fused(self, inputs, activation):
        X = ops.as_device(make_list_if_not(inputs)[0])
        self._mem[0] = X
        y = ops.dense_fwd(X, self.w.value, activation.kind, activation.alpha)
        self._fused_out = y
        return y

    @track_method('backward')
    def backward_fused(self, grads, activation=None, act_grad_applied=False, input_activation=None):
        """`activation`: the fused activation behind this layer (its gradient is taken from the output unless
        the consumer already applied it); `input_activation`: the fused activation that produced this layer's
        input -- dx is then returned w.r.t. that activation's input."""
        grad = ops.as_device(make_list_if_not(grads)[0])
        if activation is not None and not act_grad_applied:
            grad = ops.act_bwd_from_output(activation.kind, self._fused_out, grad, activation.alpha)
        x = self._mem[0]
        if input_activation is None:
            dx = ops.dense_bwd(x, self.w.value, grad, self.w.grad, accumulate=True)
        else:
            dx = ops.dense_bwd(x, self.w.value, grad, self.w.grad, accumulate=True, x_act=input_activation.kind,
                               x_alpha=input_activation.alpha)
        self._fused_out = None
        self.clear_memory()
        return dx

    # this layer fed by Conv2DToBatchedFixedWidthed + Flatten: one implicit GEMM on the conv feature map
    # (Model._find_windows; ops.windows_dense_fwd)
    @track_method('forward')
    def forward_windows(self, x, width, activation=None):
        x = ops.as_device(x)
        self._mem[0] = x
        y = ops.windows_dense_fwd(x, self.w.value, width, None if activation is None else activation.kind,
                                  0.0 if activation is None else activation.alpha)
        self._fused_out = y if activation is not None else None
        return y

    @track_method('backward')
    def backward_windows(self, grads, width, activation=None, act_grad_applied=False, input_activation=None):
        """Returns the gradient w.r.t. the feature map the windows were cut from; with `input_activation`
        (the fused activation that produced it) times that activation's derivative.  `activation` as in
        backward_fused."""
        grad = ops.as_device(make_list_if_not(grads)[0])
        if activation is not None and not act_grad_applied:
            grad = ops.act_bwd_from_output(activation.kind, self._fused_out, grad, activation.alpha)
        x = self._mem[0]
        if input_activation is None:
            dx = ops.windows_dense_bwd(x, self.w.value, grad, self.w.grad, width, accumulate=True)
        else:
            dx = ops.windows_dense_bwd(x, self.w.value, grad, self.w.grad, width, accumulate=True, x_act=x,
                                       act=input_activation.kind, alpha=input_activation.alpha)
        self._fused_out = None
        self.clear_memory()
        return dx

    def get_output_shapes(self, input_shapes):
        return [(make_list_if_not(input_shapes)[0][0], self.n_output)]

    def is_fully_convolutional(self):
        return False

    def changes_receptive_field(self):
        return True

    def _get_receptive_field(self, axis, position, output_id):
        raise NotImplementedError('The method is not supported by Fully Connected Layer')

    def params(self):
        return {'w': self.w}


class Noop(BaseLayer):
    """layers.py:366-374."""

    def _forward(self, X, mem_id=0):
        return X

    def _backward(self, grad, mem_id=0):
        return grad

    def get_output_shapes(self, input_shapes):
        return make_list_if_not(input_shapes)


class _Activation(BaseLayer):
    kind = None
    alpha = 0.0

    def _forward(self, X, mem_id=0):
        self._mem[mem_id] = X               # the reference stashes the mask / X (layers.py:379,396,409)
        return ops.act_fwd(self.kind, X, self.alpha)

    def _backward(self, grad, mem_id=0):
        return ops.act_bwd(self.kind, self._mem[mem_id], grad, self.alpha)

    def get_output_shapes(self, input_shapes):
        return make_list_if_not(input_shapes)


class Relu(_Activation):
    """layers.py:377-387 (mask is X >= 0)."""
    kind = 'relu'


class LeakyRelu(_Activation):
    """layers.py:390-404."""
    kind = 'leaky'

    def __init__(self, alpha=0.01, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.alpha = alpha


class Sigmoid(_Activation):
    """layers.py:407-418 (backward recomputed from the stashed input)."""
    kind = 'sigmoid'
