"""Optimizers (reference: nn/optimizers.py:14-98): `add_param(param)`, `update(param)`, state keyed
by id(param).  Each update is ONE fused in-place kernel (the reference builds 4-6 temporaries per
parameter).  Semantics kept exactly:

  Adam      v = b1 v + (1-b1) g ; a = b2 a + (1-b2) g^2 ; w -= lr / (sqrt(a) + 1e-8) * v
            -- no bias correction, eps outside the sqrt (optimizers.py:56-61)
  Momentum  v = mu v - lr g ; w += v   (momentum=0 is the "SGD" of BASELINE config 3) (:75-78)
  RMSProp   a = rho a + (1-rho) g^2 ; w -= lr / (sqrt(a) + 1e-8) * g               (:92-95)
  Adagrad   the reference reads `state.lr`, which is never set, and raises AttributeError on the
            first update (:40); this mirror raises the same error type.

`update_pack(pack)` applies the same kernel to a whole ParamPack (all parameters of a model in
one flat buffer, see layers.ParamPack) in a single launch.
"""
from . import ops
from .gpu import CP

EPS = 1e-8


class State:
    def __init__(self, **arrays):
        self.__dict__.update(arrays)


class BaseOptimizer:
    state_names = ()

    def __init__(self):
        self.groups = {}
        self._pack_state = {}

    def add_param(self, param):
        self.groups[id(param)] = (param, None)      # state is allocated on first use / pack bind

    def _state_of(self, param):
        entry = self.groups.get(id(param))
        if entry is None:
            raise KeyError('parameter is not registered with this optimizer (add_param)')
        p, state = entry
        if state is None:
            state = State(**{n: CP.full(p.value.shape, init, p.value.dtype)
                             for n, init in zip(self.state_names, self.initials)})
            self.groups[id(param)] = (p, state)
        return state

    def bind_pack(self, pack):
        """Give every parameter of `pack` state views into flat state buffers (one per state name)."""
        flats = {n: CP.full((pack.total,), init, pack.dtype) for n, init in zip(self.state_names, self.initials)}
        for p, off, size in pack.entries:
            views = {n: pack.view_of(flats[n], off, size, p.value.shape) for n in self.state_names}
            old = self.groups.get(id(p), (p, None))[1]
            if old is not None:
                for n in self.state_names:
                    views[n].t.copy_(getattr(old, n).t)
            self.groups[id(p)] = (p, State(**views))
        self._pack_state[id(pack)] = flats
        return flats

    def update(self, param):
        raise NotImplementedError()

    def update_pack(self, pack):
        raise NotImplementedError()

    def _flats(self, pack):
        flats = self._pack_state.get(id(pack))
        if flats is None:
            flats = self.bind_pack(pack)
        return flats

    # -- hyper-parameters in device memory (the fused tail kernels read them there) ---------------------------------
    # A train step captured in a HIP graph freezes by-value kernel arguments; the reference's trainer decays lr
    # every epoch and on every NaN rollback (my_model/trainer.py:260), so the fused kernels take lr & co. from a
    # four-double device array that `refresh_hyper()` rewrites whenever the Python attributes have changed.
    def _hyper_values(self):
        return None

    def hyper_array(self):
        """The device array (created on first use on a GPU), brought up to date unless a graph is being captured."""
        import numpy as np
        import torch
        values = self._hyper_values()
        if values is None or not CP.has_device():
            return None
        if getattr(self, '_hyper', None) is None:
            self._hyper = CP.copy(np.asarray(values, dtype=np.float64), np.float64)
            self._hyper_host = tuple(values)
        elif not torch.cuda.is_current_stream_capturing():
            self.refresh_hyper()
        return self._hyper

    def refresh_hyper(self):
        """Push changed hyper-parameters to the device array.  Returns True when something changed.  Every stream
        may hold kernels that still read the old values (the nets share one optimizer), hence the device-wide
        synchronisation -- changes happen once per epoch, not per step."""
        import numpy as np
        import torch
        values = self._hyper_values()
        if values is None or getattr(self, '_hyper', None) is None or tuple(values) == self._hyper_host:
            return False
        torch.cuda.synchronize()
        self._hyper.set(np.asarray(values, dtype=np.float64))
        torch.cuda.synchronize()
        self._hyper_host = tuple(values)
        return True


class Adagrad(BaseOptimizer):
    state_names = ('accumulated',)

    def __init__(self, lr=0.01, initial_accumulated=0):
        super().__init__()
        self.lr = lr
        self.initials = [initial_accumulated]

    def update(self, param):
        raise AttributeError("'State' object has no attribute 'lr' (the reference's Adagrad.update reads "
                             "state.lr, which is never set: nn/optimizers.py:40)")

    update_pack = update


class Adam(BaseOptimizer):
    state_names = ('velocity', 'accumulated')

    def __init__(self, lr=0.001, beta1=0.9, beta2=0.999, initial_velocity=0, initial_accumulated=0):
        super().__init__()
        self.lr, self.beta1, self.beta2 = lr, beta1, beta2
        self.initials = [initial_velocity, initial_accumulated]

    def _hyper_values(self):
        return (float(self.lr), float(self.beta1), float(self.beta2), EPS)

    def update(self, param):
        s = self._state_of(param)
        ops.adam_step(param.value, param.grad, s.velocity, s.accumulated, self.lr, self.beta1, self.beta2, EPS)

    def update_pack(self, pack):
        f = self._flats(pack)
        ops.adam_step(pack.value, pack.grad, f['velocity'], f['accumulated'], self.lr, self.beta1, self.beta2, EPS)

    def update_pack_fused(self, pack, reg_ranges):
        """Regularisers + update + gradient reset in one pass over the pack (Model.train_finish)."""
        f = self._flats(pack)
        loss = ops.adam_step_fused(pack.value, pack.grad, f['velocity'], f['accumulated'], self.lr, self.beta1,
                                   self.beta2, EPS, reg_ranges, hyper=self.hyper_array())
        pack.grad_dirty = False
        return loss


class Momentum(BaseOptimizer):
    state_names = ('velocity',)

    def __init__(self, lr, momentum=0, initial_velocity=0):
        super().__init__()
        self.lr, self.momentum = lr, momentum
        self.velocity = 0
        self.initials = [initial_velocity]

    def _hyper_values(self):
        return (float(self.lr), float(self.momentum), 0.0, 0.0)

    def update(self, param):
        s = self._state_of(param)
        ops.momentum_step(param.value, param.grad, s.velocity, self.lr, self.momentum)

    def update_pack(self, pack):
        f = self._flats(pack)
        ops.momentum_step(pack.value, pack.grad, f['velocity'], self.lr, self.momentum)

    def update_pack_fused(self, pack, reg_ranges):
        """Regularisers + update + gradient reset in one pass over the pack (Model.train_finish).
        Returns the regularisation loss."""
        f = self._flats(pack)
        loss = ops.momentum_step_fused(pack.value, pack.grad, f['velocity'], self.lr, self.momentum, reg_ranges,
                                       hyper=self.hyper_array())
        pack.grad_dirty = False                     # the kernel left the gradient buffer zeroed
        return loss


class RMSProp(BaseOptimizer):
    state_names = ('accumulated',)

    def __init__(self, lr=0.01, rho=0.99, initial_accumulated=0):
        super().__init__()
        self.lr, self.rho = lr, rho
        self.initials = [initial_accumulated]

    def update(self, param):
        s = self._state_of(param)
        ops.rmsprop_step(param.value, param.grad, s.accumulated, self.lr, self.rho, EPS)

    def update_pack(self, pack):
        f = self._flats(pack)
        ops.rmsprop_step(pack.value, pack.grad, f['accumulated'], self.lr, self.rho, EPS)
