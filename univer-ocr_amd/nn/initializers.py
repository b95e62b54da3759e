"""Weight initializers with the names of the reference's nn/initializers.py.

`name(in_num, out_num)` returns a host float64 array of shape (in_num, out_num); Param moves it to
the device.  Draws come from NumPy's global RNG, as in the reference, so `np.random.seed` makes a
model reproducible.  xavier_* scale by 1/sqrt(in_num), kaiming_* by sqrt(2/in_num); the *_uniform
variants draw U[0, 1) -- i.e. NON-NEGATIVE weights -- which is the reference's behaviour and its
default initializer (BaseLayer's `initializer=kaiming_uniform`), kept as is."""
import numpy as np

_SAMPLERS = {'normal': lambda shape: np.random.normal(size=shape),
             'uniform': lambda shape: np.random.uniform(size=shape)}
_GAINS = {'xavier': 1.0, 'kaiming': 2.0}


def _make(scheme, law):
    def initializer(in_num, out_num):
        scale = np.sqrt(_GAINS[scheme] / in_num)
        return scale * _SAMPLERS[law]((in_num, out_num))
    initializer.__name__ = f'{scheme}_{law}'
    initializer.__doc__ = f'{scheme} scaling, {law} draws; shape (in_num, out_num).'
    return initializer


xavier_normal = _make('xavier', 'normal')
xavier_uniform = _make('xavier', 'uniform')
kaiming_normal = _make('kaiming', 'normal')
kaiming_uniform = _make('kaiming', 'uniform')
