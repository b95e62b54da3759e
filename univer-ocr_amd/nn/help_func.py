"""Argument normalisation shared by the layer constructors.

API of the reference's nn/help_func.py (`make_list_if_not`, `tuplize`) including the exception types
its callers rely on: ValueError for a negative size, TypeError for anything that is not an int or
a sequence of exactly `length` ints."""
import numbers


def make_list_if_not(var):
    if isinstance(var, list):
        return var
    return [var]


def _is_int(value):
    return isinstance(value, numbers.Integral) and not isinstance(value, bool)


def tuplize(name, var, length):
    """`3` -> `(3, 3)`, `(5, 3)` -> `(5, 3)` for length 2."""
    if _is_int(var):
        values = (int(var),) * length
    else:
        try:
            values = tuple(var)
        except TypeError:
            values = None
        if isinstance(var, (str, bytes)) or values is None or len(values) != length or \
                not all(_is_int(v) for v in values):
            raise TypeError(f'{name} must be either int or iterable of ints of length {length}, '
                            f'found {type(var).__name__}')
        values = tuple(int(v) for v in values)
    if min(values) < 0:
        raise ValueError(f'{name} cannot be negative, found: {var}')
    return values
