"""Small argument helpers (reference: nn/help_func.py:4-31, same names and error types)."""
from collections.abc import Iterable


def make_list_if_not(var):
    return var if isinstance(var, list) else [var]


def tuplize(name, var, length):
    """int -> (int,)*length; iterable of `length` ints -> tuple.  ValueError for negatives,
    TypeError for anything else (help_func.py:24-29)."""
    if isinstance(var, bool):
        candidate = None
    elif isinstance(var, int):
        candidate = (var,) * length
    elif isinstance(var, Iterable):
        items = tuple(var)
        ok = len(items) == length and all(isinstance(v, int) and not isinstance(v, bool) for v in items)
        candidate = items if ok else None
    else:
        candidate = None
    if candidate is not None and any(v < 0 for v in candidate):
        raise ValueError(f'{name} cannot be negative, found: {var}')
    if candidate is None:
        raise TypeError(f'{name} must be either int or iterable of ints of length {length}, '
                        f'found {type(var).__name__}')
    return candidate
