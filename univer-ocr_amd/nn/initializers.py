"""Weight initializers (reference: nn/initializers.py:4-25).  Host RNG (NumPy global state, as in
the reference); the arrays are moved to the device by Param.  Note kaiming_uniform draws U[0,1),
so default weights are non-negative -- kept as is."""
import numpy as np


def xavier_normal(in_num, out_num):
    return np.random.normal(size=(in_num, out_num)) / np.sqrt(in_num)


def xavier_uniform(in_num, out_num):
    return np.random.uniform(size=(in_num, out_num)) / np.sqrt(in_num)


def kaiming_normal(in_num, out_num):
    return np.random.normal(size=(in_num, out_num)) / np.sqrt(in_num / 2)


def kaiming_uniform(in_num, out_num):
    return np.random.uniform(size=(in_num, out_num)) / np.sqrt(in_num / 2)
