#!/usr/bin/env python3
"""python test_nn.py <test_gradients|test_identity> [use_gpu]   (same CLI as the reference's test_nn.py)

Runs the named script of univer_ocr_amd.nn.test through the HIP backend.  `use_gpu` defaults to
True here: this backend has no host compute path (use the reference for its NumPy mode)."""
import importlib
import sys


def main(test_name, use_gpu=True):
    module = importlib.import_module('univer_ocr_amd.nn.test.' + test_name)
    correct, total = module.main(str(use_gpu).lower() == 'true')
    return 0 if correct == total else 1


if __name__ == '__main__':
    sys.exit(main(*sys.argv[1:]))
