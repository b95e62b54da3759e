#!/usr/bin/env python3
"""Fused optimizer tail (regularisers + update + gradient reset) on a Char-sized parameter pack."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    from univer_ocr_amd.nn import CP, ops
    CP.use_gpu(0)
    rt = CP.runtime()
    ev = [ctypes.c_void_p() for _ in range(2)]
    for e in ev:
        assert rt.lib.uocr_event_create(ctypes.byref(e)) == 0

    def timed(fn, reps=50):
        fn()
        rt.synchronize()
        rt.call('uocr_event_record', ev[0])
        for _ in range(reps):
            fn()
        rt.call('uocr_event_record', ev[1])
        ms = ctypes.c_float()
        assert rt.lib.uocr_event_elapsed_ms_sync(ev[0], ev[1], ctypes.byref(ms)) == 0
        return ms.value * 1e3 / reps

    rng = np.random.default_rng(0)
    for n in (802_000, 4_000_000):
        w = CP.copy(rng.standard_normal(n).astype(np.float32))
        g = CP.copy(rng.standard_normal(n).astype(np.float32))
        v, a = CP.zeros((n,)), CP.zeros((n,))
        l2 = (2, 0.01)
        three = [(l2, 0, 960), (l2, 1024, 62464), (l2, 62528, 123968)]
        for name, fn, nbytes in (
                ('momentum, no ranges', lambda: ops.momentum_step_fused(w, g, v, 1e-6, 0.9, []), 6 * 4 * n),
                ('momentum, 3 L2 ranges', lambda: ops.momentum_step_fused(w, g, v, 1e-6, 0.9, three), 6 * 4 * n),
                ('adam, 3 L2 ranges', lambda: ops.adam_step_fused(w, g, v, a, 1e-6, 0.9, 0.999, 1e-8, three), 8 * 4 * n),
                ('plain momentum_step', lambda: ops.momentum_step(w, g, v, 1e-6, 0.9), 5 * 4 * n)):
            us = timed(fn)
            print(f'n={n:8d} {name:24s} {us:8.1f} us  {nbytes / us / 1e3:8.0f} GB/s')


if __name__ == '__main__':
    main()
