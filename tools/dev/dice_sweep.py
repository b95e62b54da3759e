#!/usr/bin/env python3
"""Dice loss + gradient alone (HIP events, 4 rotating buffer sets): chunk cap of the sums kernel (ctx option pair_band)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from univer_ocr_amd.nn import CP, ops  # noqa: E402

CP.use_gpu(0)
rt = CP.runtime()
CP.lazy_losses = True


def event():
    ev = ctypes.c_void_p()
    assert rt.lib.uocr_event_create(ctypes.byref(ev)) == 0
    return ev


def time_us(fn, reps=20):
    fn(0)
    rt.call('uocr_stream_sync')
    a, b = event(), event()
    rt.call('uocr_event_record', a)
    for i in range(reps):
        fn(i)
    rt.call('uocr_event_record', b)
    ms = ctypes.c_float()
    assert rt.lib.uocr_event_elapsed_ms_sync(a, b, ctypes.byref(ms)) == 0
    return ms.value * 1e3 / reps


for dtype, shape in (('float32', (32, 256, 512, 1)), ('float32', (32, 256, 512, 2)), ('float16', (8, 1024, 2048, 1)),
                     ('float16', (8, 1024, 2048, 2))):
    CP.set_dtype(dtype)
    rng = np.random.default_rng(0)
    sets = [(CP.copy(rng.random(shape).astype(np.float32)), CP.copy((rng.random(shape) > 0.6).astype(np.float32)))
            for _ in range(4)]
    for cap in (0,):
        rt.set_option('pair_band', cap)
        both = time_us(lambda i: ops.seg_loss('dice', *sets[i % 4], True, out_act='sigmoid'))
        value = time_us(lambda i: ops.seg_loss('dice', *sets[i % 4], False))
        print(f'{dtype} {shape} cap={cap}: sums + gradient {both:6.1f} us, sums only {value:6.1f} us')
    rt.set_option('pair_band', 0)
