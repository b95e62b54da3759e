#!/usr/bin/env python3
"""What this GPU's HBM delivers to the simplest kernels of this library on a 268 MB tensor
(32x256x512x16 float32): write-only (fill), read+write (activation), 2 reads + 1 write (add)
-- the practical ceilings the conv kernels' GB/s are read against."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    from univer_ocr_amd.nn import CP, ops
    CP.use_gpu(0)
    rt = CP.runtime()
    ev = [ctypes.c_void_p() for _ in range(2)]
    for e in ev:
        assert rt.lib.uocr_event_create(ctypes.byref(e)) == 0

    def timed(fn, reps=20):
        fn()
        rt.synchronize()
        rt.call('uocr_event_record', ev[0])
        for _ in range(reps):
            fn()
        rt.call('uocr_event_record', ev[1])
        ms = ctypes.c_float()
        assert rt.lib.uocr_event_elapsed_ms_sync(ev[0], ev[1], ctypes.byref(ms)) == 0
        return ms.value * 1e3 / reps

    shape = (32, 256, 512, 16)
    a, b = CP.zeros(shape), CP.zeros(shape)
    out = CP.empty(shape)
    nbytes = a.nbytes
    for name, fn, passes in (
            ('fill (1 write)', lambda: ops.fill_(a, 1.0), 1),
            ('memset (1 write)', lambda: ops.zero_(a), 1),
            ('leaky fwd (1 read + 1 write)', lambda: rt.call('uocr_act_fwd', a.code, 2, 0.01, a.ptr, out.ptr, a.size), 2),
            ('add (2 reads + 1 write)', lambda: rt.call('uocr_add', a.code, a.ptr, b.ptr, out.ptr, a.size), 3),
            ('d2d copy (1 read + 1 write)', lambda: rt.call('uocr_d2d', out.ptr, a.ptr, nbytes), 2)):
        us = timed(fn)
        print(f'{name:32s} {us:8.1f} us  {passes * nbytes / us / 1e3:8.0f} GB/s')


if __name__ == '__main__':
    main()
