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

    # pooling / activation / loss / optimizer kernels at my_model-like shapes: algorithmic bytes / time
    import numpy as np
    rng = np.random.default_rng(0)
    x4 = CP.copy(rng.standard_normal((32, 256, 512, 4)).astype(np.float32))
    g4 = CP.copy(rng.standard_normal((32, 256, 512, 4)).astype(np.float32))
    lo4 = CP.copy(rng.standard_normal((32, 128, 256, 4)).astype(np.float32))
    y4, mask = ops.maxpool2d_fwd(x4, (2, 2), (2, 2), (0, 0))
    gy4 = CP.copy(rng.standard_normal(y4.shape).astype(np.float32))
    p1 = CP.copy(rng.random((32, 256, 512, 1)).astype(np.float32))
    t1 = CP.copy((rng.random((32, 256, 512, 1)) > 0.5).astype(np.float32))
    logits = CP.copy(rng.standard_normal((2048, 162)).astype(np.float32))
    onehot = CP.copy(np.eye(162, dtype=np.float32)[rng.integers(0, 162, 2048)])
    wbuf, gbuf, vbuf = CP.zeros((1 << 22,)), CP.zeros((1 << 22,)), CP.zeros((1 << 22,))
    mb = lambda *arrs: sum(a.nbytes for a in arrs)
    rows = (
        ('maxpool 2x2 fwd (x -> y + u8 mask)', lambda: ops.maxpool2d_fwd(x4, (2, 2), (2, 2), (0, 0)), mb(x4, y4, mask)),
        ('maxpool 2x2 bwd (dy, mask -> dx)', lambda: ops.maxpool2d_bwd(gy4, mask, x4.shape, (2, 2), (2, 2), (0, 0)),
         mb(gy4, mask, x4)),
        ('upsample 2x fwd (4 ch, vector)', lambda: ops.upsample2d_fwd(lo4, (2, 2)), mb(lo4, x4)),
        ('upsample 2x bwd (4 ch, vector)', lambda: ops.upsample2d_bwd(g4, lo4.shape, (2, 2)), mb(lo4, x4)),
        ('leaky bwd from output', lambda: ops.act_bwd_from_output('leaky', x4, g4, 0.01), mb(x4, g4, x4)),
        ('sigmoid fwd', lambda: rt.call('uocr_act_fwd', x4.code, 3, 0.0, x4.ptr, g4.ptr, x4.size), mb(x4, g4)),
        ('dice loss + grad (1 ch, folded sigmoid)', lambda: ops.seg_loss('dice', p1, t1, True, out_act='sigmoid'),
         mb(p1, t1) + mb(p1, t1, p1)),
        ('softmax CE + grad (2048 x 162)', lambda: ops.softmax_ce(logits, onehot, True), mb(logits, onehot, logits)),
        ('SGD step (4M parameters)', lambda: ops.momentum_step(wbuf, gbuf, vbuf, 0.01, 0.0), mb(wbuf, gbuf) + mb(wbuf, vbuf)),
    )
    for name, fn, nb in rows:
        us = timed(fn)
        print(f'{name:40s} {us:8.1f} us  {nb / 1e6:8.1f} MB  {nb / us / 1e3:8.0f} GB/s')


if __name__ == '__main__':
    main()
