#!/usr/bin/env python3
"""The pooling / activation / upsampling / loss kernels of the hot path on buffers that do not fit the Infinity Cache
(6 distinct sets per kernel, launches cycle through them): HIP-event time and algorithmic GB/s per kernel.  Run it under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes, tools/hbm_pmc.sh)
to get the HBM bytes the counters saw for the same launches; tools/hbm_summary.py puts the two side by side.

    python tools/hbm_kernels.py [--reps 12] > events.txt
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=12)
    ap.add_argument('--sets', type=int, default=6)
    args = ap.parse_args()
    from univer_ocr_amd.nn import CP, ops
    CP.use_gpu(0)
    CP.set_dtype('float32')
    CP.lazy_losses = True
    rt = CP.runtime()
    rng = np.random.default_rng(0)
    n, h, w = 32, 256, 512

    def rand(shape, binary=False):
        t = CP.empty(shape, np.float32)
        host = rng.random((1,) + tuple(shape[1:])) if binary else rng.standard_normal((1,) + tuple(shape[1:]))
        if binary:
            host = (host > 0.5)
        slab = CP.copy(host.astype(np.float32))
        for i in range(shape[0]):
            rt.call('uocr_d2d', t.ptr + i * slab.nbytes, slab.ptr, slab.nbytes)
        return t
    sets = []
    for _ in range(args.sets):
        x4, g4, lo4 = rand((n, h, w, 4)), rand((n, h, w, 4)), rand((n, h // 2, w // 2, 4))
        y4, mask = ops.maxpool2d_fwd(x4, (2, 2), (2, 2), (0, 0))
        sets.append(dict(x4=x4, g4=g4, lo4=lo4, y4=y4, mask=mask, gy4=rand(y4.shape), p1=rand((n, h, w, 1)),
                         t1=rand((n, h, w, 1), binary=True), hi4=rand((n, h, w, 4))))
    mb = lambda s, *keys: sum(s[k].nbytes for k in keys)   # noqa: E731
    s0 = sets[0]
    rows = [
        ('maxpool2d_fwd', 'MaxPool2D 2x2 fwd (x -> y + u8 mask)', lambda s: ops.maxpool2d_fwd(s['x4'], (2, 2), (2, 2), (0, 0)),
         mb(s0, 'x4', 'y4', 'mask')),
        ('maxpool2d_bwd', 'MaxPool2D 2x2 bwd (dy, mask -> dx)',
         lambda s: ops.maxpool2d_bwd(s['gy4'], s['mask'], s['x4'].shape, (2, 2), (2, 2), (0, 0)), mb(s0, 'gy4', 'mask', 'x4')),
        ('relu_fwd', 'Relu fwd', lambda s: ops.act_fwd('relu', s['x4']), 2 * s0['x4'].nbytes),
        ('leaky_fwd', 'LeakyRelu fwd', lambda s: ops.act_fwd('leaky', s['x4'], 0.01), 2 * s0['x4'].nbytes),
        ('leaky_bwd', 'LeakyRelu bwd from output', lambda s: ops.act_bwd_from_output('leaky', s['x4'], s['g4'], 0.01), 3 * s0['x4'].nbytes),
        ('sigmoid_fwd', 'Sigmoid fwd', lambda s: ops.act_fwd('sigmoid', s['x4']), 2 * s0['x4'].nbytes),
        ('upsample_fwd', 'Upsample2D 2x fwd (4 ch)', lambda s: ops.upsample2d_fwd(s['lo4'], (2, 2)), mb(s0, 'lo4', 'x4')),
        ('upsample_bwd', 'Upsample2D 2x bwd (4 ch)', lambda s: ops.upsample2d_bwd(s['hi4'], s['lo4'].shape, (2, 2)), mb(s0, 'hi4', 'lo4')),
        ('dice', 'Dice loss + grad (1 ch, output Sigmoid folded)',
         lambda s: ops.seg_loss('dice', s['p1'], s['t1'], True, out_act='sigmoid'), 5 * s0['p1'].nbytes),
    ]
    ev = [ctypes.c_void_p() for _ in range(2)]
    for e in ev:
        assert rt.lib.uocr_event_create(ctypes.byref(e)) == 0
    print(f'{"key":14s} {"kernel":52s} {"us":>8s} {"MB":>8s} {"GB/s":>8s}   ({args.sets} buffer sets, {args.reps} launches)')
    for key, label, fn, nbytes in rows:
        for s in sets:
            fn(s)
        rt.synchronize()
        rt.call('uocr_event_record', ev[0])
        for i in range(args.reps):
            fn(sets[i % len(sets)])
        rt.call('uocr_event_record', ev[1])
        ms = ctypes.c_float()
        assert rt.lib.uocr_event_elapsed_ms_sync(ev[0], ev[1], ctypes.byref(ms)) == 0
        us = ms.value * 1e3 / args.reps
        print(f'{key:14s} {label:52s} {us:8.1f} {nbytes / 1e6:8.1f} {nbytes / us / 1e3:8.0f}')


if __name__ == '__main__':
    main()
