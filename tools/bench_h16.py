"""Line-net layers at the highres-fp16 size (8 pages 1024x2048, binary16 storage): the binary16-MFMA kernels
(conv_h16.hip, option h16=1) against the vector-ALU kernels they replace (h16=0).
    python tools/bench_h16.py [--batch 8] [--reps 10]"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    args = ap.parse_args()
    from univer_ocr_amd.nn import CP, ops
    CP.use_gpu(0)
    CP.set_dtype('float16')
    rt = CP.runtime()
    ev = [ctypes.c_void_p() for _ in range(2)]
    for e in ev:
        assert rt.lib.uocr_event_create(ctypes.byref(e)) == 0

    def timed(fn):
        fn()
        rt.synchronize()
        rt.call('uocr_event_record', ev[0])
        for _ in range(args.reps):
            fn()
        rt.call('uocr_event_record', ev[1])
        ms = ctypes.c_float()
        assert rt.lib.uocr_event_elapsed_ms_sync(ev[0], ev[1], ctypes.byref(ms)) == 0
        return ms.value * 1e3 / args.reps

    rng = np.random.default_rng(0)
    n, h, w = args.batch, args.height, args.width

    def act(shape):
        return CP.copy(rng.standard_normal(shape).astype(np.float32))

    def par(shape, s=0.2):
        return CP.copy(rng.standard_normal(shape) * s, np.float32)

    x4, g2, g4 = act((n, h, w, 4)), act((n, h, w, 2)), act((n, h, w, 4))
    xl = act((n, h // 2, w // 2, 4))
    w42, b2 = par((5, 5, 4, 2)), par((2,))
    w44, b4 = par((5, 5, 4, 4)), par((4,))
    dw42, db2 = CP.zeros((5, 5, 4, 2), np.float32), CP.zeros((2,), np.float32)
    dw44, db4 = CP.zeros((5, 5, 4, 4), np.float32), CP.zeros((4,), np.float32)
    px = n * h * w
    rows = [
        ('end 5x5 4->2 fwd+sigmoid', lambda: ops.conv2d_fwd(x4, w42, b2, (1, 1), (2, 2), 0.0, True, act='sigmoid'), 12 * px),
        ('end 5x5 4->2 dx+lrelu mask', lambda: ops.conv2d_bwd_data(g2, w42, x4.shape, (1, 1), (2, 2), x_act=x4, act='leaky', alpha=0.01), 20 * px),
        ('end 5x5 4->2 dw', lambda: ops.conv2d_bwd_weight(x4, g2, dw42, db2, (1, 1), (2, 2), 0.0, True, accumulate=False), 12 * px),
        ('up  5x5 4->4 fwd+lrelu', lambda: ops.upconv2x_fwd(xl, w44, b4, (2, 2), True, act='leaky', alpha=0.01), 10 * px),
        ('up  5x5 4->4 dx+lrelu mask', lambda: ops.upconv2x_bwd_data(g4, w44, xl.shape, (2, 2), x_act=xl, act='leaky', alpha=0.01), 12 * px),
        ('up  5x5 4->4 dw', lambda: ops.upconv2x_bwd_weight(xl, g4, dw44, db4, (2, 2), True, accumulate=False), 10 * px),
    ]
    xh4, gh4 = act((n, h // 2, w // 2, 4)), act((n, h // 4, w // 4, 4))
    rows += [
        ('down 5x5 s2 4->4 fwd+lrelu', lambda: ops.conv2d_fwd(xh4, w44, b4, (2, 2), (2, 2), 0.0, True, act='leaky', alpha=0.01), 2.5 * px),
        ('down 5x5 s2 4->4 dx+mask', lambda: ops.conv2d_bwd_data(gh4, w44, xh4.shape, (2, 2), (2, 2), x_act=xh4, act='leaky', alpha=0.01), 4.5 * px),
        ('down 5x5 s2 4->4 dw', lambda: ops.conv2d_bwd_weight(xh4, gh4, dw44, db4, (2, 2), (2, 2), 0.0, True, accumulate=False), 2.5 * px),
    ]
    x1, g1, xl1 = act((n, h, w, 1)), act((n, h, w, 1)), act((n, h // 2, w // 2, 1))
    w11, b1 = par((5, 5, 1, 1)), par((1,))
    w14 = par((5, 5, 1, 4))
    dw11, db1 = CP.zeros((5, 5, 1, 1), np.float32), CP.zeros((1,), np.float32)
    rows += [
        ('par end 5x5 1->1 fwd+sigmoid', lambda: ops.conv2d_fwd(x1, w11, b1, (1, 1), (2, 2), 0.0, True, act='sigmoid'), 4 * px),
        ('par end 5x5 1->1 dx+mask', lambda: ops.conv2d_bwd_data(g1, w11, x1.shape, (1, 1), (2, 2), x_act=x1, act='leaky', alpha=0.01), 6 * px),
        ('par end 5x5 1->1 dw', lambda: ops.conv2d_bwd_weight(x1, g1, dw11, db1, (1, 1), (2, 2), 0.0, True, accumulate=False), 4 * px),
        ('par up 1->1 fwd+lrelu', lambda: ops.upconv2x_fwd(xl1, w11, b1, (2, 2), True, act='leaky', alpha=0.01), 2.5 * px),
        ('par up 1->1 dx+mask', lambda: ops.upconv2x_bwd_data(g1, w11, xl1.shape, (2, 2), x_act=xl1, act='leaky', alpha=0.01), 3 * px),
        ('par up 1->1 dw', lambda: ops.upconv2x_bwd_weight(xl1, g1, dw11, db1, (2, 2), True, accumulate=False), 2.5 * px),
        ('par down s2 1->1 fwd', lambda: ops.conv2d_fwd(x1, w11, b1, (2, 2), (2, 2), 0.0, True, act='leaky', alpha=0.01), 2.5 * px),
        ('par down s2 1->1 dx', lambda: ops.conv2d_bwd_data(xl1, w11, x1.shape, (2, 2), (2, 2)), 2.5 * px),
        ('line down s2 1->4 fwd', lambda: ops.conv2d_fwd(x1, w14, b4, (2, 2), (2, 2), 0.0, True, act='leaky', alpha=0.01), 4 * px),
        ('par down s2 1->1 dw', lambda: ops.conv2d_bwd_weight(x1, xl1, dw11, db1, (2, 2), (2, 2), 0.0, True, accumulate=False), 2.5 * px),
        ('line down s2 1->4 dw', lambda: ops.conv2d_bwd_weight(x1, xh4, CP.zeros((5, 5, 1, 4), np.float32), db4, (2, 2), (2, 2), 0.0, True, accumulate=False), 4 * px),
        ('line down s2 1->4 dx', lambda: ops.conv2d_bwd_data(xh4, w14, x1.shape, (2, 2), (2, 2)), 4 * px),
    ]
    print(f'{"layer":30s} {"h16=0 us":>10s} {"h16=1 us":>10s} {"MB":>8s} {"GB/s":>8s}')
    for name, fn, nbytes in rows:
        t = []
        for flag in (0, 1):
            rt.set_option('h16', flag)
            t.append(timed(fn))
        print(f'{name:30s} {t[0]:10.1f} {t[1]:10.1f} {nbytes / 1e6:8.1f} {nbytes / t[1] / 1e3:8.0f}', flush=True)


if __name__ == '__main__':
    main()
