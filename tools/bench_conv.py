#!/usr/bin/env python3
"""Per-kernel micro-benchmark of Convolutional2D at the my_model workload shapes (batch 32,
256x512 pages): HIP-event time, algorithmic bytes and achieved GB/s (HBM roofline 8 TB/s spec,
~6.3 TB/s measured copy) or TFLOP/s (f32 MFMA roofline 157 TF) for forward, dx and dw/db.

    python tools/bench_conv.py [--reps 20] [--batch 32] [--option mfma=1] [--filter mono]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

LAYERS = [
    # name, H, W, cin, cout, kernel, stride, padding
    ('mono.conv_1', 256, 512, 1, 16, (3, 3), (1, 1), (1, 1)),
    ('mono.conv_2', 256, 512, 16, 1, (3, 3), (1, 1), (1, 1)),
    ('para.down_1', 256, 512, 1, 1, (5, 5), (2, 2), (2, 2)),
    ('para.down_2', 128, 256, 1, 1, (5, 5), (2, 2), (2, 2)),
    ('para.up_2', 128, 256, 1, 1, (5, 5), (1, 1), (2, 2)),
    ('para.up_1/end', 256, 512, 1, 1, (5, 5), (1, 1), (2, 2)),
    ('line.down_1', 256, 512, 1, 4, (5, 5), (2, 2), (2, 2)),
    ('line.down_2', 128, 256, 4, 4, (5, 5), (2, 2), (2, 2)),
    ('line.up_2', 128, 256, 4, 4, (5, 5), (1, 1), (2, 2)),
    ('line.up_1', 256, 512, 4, 4, (5, 5), (1, 1), (2, 2)),
    ('line.end', 256, 512, 4, 2, (5, 5), (1, 1), (2, 2)),
    ('char.conv_1', 32, 64, 1, 64, (5, 3), (2, 1), (0, 1)),
    ('char.conv_2', 14, 64, 64, 64, (5, 3), (2, 1), (0, 1)),
    ('char.conv_3', 5, 64, 64, 64, (5, 3), (2, 1), (0, 1)),
    ('wide 3x3 64->64', 256, 512, 64, 64, (3, 3), (1, 1), (1, 1)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--filter', default='')
    ap.add_argument('--option', action='append', default=[])
    ap.add_argument('--dtype', default='float32', help='activation storage (float16: pair rows only)')
    ap.add_argument('--height', type=int, default=256, help='page height of the pair rows')
    ap.add_argument('--width', type=int, default=512, help='page width of the pair rows')
    ap.add_argument('--pair-act', default='sigmoid', choices=['sigmoid', 'none'], help='output activation of the pair rows')
    args = ap.parse_args()
    from univer_ocr_amd.nn import CP, ops
    CP.use_gpu(0)
    CP.set_dtype(args.dtype)
    rt = CP.runtime()
    for opt in args.option:
        k, v = opt.split('=')
        rt.set_option(k, int(v))
    ev = [ctypes.c_void_p() for _ in range(2)]
    for e in ev:
        assert rt.lib.uocr_event_create(ctypes.byref(e)) == 0

    def timed(fn):
        # event pairs around chunks of 5 back-to-back launches, the fastest chunk counts: a host hiccup between two launches
        # (allocator, garbage collector: ~0.9 ms every dozen calls once earlier rows have filled the allocator's cache)
        # is not kernel time -- rocprofv3 shows the same kernel durations with and without it (tools/dev/t32_modes.sh)
        fn()
        rt.synchronize()
        per = min(5, args.reps)
        best = None
        for _ in range(max(1, args.reps // per)):
            rt.call('uocr_event_record', ev[0])
            for _ in range(per):
                fn()
            rt.call('uocr_event_record', ev[1])
            ms = ctypes.c_float()
            assert rt.lib.uocr_event_elapsed_ms_sync(ev[0], ev[1], ctypes.byref(ms)) == 0
            best = ms.value if best is None else min(best, ms.value)
        return best * 1e3 / per

    rng = np.random.default_rng(0)

    def pair_rows():
        # the fused Monochrome block (csrc/conv_pair.hip): FLOPs of the layer-by-layer algorithm
        # (no recompute counted): fwd 2 convs, bwd dw1 + dw2 + conv_2 dx (+ conv_1 dx)
        from univer_ocr_amd.hip import lib as hiplib
        n, h, w = args.batch, args.height, args.width
        x = CP.copy(rng.standard_normal((n, h, w, 1)).astype(np.float32))
        w1 = CP.copy((rng.standard_normal((3, 3, 1, 16)) * 0.3), np.float32)
        w2 = CP.copy((rng.standard_normal((3, 3, 16, 1)) * 0.1), np.float32)
        b1, b2 = CP.zeros((16,), np.float32), CP.zeros((1,), np.float32)
        g = CP.copy(rng.standard_normal((n, h, w, 1)).astype(np.float32))
        grads = [CP.zeros(a, np.float32) for a in (w1.shape, (16,), w2.shape, (1,))]
        sig = hiplib.ACT_SIGMOID if args.pair_act == 'sigmoid' else hiplib.ACT_NONE
        y = ops.conv_pair_fwd(x, w1, b1, w2, b2, act2=sig)
        px, conv = n * h * w, 2.0 * 9 * 16
        for op, fn, nbytes, flop in [
                ('fwd', lambda: ops.conv_pair_fwd(x, w1, b1, w2, b2, act2=sig), 8 * px, 2 * conv * px),
                ('bwd+dx', lambda: ops.conv_pair_bwd(x, y, g, w1, b1, w2, *grads, act2=sig), 16 * px, 4 * conv * px),
                ('bwd', lambda: ops.conv_pair_bwd(x, y, g, w1, b1, w2, *grads, act2=sig, need_dx=False), 12 * px,
                 3 * conv * px)]:
            us = timed(fn)
            print(f'{"mono.pair (fused)":18s} {op:6s} {us:9.1f} {nbytes / 1e6:8.1f} {nbytes / us / 1e3:8.0f} '
                  f'{flop / 1e9:8.2f} {flop / us / 1e6:7.2f}')

    def up_rows():
        # Upsample2D(2) + conv5x5 4->4 on the low-res tensor (csrc/conv_up.hip); FLOPs of the 5x5 on the
        # upsampled tensor (what the two-layer path computes), bytes = low-res tensor + high-res tensor
        for name, hl, wl, ch in (('line.up_2 (fused)', 64, 128, 4), ('line.up_1 (fused)', 128, 256, 4),
                                 ('para.up_2 (fused)', 64, 128, 1), ('para.up_1 (fused)', 128, 256, 1)):
            if args.filter not in name:
                continue
            n = args.batch
            xl = CP.copy(rng.standard_normal((n, hl, wl, ch)).astype(np.float32))
            wt = CP.copy((rng.standard_normal((5, 5, ch, ch)) * 0.1).astype(np.float32))
            b = CP.zeros((ch,))
            g = CP.copy(rng.standard_normal((n, 2 * hl, 2 * wl, ch)).astype(np.float32))
            dw, db = CP.zeros(wt.shape), CP.zeros((ch,))
            nbytes = 4 * n * hl * wl * ch * 5
            flop = 2.0 * n * 4 * hl * wl * 25 * ch * ch
            for op, fn in [('fwd', lambda: ops.upconv2x_fwd(xl, wt, b, (2, 2), True, 'leaky', 0.01)),
                           ('dgrad', lambda: ops.upconv2x_bwd_data(g, wt, xl.shape, (2, 2))),
                           ('wgrad', lambda: ops.upconv2x_bwd_weight(xl, g, dw, db, (2, 2)))]:
                us = timed(fn)
                print(f'{name:18s} {op:6s} {us:9.1f} {nbytes / 1e6:8.1f} {nbytes / us / 1e3:8.0f} {flop / 1e9:8.2f} '
                      f'{flop / us / 1e6:7.2f}')

    print(f'{"layer":18s} {"op":6s} {"us":>9s} {"MB":>8s} {"GB/s":>8s} {"GFLOP":>8s} {"TF/s":>7s}')
    if args.filter in 'mono.pair (fused)':
        pair_rows()
    if any(args.filter in name for name in ('line.up_1 (fused)', 'line.up_2 (fused)', 'para.up_1 (fused)', 'para.up_2 (fused)')):
        up_rows()
    for name, h, w, cin, cout, ks, st, pd in LAYERS:
        if args.filter not in name:
            continue
        n = args.batch
        x = CP.copy(rng.standard_normal((n, h, w, cin)).astype(np.float32))
        wt = CP.copy((rng.standard_normal((*ks, cin, cout)) * 0.1).astype(np.float32))
        b = CP.copy(rng.standard_normal(cout).astype(np.float32))
        y = ops.conv2d_fwd(x, wt, b, st, pd)
        oh, ow = y.shape[1], y.shape[2]
        g = CP.copy(rng.standard_normal(y.shape).astype(np.float32))
        dw, db = CP.zeros(wt.shape), CP.zeros(b.shape)
        in_b, out_b, w_b = 4 * n * h * w * cin, 4 * n * oh * ow * cout, 4 * wt.size
        flop = 2.0 * n * oh * ow * cout * ks[0] * ks[1] * cin
        runs = [('fwd', lambda: ops.conv2d_fwd(x, wt, b, st, pd), in_b + out_b + w_b),
                ('dgrad', lambda: ops.conv2d_bwd_data(g, wt, x.shape, st, pd), in_b + out_b + w_b),
                ('wgrad', lambda: ops.conv2d_bwd_weight(x, g, dw, db, st, pd), in_b + out_b + w_b)]
        for op, fn, nbytes in runs:
            us = timed(fn)
            print(f'{name:18s} {op:6s} {us:9.1f} {nbytes / 1e6:8.1f} {nbytes / us / 1e3:8.0f} {flop / 1e9:8.2f} '
                  f'{flop / us / 1e6:7.2f}')
        del x, y, g


if __name__ == '__main__':
    main()
