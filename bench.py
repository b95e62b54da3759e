#!/usr/bin/env python3
"""Benchmark of the hot path: document-images/sec (fwd+bwd+optimizer) on synthetic pages.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--config train-b32 (default; BASELINE.json configs[2], the configuration the metric is quoted on):
    one "step" = one train step of every my_model net on one device-resident synthetic batch: Monochrome,
    Paragraph and Line on (32,256,512,1) pages, Char on (32,32,64,1) line strips; forward, loss, backward, L2,
    SGD update for each, float32.
--config highres-fp16 (configs[4]): Monochrome, Paragraph and Line on (8,1024,2048,1) pages per GPU, binary16
    activations in HBM (UOCR_F16: float32 master weights, float32 accumulation, scaled activation gradients).
--config infer-b8 (configs[1]): forward only, 8 pages 256x512 + 8 line strips per GPU, float32.

Data parallel over N GPUs = N x the per-GPU batch per step (weak scaling), the gradient all-reduce through the
C ABI's RCCL entry points (univer_hip.h: uocr_dp_*); torch.distributed (gloo) is the host-side control plane
(rendezvous of the RCCL id, barriers, max-over-ranks of the elapsed time).  Rank 0 prints ONE JSON line with the
whole-job images/s, the roofline of the dominant kernel (HIP events around each of its launches in the timed
region + the same launch replayed alone), `roofline.secondary` (the wide 3x3 64->64 conv against the f32 MFMA
peak and the pooling / activation / loss kernels against HBM, measured right after the timed loop) and the CPU
baseline (oracle/ restatement of the reference timed on this box's host cores, same initial weights).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
F32_PEAK_TFLOPS = 157.3        # dense f32 MFMA peak = f32 vector peak on gfx950 (same guide, peak table)

CONFIGS = {
    'train-b32': dict(batch=32, height=256, width=512, dtype='float32', train=True,
                      nets=('Monochrome', 'Paragraph', 'Line', 'Char'), baseline='configs[2]'),
    'highres-fp16': dict(batch=8, height=1024, width=2048, dtype='float16', train=True,
                         nets=('Monochrome', 'Paragraph', 'Line'), baseline='configs[4]'),
    'highres-f32': dict(batch=8, height=1024, width=2048, dtype='float32', train=True,          # (diagnostic: what
                        nets=('Monochrome', 'Paragraph', 'Line'), baseline='configs[4] in float32'),   # float16 buys)
    'infer-b8': dict(batch=8, height=256, width=512, dtype='float32', train=False,
                     nets=('Monochrome', 'Paragraph', 'Line', 'Char'), baseline='configs[1]'),
}
FEEDS = {'Monochrome': ('image', 'monochrome'), 'Paragraph': ('monochrome', 'paragraph'),
         'Line': ('monochrome', 'line'), 'Char': ('char_lines', 'char_labels')}


class EventTimer:
    """HIP events of the C ABI on the stream of the runtime's CURRENT ctx."""

    def __init__(self, rt):
        self.rt, self.pool = rt, []

    def event(self):
        if self.pool:
            return self.pool.pop()
        ev = ctypes.c_void_p()
        assert self.rt.lib.uocr_event_create(ctypes.byref(ev)) == 0
        return ev

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        assert self.rt.lib.uocr_event_elapsed_ms_sync(a, b, ctypes.byref(ms)) == 0
        return ms.value

    def time_us(self, fn, reps=10):
        """mean duration of fn() over `reps` back-to-back calls on an otherwise idle GPU"""
        fn()
        self.rt.call('uocr_stream_sync')
        a, b = self.event(), self.event()
        self.rt.call('uocr_event_record', a)
        for _ in range(reps):
            fn()
        self.rt.call('uocr_event_record', b)
        us = self.elapsed_ms(a, b) * 1e3 / reps
        self.pool += [a, b]
        return us


    def time_us_rotating(self, fns, reps=12):
        """mean duration of one call when the calls cycle through `fns` (the same kernel on DIFFERENT buffer sets whose
        total footprint exceeds the 256 MiB Infinity Cache: every launch reads from HBM, not from the cache the
        previous launch of the same buffers left behind)"""
        for fn in fns:
            fn()
        self.rt.call('uocr_stream_sync')
        a, b = self.event(), self.event()
        self.rt.call('uocr_event_record', a)
        for i in range(reps):
            fns[i % len(fns)]()
        self.rt.call('uocr_event_record', b)
        us = self.elapsed_ms(a, b) * 1e3 / reps
        self.pool += [a, b]
        return us


class KernelProbe:
    """HIP-event pairs around every launch of one C-ABI entry point; events are recorded on the stream the
    kernel is launched on (the ctx stream of its lane)."""

    def __init__(self, runtime, name, timer):
        self.rt, self.name, self.timer = runtime, name, timer
        self.pairs = []
        self.enabled = False
        self.last_args = None
        self._orig = runtime.call
        runtime.call = self._call

    def _call(self, name, *args):
        if self.enabled and name == self.name:
            self.last_args = args
            a, b = self.timer.event(), self.timer.event()
            self._orig('uocr_event_record', a)
            self._orig(name, *args)
            self._orig('uocr_event_record', b)
            self.pairs.append((a, b))
        else:
            if name == self.name:
                self.last_args = args
            self._orig(name, *args)

    def solo_us(self, reps=10):
        """The same launch (same buffers) replayed alone on an otherwise idle GPU."""
        if self.last_args is None:
            return None
        args = self.last_args
        return self.timer.time_us(lambda: self._orig(self.name, *args), reps)

    def mean_us(self):
        total = sum(self.timer.elapsed_ms(a, b) for a, b in self.pairs)
        return 1e3 * total / max(1, len(self.pairs)), len(self.pairs)


def flat_weights(models):
    """{net: {'<layer>/<param>': float64 array}} of the trainer's CURRENT weights (for the CPU baseline)."""
    import numpy as np
    out = {}
    for name, model in models.items():
        out[name] = {pn: np.asarray(p.value.numpy(), dtype=np.float64) for pn, p in model.params().items()}
    return out


def cpu_baseline(cfg, args, weights, budget_s=25.0):
    """The oracle (float64 NumPy restatement of the reference) on a bounded sample of the SAME workload with the
    SAME initial weights as the GPU run: train steps of the configuration's nets on `sample` pages, host cores
    of this box; and the reference's own cost model (one Python iteration per output pixel,
    nn/layers/convolutional.py:90-96,121-134) on one page of the Monochrome net."""
    import numpy as np

    from oracle import nn_oracle as O
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    train = cfg['train']
    sample = cpu_sample_pages(cfg)
    data = make_page_batch(sample, cfg['height'], cfg['width'], args.char_width, seed=1234)
    warm = make_page_batch(1, 64, 128, args.char_width, seed=1)
    nets = {name: O.make_net(name, {k: v.copy() for k, v in weights[name].items()}) for name in cfg['nets']}

    def opt():
        return O.MomentumState(args.lr, 0.0) if args.optimizer == 'sgd' else O.AdamState(args.lr)
    opts = {n: opt() for n in nets}

    def one_step(batch, record=None):
        for name, net in nets.items():
            x, y = FEEDS[name]
            if train:
                losses, _ = net.train_step(batch[x], batch[y], opts[name])
                if record is not None:
                    record.setdefault(name, []).append(float(losses['output_losses'][0]))
            else:
                net.forward(batch[x])
    saved = {n: {k: v.copy() for k, v in net.params.items()} for n, net in nets.items()}
    one_step(warm)                               # page-in, BLAS thread pools (on a tiny page)
    for n, net in nets.items():                  # the warm-up step must not change the timed run's weights
        net.params = saved[n]
    opts = {n: opt() for n in nets}
    losses = {}
    steps, t0 = 0, time.perf_counter()
    while True:
        one_step(data, losses)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 3:
            break
    finite = all(np.isfinite(v).all() for v in losses.values())
    assert finite, f'CPU baseline diverged (losses {losses}): it no longer measures the GPU run\'s arithmetic'
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:   # noqa: BLE001
        threads = os.cpu_count() or 1
    what = 'train steps' if train else 'forward passes'
    out = {'value': round(sample * steps / el, 3), 'unit': 'images/s', 'cores': int(threads), 'blas_threads': int(threads),
           'host_cpus': os.cpu_count(), 'kind': 'port',
           'sample': f'{steps} {what} of {"+".join(cfg["nets"])} on {sample} pages {cfg["height"]}x{cfg["width"]}'
                     f'{" + %d line strips 32x%d" % (sample, args.char_width) if "Char" in cfg["nets"] else ""}, float64 '
                     f'NumPy oracle (im2col + BLAS), the GPU run\'s initial weights, {el:.1f} s',
           'first_losses': {n: round(v[0], 6) for n, v in losses.items()} if train else None}
    # the reference's cost model: a Python loop over output pixels (what its NumPy mode executes)
    name = 'Monochrome'
    page = make_page_batch(1, 256, 512, args.char_width, seed=1234)
    loop_net = O.make_net(name, {k: v.copy() for k, v in weights[name].items()}, loops=True)
    t0 = time.perf_counter()
    if train:
        loop_net.train_step(page['image'], page['monochrome'], opt())
    else:
        loop_net.forward(page['image'])
    el = time.perf_counter() - t0
    out['reference_cost_model'] = {
        'value': round(1.0 / el, 4), 'unit': 'images/s', 'cores': 1, 'kind': 'reference-cost-model',
        'sample': f'1 {"train step" if train else "forward"} of the Monochrome net on one 256x512 page with the '
                  f'convolutions as one Python iteration per output pixel (nn/layers/convolutional.py:90-96, '
                  f'121-134: 131 072 iterations of reshape + concatenate + dot per conv and direction), {el:.1f} s'}
    return out


def gpu_first_losses(cfg, args, weights, sample):
    """The first train step of the configuration's nets on the CPU baseline's sample pages (same seed, same initial
    weights) through the production path on the GPU: ties the timed arithmetic to the oracle's `first_losses`."""
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    data = make_page_batch(sample, cfg['height'], cfg['width'], args.char_width, seed=1234)
    small = PageTrainer(sample, cfg['height'], cfg['width'], args.char_width, args.optimizer, args.lr, seed=0,
                        nets=cfg['nets'], graphs=False, pipelined=False, input_grads=not args.skip_input_grads)
    for name, model in small.models.items():
        nested = {}
        for key, value in weights[name].items():
            layer, pname = key.rsplit('/', 1)
            nested.setdefault(layer, {})[pname] = value.tolist()
        model.set_weights(nested)
    losses = small.step(small.make_context(data))
    return {name: float(l['output_losses'][0]) for name, l in losses.items()}


def parse_xcds(spec):
    """'0-2,3-5,6-7' -> ((0, 1, 2), (3, 4, 5), (6, 7))"""
    groups = []
    for part in spec.split(','):
        lo, _, hi = part.partition('-')
        groups.append(tuple(range(int(lo), int(hi or lo) + 1)))
    return tuple(groups)


def cpu_sample_pages(cfg):
    return 8 if cfg['height'] * cfg['width'] <= 256 * 512 else 1


def secondary_rooflines(rt, timer, watchdog):
    """Kernels north_star prices besides the dominant one, measured alone right after the timed loop:
    the wide Conv2D (3x3, 64 -> 64, batch 32, 256x512: the shape of the ">= 50 % of MFMA peak on Conv2D at batch
    32" target) against the f32 MFMA peak, pooling / activation / loss kernels against the HBM peak.
    achieved = algorithmic flops or bytes / mean launch time."""
    import numpy as np

    from univer_ocr_amd.nn import CP, ops
    CP.set_dtype('float32')
    rng = np.random.default_rng(0)
    out = []
    n, h, w, c = 32, 256, 512, 64

    def rand(shape):
        # one image of random data from the host, replicated on the device (1 GB tensors: keep host memory small)
        t = CP.empty(shape, np.float32)
        slab = CP.copy(rng.standard_normal((1,) + tuple(shape[1:])).astype(np.float32))
        for i in range(shape[0]):
            t.t[i].copy_(slab.t[0], non_blocking=True)
        return t
    x = rand((n, h, w, c))
    wt = CP.copy((rng.standard_normal((3, 3, c, c)) * 0.05).astype(np.float32))
    b = CP.copy(rng.standard_normal(c).astype(np.float32))
    y = ops.conv2d_fwd(x, wt, b, (1, 1), (1, 1))
    dw, db = CP.zeros(wt.shape), CP.zeros(b.shape)
    flop = 2.0 * n * h * w * c * 9 * c
    for label, fn in (('fwd', lambda: ops.conv2d_fwd(x, wt, b, (1, 1), (1, 1))),
                      ('dx', lambda: ops.conv2d_bwd_data(y, wt, x.shape, (1, 1), (1, 1))),
                      ('dw', lambda: ops.conv2d_bwd_weight(x, y, dw, db, (1, 1), (1, 1)))):
        us = timer.time_us(fn, 5)
        tf = flop / us / 1e6
        out.append({'kernel': f'Conv2D 3x3 64->64 {label}, batch 32, 256x512 (f32 MFMA implicit GEMM)', 'bound': 'mfma',
                    'achieved': round(tf, 1), 'peak': F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(tf / F32_PEAK_TFLOPS, 3), 'launch_us': round(us, 1)})
        watchdog.beat('secondary rooflines')
    del x, y
    # HBM-bound kernels: SETS distinct buffer sets per kernel (>= 6 x 134 MB per tensor: far beyond the 256 MiB Infinity
    # Cache), launches cycle through them, so every launch streams its bytes from HBM (round 2 timed back-to-back
    # launches on ONE set and read cache-assisted rates above what HBM delivers)
    SETS = 6
    mb = lambda *arrs: sum(a.nbytes for a in arrs)   # noqa: E731
    lazy, CP.lazy_losses = CP.lazy_losses, True
    sets = []
    for _ in range(SETS):
        x4, g4, lo4 = rand((n, h, w, 4)), rand((n, h, w, 4)), rand((n, h // 2, w // 2, 4))
        y4, mask = ops.maxpool2d_fwd(x4, (2, 2), (2, 2), (0, 0))
        gy4 = rand(y4.shape)
        p1 = CP.copy(rng.random((n, h, w, 1)).astype(np.float32))
        t1 = CP.copy((rng.random((n, h, w, 1)) > 0.5).astype(np.float32))
        sets.append(dict(x4=x4, g4=g4, lo4=lo4, y4=y4, mask=mask, gy4=gy4, p1=p1, t1=t1))
    logits = CP.copy(rng.standard_normal((2048, 162)).astype(np.float32))
    onehot = CP.copy(np.eye(162, dtype=np.float32)[rng.integers(0, 162, 2048)])
    s0 = sets[0]
    rows = (
        ('MaxPool2D 2x2 fwd (x -> y + u8 mask)', lambda s: ops.maxpool2d_fwd(s['x4'], (2, 2), (2, 2), (0, 0)),
         mb(s0['x4'], s0['y4'], s0['mask'])),
        ('MaxPool2D 2x2 bwd (dy, mask -> dx)', lambda s: ops.maxpool2d_bwd(s['gy4'], s['mask'], s['x4'].shape, (2, 2), (2, 2), (0, 0)),
         mb(s0['gy4'], s0['mask'], s0['x4'])),
        ('Relu fwd', lambda s: ops.act_fwd('relu', s['x4']), mb(s0['x4'], s0['x4'])),
        ('LeakyRelu bwd from output', lambda s: ops.act_bwd_from_output('leaky', s['x4'], s['g4'], 0.01), mb(s0['x4'], s0['g4'], s0['x4'])),
        ('Upsample2D 2x fwd (4 ch)', lambda s: ops.upsample2d_fwd(s['lo4'], (2, 2)), mb(s0['lo4'], s0['x4'])),
        ('Dice loss + grad (1 ch, output Sigmoid folded)', lambda s: ops.seg_loss('dice', s['p1'], s['t1'], True, out_act='sigmoid'),
         mb(s0['p1'], s0['t1']) + mb(s0['t1'], s0['p1'], s0['p1'])),
    )
    for label, fn, nbytes in rows:
        us = timer.time_us_rotating([lambda s=s_: fn(s) for s_ in sets], 2 * SETS)
        gbs = nbytes / us / 1e3
        out.append({'kernel': label, 'bound': 'hbm', 'achieved': round(gbs, 0), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(gbs / HBM_PEAK_GBS, 3), 'launch_us': round(us, 1), 'buffer_sets': SETS})
        watchdog.beat('secondary rooflines')
    us = timer.time_us(lambda: ops.softmax_ce(logits, onehot, True), 10)
    nbytes = mb(logits, onehot, logits)
    out.append({'kernel': 'SoftmaxCE + grad (2048 x 162; 4 MB: launch-latency-bound, cache-resident)', 'bound': 'hbm',
                'achieved': round(nbytes / us / 1e3, 0), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(nbytes / us / 1e3 / HBM_PEAK_GBS, 3), 'launch_us': round(us, 1), 'buffer_sets': 1})
    CP.lazy_losses = lazy
    return out


def secondary_rooflines_f16(rt, timer, watchdog, batch, height, width):
    """highres-fp16: the Line net's full-resolution layers (binary16 storage, binary16 MFMAs: csrc/conv_h16.hip,
    conv_h16w.hip) alone, against the HBM peak.  Algorithmic bytes = every tensor the op reads or writes once, 2 bytes
    per activation element."""
    import numpy as np

    from univer_ocr_amd.nn import CP, ops
    CP.set_dtype('float16')
    rng = np.random.default_rng(0)
    n, h, w = batch, height, width

    def act(shape):
        t = CP.empty(shape, np.float16)
        slab = CP.copy(rng.standard_normal((1,) + tuple(shape[1:])).astype(np.float32))
        for i in range(shape[0]):
            t.t[i].copy_(slab.t[0], non_blocking=True)
        return t

    def par(shape, s=0.2):
        return CP.copy(rng.standard_normal(shape) * s, np.float32)
    SETS = 4             # 4 x (134 + 67 + 134 + 34 MB): every launch streams from HBM, not from the Infinity Cache
    sets = [dict(x4=act((n, h, w, 4)), g2=act((n, h, w, 2)), g4=act((n, h, w, 4)), xl=act((n, h // 2, w // 2, 4)))
            for _ in range(SETS)]
    w42, b2, w44, b4 = par((5, 5, 4, 2)), par((2,)), par((5, 5, 4, 4)), par((4,))
    dw42, db2 = CP.zeros((5, 5, 4, 2), np.float32), CP.zeros((2,), np.float32)
    dw44, db4 = CP.zeros((5, 5, 4, 4), np.float32), CP.zeros((4,), np.float32)
    px = n * h * w
    rows = (
        ('Line end conv 5x5 4->2 + Sigmoid fwd (f16 MFMA, Toeplitz rows)',
         lambda s: ops.conv2d_fwd(s['x4'], w42, b2, (1, 1), (2, 2), 0.0, True, act='sigmoid'), 12 * px),
        ('Line end conv dx + LeakyReLU mask', lambda s: ops.conv2d_bwd_data(s['g2'], w42, s['x4'].shape, (1, 1), (2, 2), x_act=s['x4'],
                                                                         act='leaky', alpha=0.01), 20 * px),
        ('Line end conv dw', lambda s: ops.conv2d_bwd_weight(s['x4'], s['g2'], dw42, db2, (1, 1), (2, 2), 0.0, True, accumulate=False),
         12 * px),
        ('Line up_1 (upsample 2x + conv 5x5 4->4 + LeakyReLU) fwd',
         lambda s: ops.upconv2x_fwd(s['xl'], w44, b4, (2, 2), True, act='leaky', alpha=0.01), 10 * px),
        ('Line up_1 dx + LeakyReLU mask', lambda s: ops.upconv2x_bwd_data(s['g4'], w44, s['xl'].shape, (2, 2), x_act=s['xl'], act='leaky',
                                                                       alpha=0.01), 12 * px),
        ('Line up_1 dw', lambda s: ops.upconv2x_bwd_weight(s['xl'], s['g4'], dw44, db4, (2, 2), True, accumulate=False), 10 * px),
    )
    out = []
    for label, fn, nbytes in rows:
        us = timer.time_us_rotating([lambda s=s_: fn(s) for s_ in sets], 3 * SETS)
        gbs = nbytes / us / 1e3
        out.append({'kernel': f'{label}, {n} x {h} x {w}', 'bound': 'hbm', 'achieved': round(gbs, 0), 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 3), 'launch_us': round(us, 1), 'buffer_sets': SETS})
        watchdog.beat('secondary rooflines')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='train-b32', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=None, help='pages per GPU per step (default: the configuration\'s)')
    ap.add_argument('--height', type=int, default=None)
    ap.add_argument('--width', type=int, default=None)
    ap.add_argument('--char-width', type=int, default=64)
    ap.add_argument('--optimizer', default='sgd', choices=['sgd', 'adam'])
    ap.add_argument('--lr', type=float, default=0.0015)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip roofline.secondary (the kernels measured after the loop)')
    ap.add_argument('--no-overlap', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='end every step with the main stream waiting for all net streams (default: a net starts '
                         'its next step as soon as ITS previous step is done; the timed region still ends with a full '
                         'device synchronisation)')
    ap.add_argument('--dp-single-collective', action='store_true',
                    help='N > 1: ONE all-reduce per step over the flat gradient buffer of all nets (default: one per '
                         'net, issued in a fixed order from the communication lane)')
    ap.add_argument('--no-graphs', action='store_true', help='eager launches (default: HIP graphs)')
    ap.add_argument('--skip-input-grads', action='store_true',
                    help='DIAGNOSTIC: do not compute the gradient w.r.t. the page inputs (unused by training; the '
                         'reference computes it, and so does the default run)')
    ap.add_argument('--h2d', action='store_true',
                    help='upload every batch as uint8 from pinned host memory on a copy stream and convert on the '
                         'device (PCIe-inclusive rate; flagged in metric and config, never the headline value)')
    ap.add_argument('--lane-per-net', action='store_true', help='one lane per net instead of the balanced groups')
    ap.add_argument('--lane-groups', default=None,
                    help='experiment: nets per lane, e.g. "Monochrome+Char,Paragraph+Line" (default: Monochrome+Paragraph,Line,Char)')
    ap.add_argument('--lane-xcds', default=None,
                    help='EXPERIMENT: CU-partitioned lanes, XCDs per lane group, e.g. "0-2,3-5,6-7" (Monochrome+Paragraph | Line | Char)')
    ap.add_argument('--option', action='append', default=[],
                    help='kernel selection knob of the C ABI, key=value (uocr_ctx_set_option: split_blocks, split_min, '
                         'gemm_bm, mfma, tiled, xcd_remap); experiments only')
    ap.add_argument('--steady-steps', type=int, default=300,
                    help='N = 1: steps of the steady-state measurement after the timed region (0 = off)')
    ap.add_argument('--step-timeout', type=float, default=300.0,
                    help='watchdog: exit with code 3 when no step / phase completes for this many seconds')
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for key in ('batch', 'height', 'width'):
        if getattr(args, key) is not None:
            cfg[key] = getattr(args, key)
    train = cfg['train']

    # The contract is ONE line on stdout.  Libraries write there too (RCCL prints a version banner when the
    # communicator is created): everything but the final JSON line goes to stderr, at the descriptor level.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    # One hardware queue per stream: ROCm's default is 4 queues per process for main + 3 net lanes + copy /
    # communication streams, so two of them share a queue and run one after the other (1.25 -> 1.10 ms/step).
    # Read by the HIP runtime when it starts, hence set before torch touches the GPU.
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    # RCCL shares device buffers between the ranks' processes through dmabuf IPC; the host driver of this pool
    # supports no other mode (without it: hipIpcGetMemHandle: invalid argument)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('NCCL_DEBUG', 'WARN')       # RCCL's own warnings go to stderr next to this script's

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
        args.gpus = world
    from univer_ocr_amd.watchdog import Watchdog
    watchdog = Watchdog(args.step_timeout, rank, 'bench watchdog')
    graphs = not args.no_graphs
    # UOCR_BENCH_REHEARSAL=1: several ranks on ONE card, gradients through gloo staged over the host
    # (parallel.DataParallel(backend='gloo')) -- exercises this file's multi-rank path where RCCL would refuse
    # two ranks on one device; never a measurement
    rehearsal = os.environ.get('UOCR_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    # UOCR_BENCH_FORCE_DP=1: a ONE-rank RCCL communicator (C ABI) with the whole data-parallel machinery on
    # -- the closest a one-GPU box gets to the N > 1 data path; never a measurement either
    force_dp = world == 1 and os.environ.get('UOCR_BENCH_FORCE_DP') == '1'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('gloo', rank=rank, world_size=world)      # host-side control plane only
    watchdog.beat('rendezvous done')

    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP

    CP.use_gpu(local_rank)
    CP.set_dtype(cfg['dtype'])
    CP.lazy_losses = True                        # losses stay on the device; fetched after the loop
    rt = CP.runtime()
    timer = EventTimer(rt)

    use_dp = train and (world > 1 or force_dp)
    def build_trainer(dp_backend):
        return PageTrainer(cfg['batch'], cfg['height'], cfg['width'], args.char_width, args.optimizer, args.lr,
                           seed=0, nets=cfg['nets'], overlap=not args.no_overlap,
                           input_grads=not args.skip_input_grads, graphs=graphs,
                           pipelined=not args.no_pipeline, data_parallel=use_dp,
                           dp_coalesce=args.dp_single_collective, dp_backend=dp_backend,
                           **({'lane_groups': None} if args.lane_per_net else
                              {'lane_groups': tuple(tuple(g.split('+')) for g in args.lane_groups.split(','))}
                              if args.lane_groups else {}),
                           **({'lane_xcds': parse_xcds(args.lane_xcds)} if args.lane_xcds else {}))
    dp_fallback = None
    trainer, failure = None, None
    try:
        trainer = build_trainer('gloo' if rehearsal else None)
    except Exception as exc:       # noqa: BLE001
        if not (use_dp and world > 1 and not rehearsal):
            raise
        failure = f'{type(exc).__name__}: {exc}'
    if use_dp and world > 1 and not rehearsal:
        # The RCCL communicator may have failed on ONE rank only (its self-test, its memory ...): the ranks agree over the
        # gloo control plane before anybody takes the fallback branch -- a rank that went on alone would sit in RCCL
        # while the others sit in gloo.  A number with the gradients staged through the host and a LOUD flag beats no
        # number: config.grad_allreduce says so and the JSON line carries the error.
        ok = torch.tensor([0 if failure else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            reasons = [None] * world
            dist.all_gather_object(reasons, failure)
            dp_fallback = '; '.join(f'rank {i}: {r}' for i, r in enumerate(reasons) if r) or 'unknown'
            print(f'[bench] rank {rank}: RCCL data-parallel setup failed somewhere ({dp_fallback}); ALL ranks FALL BACK to '
                  f'gloo staging', file=sys.stderr, flush=True)
            if trainer is not None and trainer.dp is not None:
                trainer.dp.close()                   # uocr_dp_finalize: no half-open communicator stays behind
            trainer = None
            torch.cuda.synchronize()
            trainer = build_trainer('gloo')
    for opt in args.option:
        key, value = opt.split('=')
        rt.set_option(key, int(value))          # (every lane)
    watchdog.beat('trainer built')
    initial = flat_weights(trainer.models) if rank == 0 and not args.no_cpu_baseline and world == 1 else None
    first_gpu = None
    if initial is not None and train:
        # the production path on the CPU baseline's sample pages, before anything is timed (compared further down)
        first_gpu = gpu_first_losses(cfg, args, initial, cpu_sample_pages(cfg))
        watchdog.beat('first losses on the baseline sample')
    layers = make_page_batch(cfg['batch'], cfg['height'], cfg['width'], args.char_width, seed=1234 + rank)
    context = trainer.make_context(layers)       # inputs resident in HBM before the timed region
    watchdog.beat('inputs resident')

    # Dominant kernel = the single longest launch of the step in the rocprofv3 trace (profiles/): the fused
    # backward of the Monochrome block (csrc/conv_pair.hip, entry uocr_conv_pair_bwd).  float32: it recomputes the
    # 16-channel activation on chip, so it is bound by f32 multiply-adds, priced in FLOP/s against the f32 peak
    # (ALGORITHMIC flops = the layer-by-layer algorithm, recompute not counted: conv_2 dw, conv_2 dx, conv_1 dw
    # (+ conv_1 dx), each 2*9*16 per pixel).  float16 (configs[4]): the configuration is the HBM-bound regime
    # by definition, so the same launch is priced in algorithmic BYTES (x, y, dy read, dx written, 2 B each)
    # against the HBM peak.
    npix = cfg['batch'] * cfg['height'] * cfg['width']
    n_convs = 3 if args.skip_input_grads else 4
    esize = 2 if cfg['dtype'] == 'float16' else 4
    if train:
        probe = KernelProbe(rt, 'uocr_conv_pair_bwd', timer)
        # (the output Sigmoid' is applied by the Dice gradient kernel: the launch reads x and g = dLoss/d(conv_2 output) and
        # writes dx)
        dominant = {'kernel': 'uocr_conv_pair_bwd: fused backward of conv3x3(1->16)+LeakyReLU+conv3x3(16->1) (Monochrome; ' +
                              ('pair_strip_bwd_kernel + pair_strip_finish, csrc/conv_pair_strip.hip)' if cfg['dtype'] == 'float32'
                               else 'pair_wave_bwd_h_kernel + pair_strip_finish, csrc/conv_pair_strip_h.hip)'),
                    'flops': 2.0 * 9 * 16 * n_convs * npix, 'bytes': float(esize) * npix * (n_convs - 1)}
    else:
        probe = KernelProbe(rt, 'uocr_conv_pair_fwd', timer)
        dominant = {'kernel': 'uocr_conv_pair_fwd: fused forward of conv3x3(1->16)+LeakyReLU+conv3x3(16->1)+Sigmoid '
                              '(Monochrome)', 'flops': 2.0 * 9 * 16 * 2 * npix, 'bytes': float(esize) * npix * 2}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    feeder = None
    if args.h2d:       # uint8 batches, pinned double buffers, copy stream (my_model/pipeline.py)
        from univer_ocr_amd.my_model.pipeline import PageFeeder, to_uint8_layers
        layers_u8 = to_uint8_layers(layers)
        feeder = PageFeeder(layers_u8)
        feeder.stage(layers_u8)

    statics = {}

    def one_step():
        if not train:
            trainer.forward(context)
            return None
        if feeder is None:
            return trainer.step(context)
        if statics:                     # graph replay: convert straight into the arrays the graphs read
            trainer.join()              # ... once the lanes are done with the previous batch
            ctx = feeder.context(into=statics)
        else:
            ctx = feeder.context()      # batch i (uploaded while step i-1 ran)
        feeder.stage(layers_u8)         # start the upload of batch i+1
        return trainer.step(ctx)

    if graphs and train:
        try:
            trainer.capture(context)
        except Exception as exc:                   # keep measuring: eager launches are the same computation
            print(f'[bench] HIP graph capture failed ({type(exc).__name__}: {exc}); continuing with eager launches',
                  file=sys.stderr, flush=True)
            trainer.graphs, trainer._captured, graphs = False, None, False
            torch.cuda.synchronize()
        if feeder is not None and graphs:
            statics.update(trainer.static_inputs())
    watchdog.beat('graphs captured')
    losses = None
    for i in range(args.warmup):
        losses = one_step()
        watchdog.beat(f'warm-up step {i}')
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = one_step()
        watchdog.beat(f'timed step {i}')
    barrier()
    elapsed = time.perf_counter() - t0
    watchdog.beat('timed region done')
    # The dominant launch inside the loop: events cannot be recorded into a captured graph on this ROCm, so the timed
    # region above replays ALL nets from graphs, and here the probed kernel's net runs eagerly for a few more steps
    # (same arithmetic) with a HIP-event pair around each launch of the entry point, the other lanes as before
    if train:                                   # (every rank: the steps carry collectives)
        probed_net = 'Monochrome'
        if graphs:
            trainer.set_eager_nets((probed_net,))
        one_step()
        barrier()
        probe.enabled = True
        probe_steps = max(10, min(args.steps, 30))
        if trainer.dp is not None:
            trainer.dp.profile, dp_before = True, trainer.dp.collectives
        for i in range(probe_steps):
            one_step()
        barrier()
        probe.enabled = False
        dp_stats = None
        if trainer.dp is not None:
            trainer.dp.profile = False
            dp_stats = trainer.dp.stats() or {}
            dp_stats['collectives_per_step'] = round((trainer.dp.collectives - dp_before) / probe_steps, 2)
            dp_stats['backend'] = trainer.dp.backend
            ver = ctypes.c_int()
            if trainer.dp.backend == 'rccl' and rt.lib.uocr_dp_version(ctypes.byref(ver)) == 0:
                dp_stats['rccl_version'] = ver.value
        if graphs:
            trainer.set_eager_nets(())
        watchdog.beat('in-loop probe done')

    # steady state: the driver's run is short (20 steps = 17 ms), so the same loop once more over >= 300 steps in
    # chunks of 10 with a full synchronisation after each; the median chunk is robust against one slow chunk
    steady = None
    if world == 1 and args.steady_steps > 0:
        chunk, times = 10, []
        for c in range(max(1, args.steady_steps // chunk)):
            barrier()
            c0 = time.perf_counter()
            for _ in range(chunk):
                losses = one_step()
            barrier()
            times.append((time.perf_counter() - c0) / chunk)
            watchdog.beat(f'steady-state chunk {c}')
        times.sort()
        med = times[len(times) // 2]
        steady = {'steps': chunk * len(times), 'chunk_steps': chunk,
                  'ms_per_step_median': round(1e3 * med, 4), 'ms_per_step_mean': round(1e3 * sum(times) / len(times), 4),
                  'ms_per_step_min': round(1e3 * times[0], 4),
                  'images_per_s': round(cfg['batch'] / med, 2),
                  'note': 'measured right after the timed region; every chunk of 10 steps ends with a device synchronisation'}

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final = {n: [float(v) for v in l['output_losses']] for n, l in losses.items()} if losses else None

    if rank == 0:
        kernel_us, launches = probe.mean_us()
        solo_us = probe.solo_us()
        watchdog.beat('solo replay done')

        def priced(us):
            if not us:
                return None
            if cfg['dtype'] == 'float16':
                return dominant['bytes'] / us / 1e3          # GB/s
            return dominant['flops'] / us / 1e6              # TFLOP/s
        hbm_bound = cfg['dtype'] == 'float16'
        peak = HBM_PEAK_GBS if hbm_bound else F32_PEAK_TFLOPS
        in_loop, solo = priced(kernel_us), priced(solo_us)
        # `achieved` / `frac`: the launch ALONE on the GPU (the kernel's own roofline; this is what the rocprofv3
        # per-kernel average under profiles/ corresponds to).  In the timed region the nets run on three streams
        # and share the machine: that figure is reported next to it.
        headline = solo if solo else in_loop
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, 'profiles', 'dominant_kernel_traffic.json')
        if os.path.exists(tpath) and args.config == 'train-b32' and cfg['batch'] == 32 and not args.skip_input_grads:
            with open(tpath) as f:
                rec = json.load(f)
            traffic = rec.get('hbm_bytes_per_launch')
            traffic_source = ('NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this '
                              'command, committed as profiles/dominant_kernel_traffic.json (' +
                              str(rec.get('source', 'see profiles/README.md')) + ')')
        if dp_fallback:
            allreduce_desc = f'FALLBACK: gloo through the host, RCCL setup failed: {dp_fallback}'
        elif world > 1 and rehearsal:
            allreduce_desc = 'gloo (REHEARSAL on one card)'
        elif world > 1:
            allreduce_desc = 'rccl through the C ABI (uocr_dp_allreduce_sum), ' + (
                'ONE collective per step' if args.dp_single_collective else 'one collective per net, fixed order')
        else:
            allreduce_desc = 'rccl through the C ABI, ONE rank (REHEARSAL)' if force_dp else None
        images = cfg['batch'] * world * args.steps
        what = 'fwd+bwd' if train else 'fwd only'
        out = {
            'metric': f'document-images/sec ({what}) on {cfg["height"]}x{cfg["width"]} synthetic pages' +
                      (' [DIAGNOSTIC: PCIe upload of every batch inside the timed region]' if args.h2d else '') +
                      (' [DIAGNOSTIC: page-input gradients not computed]' if args.skip_input_grads else ''),
            'value': round(images / elapsed, 2),
            'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'float32': 'f32', 'float16': 'f16 storage / f32 accumulate', 'float64': 'f64'}[cfg['dtype']],
            'data': 'synthetic',
            'config': {
                'workload': f'BASELINE {cfg["baseline"]}: my_model ' +
                            (f'train step (fwd+loss+bwd+L2+{args.optimizer.upper()})' if train else 'forward pass') +
                            f' of {"+".join(n for n in cfg["nets"] if n != "Char")} on '
                            f'({cfg["batch"]},{cfg["height"]},{cfg["width"]},1) pages' +
                            (f' and Char on ({cfg["batch"]},32,{args.char_width},1) line strips' if 'Char' in cfg['nets'] else '') +
                            ' per GPU',
                'name': args.config,
                'batch_per_gpu': cfg['batch'], 'global_batch': cfg['batch'] * world,
                'page': [cfg['height'], cfg['width']], 'optimizer': args.optimizer if train else None,
                'parallelism': f'dp{world}',
                'grad_allreduce': allreduce_desc,
                'final_losses': final,
                'h2d_inclusive': bool(args.h2d), 'input_grads': not args.skip_input_grads, 'hip_graphs': bool(graphs),
                'pipelined_lanes': not args.no_pipeline,
                'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'default (4)'),
            },
            'roofline': {'bound': 'hbm' if hbm_bound else 'mfma', 'kernel': dominant['kernel'],
                         'achieved': None if headline is None else round(headline, 2),
                         'peak': peak, 'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
                         'frac': None if headline is None else round(headline / peak, 4),
                         'traffic': traffic, 'traffic_source': traffic_source,
                         'solo_launch_us': None if solo_us is None else round(solo_us, 2),
                         'in_loop_launch_us': round(kernel_us, 2) if launches else None, 'in_loop_achieved': None if in_loop is None else round(in_loop, 2),
                         'launches_timed': launches,
                         'algorithmic_flops_per_launch': dominant['flops'],
                         'algorithmic_bytes_per_launch': dominant['bytes']},
        }
        if world == 1 and not args.no_secondary:
            del context, layers
            trainer.join()
            torch.cuda.synchronize()
            if cfg['dtype'] == 'float16':
                out['roofline']['secondary'] = secondary_rooflines_f16(rt, timer, watchdog, cfg['batch'], cfg['height'],
                                                                       cfg['width'])
            else:
                out['roofline']['secondary'] = secondary_rooflines(rt, timer, watchdog)
        if steady is not None:
            out['steady_state'] = steady
        if train and dp_stats is not None:
            out['dp'] = dp_stats       # (rank 0's view: per-collective device times of the probe steps)
        if world == 1 and not args.no_cpu_baseline:
            watchdog.limit = max(watchdog.limit, 600.0)      # host-only phase: no collective can hang here
            out['cpu_baseline'] = cpu_baseline(cfg, args, initial)
            if first_gpu is not None and out['cpu_baseline'].get('first_losses'):
                ref = out['cpu_baseline']['first_losses']
                rel = {n: abs(first_gpu[n] - ref[n]) / max(1e-30, abs(ref[n])) for n in ref}
                out['cpu_baseline']['gpu_first_losses'] = {n: round(v, 6) for n, v in first_gpu.items()}
                out['cpu_baseline']['gpu_vs_cpu_first_loss_rel_diff'] = {n: float(f'{v:.3e}') for n, v in rel.items()}
                tol = 1e-4 if cfg['dtype'] == 'float32' else 5e-3
                if max(rel.values()) > tol:
                    print(f'[bench] the GPU path and the CPU oracle disagree on the first train step of the baseline sample: '
                          f'{rel} (tolerance {tol}); the timed arithmetic is not the reference\'s -- no result line',
                          file=sys.stderr, flush=True)
                    os._exit(4)
        import math
        if final and not all(math.isfinite(v) for vs in final.values() for v in vs):
            out['invalid'] = f'non-finite final losses {final}: the run diverged, the throughput value measures nothing'
            print('[bench] ' + out['invalid'], file=sys.stderr, flush=True)
        print(json.dumps(out), file=json_out, flush=True)
        if out.get('invalid'):
            os._exit(5)
    if trainer.dp is not None:
        watchdog.beat('closing the communicator')
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        trainer.dp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
