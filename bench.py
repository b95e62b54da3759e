#!/usr/bin/env python3
"""Benchmark of the hot path: document-images/sec (fwd+bwd+optimizer) on 256x512 synthetic pages.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one train step of every my_model net on one device-resident synthetic batch
(BASELINE.json configs[2]: fwd+bwd+SGD, batch 32 per GPU, 256x512 pages): Monochrome, Paragraph and
Line on (32,256,512,1) pages, Char on (32,32,64,1) line strips; forward, loss, backward, L2,
optimizer update for each.  Data parallel over N GPUs = N x 32 pages per step (weak scaling), one
RCCL all-reduce per net per step.  Rank 0 prints ONE JSON line with the whole-job images/s, the
roofline of the dominant kernel (timed live with HIP events around each of its launches in the
timed region) and the CPU baseline (oracle/ restatement timed on this box's host cores).
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


class KernelProbe:
    """HIP-event pairs around every launch of one C-ABI entry point whose int arguments match;
    events are recorded on the stream the kernel is launched on (the ctx stream)."""

    def __init__(self, runtime, name, match):
        self.rt, self.name, self.match = runtime, name, match
        self.pairs, self.pool = [], []
        self.enabled = False
        self._orig = runtime.call
        runtime.call = self._call

    def _event(self):
        if self.pool:
            return self.pool.pop()
        ev = ctypes.c_void_p()
        assert self.rt.lib.uocr_event_create(ctypes.byref(ev)) == 0
        return ev

    def _call(self, name, *args):
        if self.enabled and name == self.name and self.match(args):
            self.last_args = args
            a, b = self._event(), self._event()
            self._orig('uocr_event_record', a)
            self._orig(name, *args)
            self._orig('uocr_event_record', b)
            self.pairs.append((a, b))
        else:
            self._orig(name, *args)

    def solo_ms(self, reps=10):
        """The same launch (same buffers) replayed alone on an otherwise idle GPU."""
        if getattr(self, 'last_args', None) is None:
            return None
        self._orig('uocr_stream_sync')
        a, b = self._event(), self._event()
        self._orig(self.name, *self.last_args)
        self._orig('uocr_event_record', a)
        for _ in range(reps):
            self._orig(self.name, *self.last_args)
        self._orig('uocr_event_record', b)
        ms = ctypes.c_float()
        assert self.rt.lib.uocr_event_elapsed_ms_sync(a, b, ctypes.byref(ms)) == 0
        return ms.value / reps

    def mean_ms(self):
        total, ms = 0.0, ctypes.c_float()
        for a, b in self.pairs:
            assert self.rt.lib.uocr_event_elapsed_ms_sync(a, b, ctypes.byref(ms)) == 0
            total += ms.value
        return total / max(1, len(self.pairs)), len(self.pairs)


def cpu_baseline(height, width, char_width, optimizer, lr, budget_s=12.0):
    """The oracle (NumPy restatement of the reference, float64) on a bounded sample of the SAME
    workload: train steps of the four nets on 2 pages + 2 line strips, host cores of this box."""
    import numpy as np

    from oracle import nn_oracle as O
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    sample = 2
    data = make_page_batch(sample, height, width, char_width, seed=1234)
    rng = np.random.default_rng(0)
    nets = {}
    for name in ('Monochrome', 'Paragraph', 'Line', 'Char'):
        spec, _ = O.NET_SPECS[name]()
        nets[name] = O.make_net(name, O.kaiming_uniform_weights(spec, rng))
    feeds = {'Monochrome': ('image', 'monochrome'), 'Paragraph': ('monochrome', 'paragraph'),
             'Line': ('monochrome', 'line'), 'Char': ('char_lines', 'char_labels')}
    opts = {n: (O.MomentumState(lr, 0.0) if optimizer == 'sgd' else O.AdamState(lr)) for n in nets}

    def one_step():
        for name, net in nets.items():
            x, y = feeds[name]
            net.train_step(data[x], data[y], opts[name])
    one_step()                                   # warm-up (page-in, BLAS threads)
    steps, t0 = 0, time.perf_counter()
    while True:
        one_step()
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 8:
            break
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:   # noqa: BLE001
        threads = os.cpu_count() or 1
    return {'value': round(sample * steps / el, 3), 'unit': 'images/s', 'cores': int(threads),
            'host_cpus': os.cpu_count(), 'kind': 'port',
            'sample': f'{steps} train steps of the 4 nets on {sample} pages {height}x{width} + {sample} line '
                      f'strips 32x{char_width}, float64 NumPy oracle (im2col+BLAS), {el:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32, help='pages per GPU per step')
    ap.add_argument('--height', type=int, default=256)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--char-width', type=int, default=64)
    ap.add_argument('--optimizer', default='sgd', choices=['sgd', 'adam'])
    ap.add_argument('--lr', type=float, default=0.0015)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-overlap', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='end every step with the main stream waiting for all four net streams (default: a net starts '
                         'its next step as soon as ITS previous step is done; the timed region still ends with a full '
                         'device synchronisation)')
    ap.add_argument('--dp-side-stream', action='store_true',
                    help='N > 1: all-reduce with async_op=True on torch\'s internal NCCL stream (+ early bucket) instead of '
                         'synchronously inside each net\'s lane (the default; the side stream is one hardware queue too many)')
    ap.add_argument('--no-graphs', action='store_true', help='eager launches (default: HIP graphs)')
    ap.add_argument('--graphs', action='store_true',
                    help='(the default) replay Paragraph, Line and Char as HIP graphs (PageTrainer(graphs=True)); '
                         'Monochrome stays eager so that the HIP events around the dominant kernel keep working.  Host '
                         'enqueue 1.1 -> 0.3 ms/step; under data parallelism the all-reduce is issued eagerly between a '
                         'net\'s two graphs, from its lane (one-rank RCCL rehearsal, UOCR_BENCH_FORCE_DP=1: 0.98 ms/step '
                         'with graphs, 1.26 ms eager -- the eager step is bound by the host)')
    ap.add_argument('--skip-input-grads', action='store_true',
                    help='DIAGNOSTIC: do not compute the gradient w.r.t. the page inputs (unused by training; the '
                         'reference computes it, and so does the default run)')
    ap.add_argument('--solo-replay', action='store_true',
                    help='after the timed loop replay the dominant launch alone (roofline.solo_*); off by default so '
                         'that a rocprofv3 trace of this command averages only the in-loop launches')
    ap.add_argument('--h2d', action='store_true',
                    help='upload every batch as uint8 from pinned host memory on a copy stream and convert on the '
                         'device (PCIe-inclusive rate; flagged in metric and config, never the headline value)')
    args = ap.parse_args()

    # The contract is ONE line on stdout.  Libraries write there too (RCCL prints a version banner when the
    # communicator is created): everything but the final JSON line goes to stderr, at the descriptor level.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    # One hardware queue per stream: ROCm's default is 4 queues per process for main + 3 net lanes + copy /
    # RCCL streams, so two of them share a queue and run one after the other (1.25 -> 1.10 ms/step).  The GPU
    # keeps 4 queues running at a time, which is why PageTrainer uses 3 lanes, not 4.  Read by the HIP runtime
    # when it starts, hence set before torch touches the GPU.
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
        args.gpus = world
    args.graphs = not args.no_graphs
    # UOCR_BENCH_REHEARSAL=1: several ranks on ONE card with the gloo backend (gradients staged through the
    # host, parallel.DataParallel) -- exercises this file's multi-rank path where RCCL would refuse two ranks
    # on one device; never a measurement
    rehearsal = os.environ.get('UOCR_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    # UOCR_BENCH_FORCE_DP=1: a ONE-rank RCCL process group with the data-parallel machinery switched on
    # (flat gradient buffers, all-reduce between the graphs, waits on the lanes) -- the closest a one-GPU box
    # gets to the N > 1 code path with the real backend; never a measurement either
    force_dp = world == 1 and os.environ.get('UOCR_BENCH_FORCE_DP') == '1'
    if world > 1 or force_dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if rehearsal:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))

    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP

    CP.use_gpu(local_rank)
    CP.set_dtype('float32')
    CP.lazy_losses = True                        # losses stay on the device; fetched after the loop
    rt = CP.runtime()

    trainer = PageTrainer(args.batch, args.height, args.width, args.char_width, args.optimizer, args.lr,
                          seed=0, overlap=not args.no_overlap, input_grads=not args.skip_input_grads,
                          graphs=args.graphs, eager_nets=('Monochrome',),    # probed kernel stays eager
                          pipelined=not args.no_pipeline, data_parallel=True if force_dp else None,
                          dp_side_stream=args.dp_side_stream)
    layers = make_page_batch(args.batch, args.height, args.width, args.char_width, seed=1234 + rank)
    context = trainer.make_context(layers)       # inputs resident in HBM before the timed region

    # dominant kernel = the single longest launch of the step in the rocprofv3 trace (profiles/): the fused
    # backward of the Monochrome block (csrc/conv_pair.hip, entry uocr_conv_pair_bwd).  It recomputes the
    # 16-channel activation on chip, so it is bound by f32 multiply-adds, not HBM: priced in FLOP/s against
    # the f32 peak (157.3 TF, matrix = vector rate on gfx950).  ALGORITHMIC flops = the layer-by-layer
    # algorithm, recompute not counted: conv_2 dw, conv_2 dx, conv_1 dw (+ conv_1 dx when the page-input
    # gradient is wanted), each 2*9*16 per pixel.
    npix = args.batch * args.height * args.width
    n_convs = 3 if args.skip_input_grads else 4
    probe = KernelProbe(rt, 'uocr_conv_pair_bwd', lambda a: True)
    dominant = {'kernel': 'fused backward of conv3x3(1->16)+LeakyReLU+conv3x3(16->1)+Sigmoid (Monochrome; '
                          'conv_pair_bwd_kernel + its finish kernel)',
                'flops': 2.0 * 9 * 16 * n_convs * npix,
                'bytes': 4.0 * npix * (4 if not args.skip_input_grads else 3)}
    traffic = None            # HBM bytes per launch from rocprofv3 PMC passes, when a profile is committed
    tpath = os.path.join(ROOT, 'profiles', 'dominant_kernel_traffic.json')
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get('hbm_bytes_per_launch')

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
        if feeder is None:
            return trainer.step(context)
        if statics:                     # graph replay: convert straight into the arrays the graphs read
            trainer.join()              # ... once the lanes are done with the previous batch
            ctx = feeder.context(into=statics)
        else:
            ctx = feeder.context()      # batch i (uploaded while step i-1 ran)
        feeder.stage(layers_u8)         # start the upload of batch i+1
        return trainer.step(ctx)

    if args.graphs:
        try:
            trainer.capture(context)
        except Exception as exc:                   # keep measuring: eager launches are the same computation
            print(f'[bench] HIP graph capture failed ({type(exc).__name__}: {exc}); continuing with eager launches',
                  file=sys.stderr, flush=True)
            trainer.graphs, trainer._captured, args.graphs = False, None, False
            torch.cuda.synchronize()
        if feeder is not None and args.graphs:
            statics.update(trainer.static_inputs())
    for _ in range(args.warmup):
        losses = one_step()
    barrier()
    probe.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    probe.enabled = False

    t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    final = {n: [float(v) for v in l['output_losses']] for n, l in losses.items()}

    if rank == 0:
        kernel_ms, launches = probe.mean_ms()
        achieved = dominant['flops'] / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        solo_ms = probe.solo_ms() if args.solo_replay else None
        images = args.batch * world * args.steps
        out = {
            'metric': 'document-images/sec (fwd+bwd) on 256x512 synthetic pages' +
                      (' [DIAGNOSTIC: PCIe upload of every batch inside the timed region]' if args.h2d else '') +
                      (' [DIAGNOSTIC: page-input gradients not computed]' if args.skip_input_grads else ''),
            'value': round(images / elapsed, 2),
            'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': f'BASELINE configs[2]: my_model train step (fwd+loss+bwd+L2+{args.optimizer.upper()}) of '
                            f'Monochrome+Paragraph+Line on ({args.batch},{args.height},{args.width},1) pages and Char on '
                            f'({args.batch},32,{args.char_width},1) line strips per GPU',
                'batch_per_gpu': args.batch, 'global_batch': args.batch * world,
                'page': [args.height, args.width], 'optimizer': args.optimizer,
                'parallelism': f'dp{world}',
                'grad_allreduce': ('gloo (REHEARSAL on one card)' if rehearsal else 'rccl, 1 flat buffer per net') if world > 1 else ('rccl, ONE rank (REHEARSAL)' if force_dp else None),
                'final_losses': final,
                'h2d_inclusive': bool(args.h2d), 'input_grads': not args.skip_input_grads, 'hip_graphs': bool(args.graphs), 'pipelined_lanes': not args.no_pipeline,
                'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'default (4)'),
            },
            'roofline': {'bound': 'mfma', 'kernel': dominant['kernel'], 'achieved': round(achieved, 2),
                         'peak': F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / F32_PEAK_TFLOPS, 4),
                         'traffic': traffic, 'avg_launch_us': round(kernel_ms * 1e3, 2), 'launches_timed': launches,
                         'algorithmic_flops_per_launch': dominant['flops'],
                         'algorithmic_bytes_per_launch': dominant['bytes'],
                         # in the timed region the four nets run on four streams, so this launch shares HBM
                         # with other kernels; the same launch replayed alone right after the loop:
                         'solo_launch_us': None if solo_ms is None else round(solo_ms * 1e3, 2),
                         'solo_achieved': None if not solo_ms else round(dominant['flops'] / (solo_ms * 1e-3) / 1e12, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.height, args.width, args.char_width, args.optimizer, args.lr)
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
