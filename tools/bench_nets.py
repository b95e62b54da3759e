"""Per-net step time: each of the four nets trained alone (one stream), then all four with lanes.
Shows which net bounds the multi-stream step.  Usage: python tools/bench_nets.py [--batch 32]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--skip-input-grads', action='store_true')
    ap.add_argument('--graphs', action='store_true')
    ap.add_argument('--pipelined', action='store_true')
    ap.add_argument('--lanes4', action='store_true', help='one lane per net instead of the 3 balanced lanes')
    ap.add_argument('--option', action='append', default=[], help='ctx option key=value (applied to every lane)')
    ap.add_argument('--only', default='', help='comma-separated nets: time just this combination')
    args = ap.parse_args()
    if args.pipelined:                          # one hardware queue per net stream (see bench.py)
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    import torch
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    CP.lazy_losses = True          # as bench.py: losses stay on the device until read
    layers = make_page_batch(args.batch, 256, 512, 64, seed=1)
    all_nets = ('Monochrome', 'Paragraph', 'Line', 'Char')
    combos = [tuple(args.only.split(','))] if args.only else [(n,) for n in all_nets] + [all_nets]
    for nets in combos:
        trainer = PageTrainer(args.batch, 256, 512, 64, nets=nets, input_grads=not args.skip_input_grads,
                              graphs=args.graphs, pipelined=args.pipelined,
                              **({'lane_groups': None} if args.lanes4 else {}))
        for opt in args.option:
            k, v = opt.split('=')
            CP.runtime().set_option(k, int(v))
        context = trainer.make_context(layers)
        for _ in range(5):
            trainer.step(context)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(context)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f'{"+".join(nets):40s} {1e3 * t / args.steps:7.3f} ms/step   host enqueue {1e3 * t_host / args.steps:6.3f} ms',
              flush=True)


if __name__ == '__main__':
    main()
