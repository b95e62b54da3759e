"""When does each net's lane finish inside the concurrent step?  (timing events at fork and at each join)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from univer_ocr_amd.my_model.synthetic import make_page_batch
from univer_ocr_amd.my_model.trainer import PageTrainer
from univer_ocr_amd.nn import CP

import argparse
ap = argparse.ArgumentParser()
ap.add_argument('--highres', action='store_true', help='the highres-fp16 configuration: 8 pages 1024x2048, float16, no Char')
args = ap.parse_args()
CP.use_gpu(0)
CP.set_dtype('float16' if args.highres else 'float32')
CP.lazy_losses = True
if args.highres:
    make = lambda **kw: PageTrainer(8, 1024, 2048, 64, nets=('Monochrome', 'Paragraph', 'Line'), **kw)
    batch = make_page_batch(8, 1024, 2048, 64, seed=1)
else:
    make = lambda **kw: PageTrainer(32, **kw)
    batch = make_page_batch(32, seed=1)
trainer = make()
context = trainer.make_context(batch)
for _ in range(5):
    trainer.step(context)
rt = CP.runtime()
acc = {}
reps = 20
for _ in range(reps):
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    start = torch.cuda.Event(enable_timing=True)
    start.record(main)
    comps = trainer.model_system.components
    mids, ends = {}, {}
    for comp in comps:
        with rt.lane(trainer.lanes[comp.name]) as stream:
            stream.wait_event(start)
            comp.selector(context)
            X, y = next(comp.selector.get())
            comp.model.train_begin(X, y)
            mids[comp.name] = torch.cuda.Event(enable_timing=True)
            mids[comp.name].record(stream)
    for comp in comps:
        with rt.lane(trainer.lanes[comp.name]) as stream:
            comp.model.train_finish()
            ends[comp.name] = torch.cuda.Event(enable_timing=True)
            ends[comp.name].record(stream)
    torch.cuda.synchronize()
    for name in mids:
        a = acc.setdefault(name, [0.0, 0.0])
        a[0] += start.elapsed_time(mids[name])
        a[1] += start.elapsed_time(ends[name])
for name, (m, e) in acc.items():
    print(f'{name:12s} fwd+bwd done at {m / reps:6.3f} ms, step done at {e / reps:6.3f} ms')

# the same with the per-net HIP graphs (host enqueue out of the picture)
trainer = make(graphs=True)
context = trainer.make_context(batch)
trainer.capture(context)
for _ in range(3):
    trainer.step(context)
acc = {}
for _ in range(reps):
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    start = torch.cuda.Event(enable_timing=True)
    start.record(main)
    mids, ends = {}, {}
    for comp in trainer._lane_order():               # the enqueue order of PageTrainer.step
        with rt.lane(trainer.lanes[comp.name]) as stream:
            stream.wait_event(start)
            trainer._captured[comp.name]['begin'].replay()
            mids[comp.name] = torch.cuda.Event(enable_timing=True)
            mids[comp.name].record(stream)
            trainer._captured[comp.name]['finish'].replay()
            ends[comp.name] = torch.cuda.Event(enable_timing=True)
            ends[comp.name].record(stream)
    torch.cuda.synchronize()
    for name in mids:
        a = acc.setdefault(name, [0.0, 0.0])
        a[0] += start.elapsed_time(mids[name])
        a[1] += start.elapsed_time(ends[name])
print('with graphs:')
for name, (m, e) in acc.items():
    print(f'{name:12s} fwd+bwd done at {m / reps:6.3f} ms, step done at {e / reps:6.3f} ms')
