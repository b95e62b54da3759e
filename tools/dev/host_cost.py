#!/usr/bin/env python3
"""Is the page step bound by the host's graph launches?  (1) host time of each uocr_graph_launch, per net;
(2) the same replays issued from one thread per lane (ctypes drops the GIL inside the call) against one thread."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np
import torch
from univer_ocr_amd.nn import CP
from univer_ocr_amd.my_model.trainer import PageTrainer
from univer_ocr_amd.my_model.synthetic import make_page_batch

dtype = sys.argv[1] if len(sys.argv) > 1 else 'float32'
shape = (32, 256, 512) if dtype == 'float32' else (8, 1024, 2048)
nets = ('Monochrome', 'Paragraph', 'Line', 'Char') if dtype == 'float32' else ('Monochrome', 'Paragraph', 'Line')
CP.use_gpu(0)
CP.set_dtype(dtype)
CP.lazy_losses = True
rt = CP.runtime()
trainer = PageTrainer(*shape, 64, seed=0, nets=nets, graphs=True, pipelined=True)
context = trainer.make_context(make_page_batch(*shape, 64, seed=1))
trainer.capture(context)
for _ in range(10):
    trainer.step(context)
torch.cuda.synchronize()
STEPS = 100
t0 = time.perf_counter()
for _ in range(STEPS):
    trainer.step(context)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'trainer.step: {1e3 * (time.perf_counter() - t0) / STEPS:.3f} ms/step, host {1e3 * th / STEPS:.3f} ms/step', flush=True)

launch = rt.lib.uocr_graph_launch
jobs = {}          # lane -> [(ctx, [handles])]
for comp in trainer._lane_order():
    e = trainer._captured[comp.name]
    ctx = rt._lanes[trainer.lanes[comp.name]][0]
    handles = [g.handle for g in e['begin'].parts] + [e['finish'].handle]
    jobs.setdefault(trainer.lanes[comp.name], []).append((comp.name, ctx, handles))
# (1) host time per launch, one thread
cost = {}
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    for lane, items in jobs.items():
        for name, ctx, handles in items:
            for k, h in enumerate(handles):
                c0 = time.perf_counter()
                launch(ctx, h)
                cost.setdefault((name, k), []).append(time.perf_counter() - c0)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'one thread, raw launches: {1e3 * (time.perf_counter() - t0) / STEPS:.3f} ms/step, host {1e3 * th / STEPS:.3f} ms/step')
for key, v in cost.items():
    print(f'   {key[0]:12s} graph {key[1]}: host median {1e6 * np.median(v):7.1f} us')
# (2) one thread per lane
def worker(items, n):
    for _ in range(n):
        for name, ctx, handles in items:
            for h in handles:
                launch(ctx, h)
for rep in range(2):
    threads = [threading.Thread(target=worker, args=(items, STEPS)) for items in jobs.values()]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f'one thread per lane: {1e3 * (time.perf_counter() - t0) / STEPS:.3f} ms/step, host {1e3 * th / STEPS:.3f} ms/step', flush=True)
