#!/usr/bin/env python3
"""Per-node cost of a replayed HIP graph of tiny kernels, on one lane and on three lanes at once: is there a
process-wide serialiser behind uocr_graph_launch (then three lanes take three times as long), or is the cost the
queue's own launch latency (then they overlap)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np
import torch
from univer_ocr_amd.nn import CP, ops

CP.use_gpu(0)
CP.set_dtype('float32')
rt = CP.runtime()
NODES, REPS = 100, 200
lanes = [rt.add_lane() for _ in range(3)]
graphs = []
for lane in lanes:
    with rt.lane(lane):
        x = CP.empty((256,), np.float32)
        pool = torch.cuda.MemPool()
        with rt.capture(pool) as g:
            for _ in range(NODES):
                ops.fill_(x, 1.0)
        graphs.append((g, x))
torch.cuda.synchronize()
def run(which):
    for _ in range(5):
        for i in which:
            with rt.lane(lanes[i]):
                graphs[i][0].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        for i in which:
            with rt.lane(lanes[i]):
                graphs[i][0].replay()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f'lanes {which}: {1e6 * t / REPS / NODES:.2f} us per node per lane (wall), host {1e6 * th / REPS / NODES:.2f} us', flush=True)
run([0]); run([0, 1]); run([0, 1, 2]); run([0])
