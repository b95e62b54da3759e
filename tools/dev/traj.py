#!/usr/bin/env python3
"""Loss trajectories of the page step with and without deferred weight-gradient groups (summation-order sensitivity)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from univer_ocr_amd.nn import CP
from univer_ocr_amd.my_model.trainer import PageTrainer
from univer_ocr_amd.my_model.synthetic import make_page_batch
CP.use_gpu(0); CP.set_dtype('float32'); CP.lazy_losses = True
b = make_page_batch(32, 256, 512, 64, seed=1)
for group in ((), ('all',)):
    tr = PageTrainer(32, 256, 512, 64, optimizer='sgd', lr=0.0015, seed=0, graphs=True, pipelined=True, group_wgrad=group)
    ctx = tr.make_context(b)
    for i in range(121):
        losses = tr.step(ctx)
        if i % 20 == 0:
            tr.join()
            print(group, i, {n: round(float(l['output_losses'][0]), 6) for n, l in losses.items()})
