#!/usr/bin/env python3
"""Where do the copy / fill launches of a captured train step come from?  Logs every uocr_d2d / uocr_memset_zero /
uocr_fill call (with its Python stack) during capture and during one replayed step."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from univer_ocr_amd.nn import CP
from univer_ocr_amd.my_model.trainer import PageTrainer
from univer_ocr_amd.my_model.synthetic import make_page_batch

CP.use_gpu(0)
CP.set_dtype('float32')
CP.lazy_losses = True
rt = CP.runtime()
orig = rt.call
phase = ['setup']
def call(name, *a):
    if name in ('uocr_d2d', 'uocr_memset_zero', 'uocr_fill') and phase[0] != 'setup':
        stack = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in traceback.extract_stack()[-8:-1][::-1])
        print(f'[{phase[0]}] {name} {stack}', flush=True)
    return orig(name, *a)
rt.call = call
trainer = PageTrainer(32, 256, 512, 64, seed=0, graphs=True, pipelined=True)
layers = make_page_batch(32, 256, 512, 64, seed=1)
context = trainer.make_context(layers)
phase[0] = 'capture'
trainer.capture(context)
for name, entry in trainer._captured.items():
    print(name, 'ring' if entry['arena'].ring is not None else 'NO RING', flush=True)
phase[0] = 'step'
trainer.step(context)
trainer.step(context)
torch.cuda.synchronize()
