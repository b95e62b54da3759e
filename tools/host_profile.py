"""cProfile of the host side of PageTrainer.step (tiny batch: the GPU is never the limit)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from univer_ocr_amd.my_model.synthetic import make_page_batch
from univer_ocr_amd.my_model.trainer import PageTrainer
from univer_ocr_amd.nn import CP

CP.use_gpu(0)
CP.set_dtype('float32')
CP.lazy_losses = True
trainer = PageTrainer(1, 32, 64, 16, pipelined=True)
context = trainer.make_context(make_page_batch(1, 32, 64, 16, seed=1))
for _ in range(20):
    trainer.step(context)
torch.cuda.synchronize()
prof = cProfile.Profile()
prof.enable()
for _ in range(300):
    trainer.step(context)
prof.disable()
torch.cuda.synchronize()
stats = pstats.Stats(prof)
stats.sort_stats('tottime').print_stats(28)
