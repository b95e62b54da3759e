import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from univer_ocr_amd.my_model.synthetic import make_page_batch
from univer_ocr_amd.my_model.trainer import PageTrainer
from univer_ocr_amd.nn import CP
CP.use_gpu(0); CP.lazy_losses = True
tr = PageTrainer(32)
ctx = tr.make_context(make_page_batch(32, seed=1))
for _ in range(5): tr.step(ctx)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): tr.step(ctx)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue {1e3*(t1-t0)/20:.3f} ms/step, total {1e3*(t2-t0)/20:.3f} ms/step, launches/step {CP.runtime().launches/25:.0f}')
# tiny batch: pure host cost
tr2 = PageTrainer(1, 32, 64, 16)
ctx2 = tr2.make_context(make_page_batch(1, 32, 64, 16, seed=1))
for _ in range(5): tr2.step(ctx2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): tr2.step(ctx2)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f'tiny-batch step (host-bound) {1e3*(t1-t0)/50:.3f} ms/step')
