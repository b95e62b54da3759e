"""Latency of a cross-stream event wait: ping-pong of tiny kernels between two streams
(kernel on A, record, B waits, kernel on B, record, A waits, ...).  Run with GPU_MAX_HW_QUEUES=4 / 8."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
streams = [torch.cuda.Stream() for _ in range(n_streams)]
extra = [torch.cuda.Stream() for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 0)]   # just to occupy queues
x = [torch.zeros(1024, device='cuda') for _ in streams]
for s in extra:
    with torch.cuda.stream(s):
        torch.zeros(16, device='cuda').add_(1)
torch.cuda.synchronize()
for reps in (200, 2000):
    t0 = time.perf_counter()
    ev = None
    for i in range(reps):
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            if ev is not None:
                streams[k].wait_event(ev)
            x[k].add_(1.0)
            ev = torch.cuda.Event()
            ev.record(streams[k])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f'queues={os.environ.get("GPU_MAX_HW_QUEUES", "default")} streams={n_streams} extra={len(extra)}: '
      f'{1e6 * dt / reps:.1f} us per hop (kernel + record + cross-stream wait)')
