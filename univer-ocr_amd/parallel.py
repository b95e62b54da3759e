"""Data-parallel training: one process per GPU, RCCL all-reduce (torch.distributed backend "nccl"
on ROCm) of each model's FLAT gradient buffer (layers.ParamPack.grad) between backward and the
L2 / optimizer step.

The reference has no multi-GPU path (no collective call sites); this is the
MI355X-native addition named by BASELINE.json.  Page images are independent, so the batch is
sharded by rank and the only exchange is the gradient sum:
  * Dice / Jaccard sum their loss over the batch (losses.py:22-24)  -> SUM over ranks equals the
    single-GPU gradient of the global batch;
  * SoftmaxCE / SigmoidCE divide by the LOCAL batch (losses.py:54-56, 70-72) -> SUM / world;
  * L2 is added after the all-reduce (Model.compute_loss_and_gradients calls grad_sync between
    backward and regularize), so it is counted once.
Gradient volume is <= 3.2 MB (803 395 parameters), i.e. latency-bound on xGMI: ONE collective per
model, issued as soon as that model's backward has been enqueued.

Where the collective runs (`side_stream`):
  * False (default): a synchronous `dist.all_reduce` from inside the net's lane, which ProcessGroupNCCL
    launches on the CURRENT stream -- the RCCL kernel sits in the lane between backward and the optimizer
    tail, the other lanes keep the GPU busy, and no further hardware queue becomes active.  The GPU runs
    about four queues at a time: with three lanes plus torch's internal NCCL stream
    plus the event traffic between them the step went from 1.0 to 2.2 ms (measured with a one-rank RCCL
    group, `UOCR_BENCH_FORCE_DP=1 python bench.py`), eager or graph replay alike.  Collectives of ONE
    communicator must not run concurrently on different streams, so every net gets its own process group
    (= its own RCCL communicator, `dist.new_group`): the lanes' collectives are independent of each other,
    small (1-2 channels) and can be resident together, so their relative order may differ between ranks;
  * True: `async_op=True` on torch's internal NCCL stream with the optimizer step waiting on the work, plus
    the early bucket of the Char net -- the classic overlap scheme, right when the compute is ONE stream.
"""
import torch
import torch.distributed as dist

from .nn import ops
from .nn.losses import SigmoidCrossEntropy, SoftmaxCrossEntropy


def mean_type_loss(model):
    losses = model.loss if isinstance(model.loss, list) else [model.loss]
    return all(isinstance(fn, (SoftmaxCrossEntropy, SigmoidCrossEntropy)) for fn in losses)


class DataParallel:
    def __init__(self, models, process_group=None, overlap=True, bucket_bytes=1 << 20, side_stream=False):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised (launch with torch.distributed.run)')
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.overlap = overlap
        self.side_stream = side_stream
        self.models = list(models.values()) if isinstance(models, dict) else list(models)
        self._pending = {}
        self._groups = {}         # id(model) -> the model's own process group (lane-stream collectives)
        self._early = {}          # id(model) -> work of the early bucket of the current step (None: staged path)
        self._plans = {}          # id(model) -> (trigger node, lo, hi) of the early bucket
        self.bucket_bytes = bucket_bytes
        for model in self.models:
            if model.pack is None:
                raise RuntimeError('data parallel needs initialised models with a ParamPack')
            value = model.pack.value.t
            if value.is_cuda and dist.get_backend(process_group) == 'gloo':
                host = value.cpu()
                dist.broadcast(host, src=0, group=process_group)
                value.copy_(host)
            else:
                dist.broadcast(value, src=0, group=process_group)       # identical replicas
            model.grad_sync = self._sync
            # (new_group is collective over the default group: same model order on every rank)
            self._groups[id(model)] = process_group if side_stream else dist.new_group(
                ranks=None if process_group is None else dist.get_process_group_ranks(process_group),
                backend=dist.get_backend(process_group))
            plan = self.early_bucket(model, bucket_bytes)
            if plan is not None and overlap and (side_stream or dist.get_backend(process_group) == 'gloo'):
                self._plans[id(model)] = plan
                model.bucket_hook = self._bucket_ready

    @staticmethod
    def early_bucket(model, bucket_bytes):
        """Bucketed all-reduce inside ONE net: the tail of the flat gradient buffer whose layers finish
        their backward first (the Char net: the dense layers, 2.9 of 3.2 MB, are done before the conv
        block's backward starts) is reduced as soon as it is final, under the rest of the backward pass.
        Returns (trigger node, lo, hi): the suffix [lo, hi) of the pack is final once `trigger` has run
        its backward; None when no suffix of >= bucket_bytes finishes early."""
        pack = model.pack
        order = {node: i for i, node in enumerate(reversed(model._plan))}
        owner = {id(p): name for name, layer in model.layers.items() for p in layer.params().values()}
        itemsize = pack.value.t.element_size()
        best, ready, last = None, -1, len(order) - 1
        for p, off, size in reversed(pack.entries):
            ready = max(ready, order[owner[id(p)]])
            if (pack.total - off) * itemsize >= bucket_bytes and off > 0 and ready < last:
                if best is None or ready < best[0]:
                    best = (ready, off)
        if best is None:
            return None
        trigger = list(reversed(model._plan))[best[0]]
        return trigger, best[1], pack.total

    def _reduce(self, model, tensor):
        """SUM all-reduce of a (slice of a) gradient buffer of `model`; returns the work to wait for, or None
        when nothing is left to wait for (lane-stream collective; gloo staging through the host, see _sync)."""
        group = self._groups[id(model)]
        if tensor.is_cuda and dist.get_backend(group) == 'gloo':
            host = tensor.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            tensor.copy_(host)
            return None
        if self.side_stream:
            return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=True)
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)         # on the caller's (lane) stream
        return None

    def _bucket_ready(self, model, node):
        """Model.backward calls this after every node; the early bucket goes out when its trigger has run."""
        trigger, lo, hi = self._plans[id(model)]
        if node == trigger:
            self._early[id(model)] = self._reduce(model, model.pack.grad.t[lo:hi])

    def _sync(self, model):
        """Called by Model.compute_loss_and_gradients right after backward."""
        grad = model.pack.grad.t
        works = []
        if id(model) in self._early:               # the tail went out during backward: only the head is left
            works.append(self._early.pop(id(model)))
            grad = grad[:self._plans[id(model)][1]]
        # (gloo with CUDA storage = rehearsal of the N > 1 path on ONE card, which RCCL refuses: _reduce
        # stages through the host and returns None.  Never used with the nccl backend.)
        works.append(self._reduce(model, grad))
        works = [w for w in works if w is not None]
        if works and self.overlap and model.defer_grad_sync:
            self._pending[id(model)] = works       # finished later by wait(model)
        else:
            self._finish(model, works)

    def _finish(self, model, works):
        for work in works or ():
            work.wait()                            # orders the compute stream after the collective
        if mean_type_loss(model):
            grad = model.pack.grad
            if grad.t.is_cuda:
                ops.scale_(grad, 1.0 / self.world)
            else:                                  # gloo tests on CPU storage: no HIP kernels there
                grad.t.mul_(1.0 / self.world)

    def wait(self, model):
        works = self._pending.pop(id(model), None)
        if works is not None:
            self._finish(model, works)

    def replicas_in_sync(self, model, tol=0.0):
        """Debug check: every rank holds the same weights."""
        mine = model.pack.value.t.detach().cpu().clone()
        ref = mine.clone()
        dist.broadcast(ref, src=0, group=self.group)
        diff = (mine - ref).abs().max()
        dist.all_reduce(diff, op=dist.ReduceOp.MAX, group=self.group)
        return float(diff.item()) <= tol
