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
model, issued asynchronously as soon as that model's backward has been enqueued, so it overlaps
the next model's forward/backward; the optimizer step waits on it.
"""
import torch
import torch.distributed as dist

from .nn import ops
from .nn.losses import SigmoidCrossEntropy, SoftmaxCrossEntropy


def mean_type_loss(model):
    losses = model.loss if isinstance(model.loss, list) else [model.loss]
    return all(isinstance(fn, (SoftmaxCrossEntropy, SigmoidCrossEntropy)) for fn in losses)


class DataParallel:
    def __init__(self, models, process_group=None, overlap=True):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised (launch with torch.distributed.run)')
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.overlap = overlap
        self.models = list(models.values()) if isinstance(models, dict) else list(models)
        self._pending = {}
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

    def _sync(self, model):
        """Called by Model.compute_loss_and_gradients right after backward."""
        grad = model.pack.grad.t
        if grad.is_cuda and dist.get_backend(self.group) == 'gloo':
            # rehearsal of the N > 1 path on ONE card (several ranks share cuda:0, which RCCL refuses):
            # stage through the host.  Never used with the nccl backend.
            host = grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            grad.copy_(host)
            self._finish(model, None)
            return
        work = dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.overlap and model.defer_grad_sync:
            self._pending[id(model)] = work        # finished later by wait(model)
        else:
            self._finish(model, work)

    def _finish(self, model, work):
        if work is not None:
            work.wait()                            # orders the compute stream after the collective
        if mean_type_loss(model):
            grad = model.pack.grad
            if grad.t.is_cuda:
                ops.scale_(grad, 1.0 / self.world)
            else:                                  # gloo tests on CPU storage: no HIP kernels there
                grad.t.mul_(1.0 / self.world)

    def wait(self, model):
        work = self._pending.pop(id(model), None)
        if work is not None:
            self._finish(model, work)

    def replicas_in_sync(self, model, tol=0.0):
        """Debug check: every rank holds the same weights."""
        mine = model.pack.value.t.detach().cpu().clone()
        ref = mine.clone()
        dist.broadcast(ref, src=0, group=self.group)
        diff = (mine - ref).abs().max()
        dist.all_reduce(diff, op=dist.ReduceOp.MAX, group=self.group)
        return float(diff.item()) <= tol
