"""Data-parallel training: one process per GPU, RCCL all-reduce over xGMI of the models' flat gradient
buffers (layers.ParamPack.grad) between backward and the L2 / optimizer step -- through the C ABI
(include/univer_hip.h: uocr_dp_init / uocr_dp_allreduce_sum / uocr_dp_broadcast / uocr_dp_finalize),
not through torch.distributed.  torch.distributed (any backend, normally gloo) is only the channel that
carries rank 0's 128-byte RCCL id to the other ranks; `rendezvous=` replaces it with any other channel.

The reference has no multi-GPU path (no collective call sites); this is the
MI355X-native addition named by BASELINE.json, hooked where the reference's step loop has a complete
gradient and has not yet updated (my_model/trainer.py:213-233 -> nn/model_system.py:104-118 ->
nn/models.py:250-254).  Page images are independent, so the batch is sharded by rank and the only
exchange is the gradient sum:
  * Dice / Jaccard sum their loss over the batch (losses.py:22-24)  -> SUM over ranks equals the
    single-GPU gradient of the global batch;
  * SoftmaxCE / SigmoidCE divide by the LOCAL batch (losses.py:54-56, 70-72) -> SUM / world;
  * L2 is added after the all-reduce (Model.compute_loss_and_gradients calls grad_sync between
    backward and regularize), so it is counted once.
Gradient volume is <= 3.2 MB (803 395 parameters), i.e. latency-bound on xGMI.

Deadlock freedom by construction: ONE communicator, and every collective is issued from ONE stream (the
communication lane: its own uocr_ctx) in an order that depends on nothing but the list of models --
`flush()` sorts the requests of a step by the models' construction order -- so every rank enqueues the
same sequence.  Lanes and the communication lane are ordered with events only (uocr_event_record /
uocr_stream_wait_event): a net's lane records "gradient complete", the communication lane waits for it,
reduces, records "reduced", and the net's lane waits for that before its optimizer tail.

All packs are re-homed into ONE flat value buffer and ONE flat gradient buffer (`coalesce=True`: what is
queued when flush() runs and lies next to each other goes out as one collective -- a PageTrainer step is
then ONE all-reduce of all four nets, `bench.py --dp-single-collective`; False, the default: one
collective per net, so a short net's optimizer tail does not wait for the longest net's backward).

backend='gloo' is the rehearsal / CPU-test path (torch.distributed on host copies of the buffers, same
queueing and ordering logic); it is never used for a measurement.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from .hip import lib as hiplib
from .nn import ops
from .nn.gpu import CP, DeviceArray
from .nn.losses import SigmoidCrossEntropy, SoftmaxCrossEntropy


def mean_type_loss(model):
    losses = model.loss if isinstance(model.loss, list) else [model.loss]
    return all(isinstance(fn, (SoftmaxCrossEntropy, SigmoidCrossEntropy)) for fn in losses)


def torch_rendezvous(id_bytes, group=None):
    """Default channel for the RCCL id: rank 0's bytes to everybody over an initialised torch.distributed
    group (gloo or nccl).  Returns (rank, world, id_bytes)."""
    if not dist.is_initialized():
        raise RuntimeError('no rendezvous: initialise torch.distributed (torch.distributed.run + '
                           'init_process_group("gloo")) or pass rendezvous=callable to DataParallel')
    box = [id_bytes]
    src = dist.get_global_rank(group, 0) if group is not None else 0      # rank 0 OF THE GROUP, as a global rank
    dist.broadcast_object_list(box, src=src, group=group)
    return dist.get_rank(group), dist.get_world_size(group), box[0]


class _Rccl:
    """The C-ABI communicator of this process, bound to a communication lane (ctx + stream) of the runtime."""
    name = 'rccl'

    def __init__(self, rendezvous=None, group=None):
        rt = CP.runtime()
        self.rt = rt
        lib = rt.lib
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = C.create_string_buffer(hiplib.DP_UNIQUE_ID_BYTES)
        if rank == 0 or rendezvous is not None:
            rc = lib.uocr_dp_get_unique_id(ident)
            if rc != 0:
                raise hiplib.HipError(f'uocr_dp_get_unique_id failed ({rc}): is librccl.so.1 on the loader path?')
        if rendezvous is not None:
            self.rank, self.world, raw = rendezvous(ident.raw)
        elif dist.is_initialized():
            self.rank, self.world, raw = torch_rendezvous(ident.raw if rank == 0 else None, group)
        else:
            self.rank, self.world, raw = 0, 1, ident.raw          # a ONE-rank communicator (rehearsal)
        # one communication lane per process, reused by every communicator made later (train.py builds a DataParallel per
        # stage: a lane per stage would leak a context, a stream and a workspace each time)
        if getattr(rt, '_comm_lane', None) is None:
            rt._comm_lane = rt.add_lane(workspace_mb=1)
        self.lane = rt._comm_lane
        with rt.lane(self.lane):
            rt.call('uocr_dp_init', self.rank, self.world, C.c_char_p(raw))
            try:
                # known-answer check of the fresh communicator: a SUM of ones over the ranks must read `world` everywhere
                # (a mis-bound library or a half-initialised ring otherwise shows up as silently wrong gradients)
                probe = CP.full((8,), 1.0, np.float32)
                rt.call('uocr_dp_allreduce_sum', probe.ptr, probe.size, probe.code & 0xff)
                rt.call('uocr_stream_sync')
                got = CP.asnumpy(probe)
                if not np.all(got == float(self.world)):
                    raise hiplib.HipError(f'RCCL all-reduce self-test failed on rank {self.rank}: sum of ones over '
                                          f'{self.world} ranks read {got.tolist()}')
            except Exception:
                try:
                    rt.call('uocr_dp_finalize')          # no half-open communicator stays behind in the library
                except hiplib.HipError:
                    pass
                raise
        self._events = []

    def event(self):
        ev = C.c_void_p()
        if self.rt.lib.uocr_event_create(C.byref(ev)) != 0:
            raise hiplib.HipError('uocr_event_create failed')
        self._events.append(ev)
        return ev

    def record(self, ev):
        """on the CURRENT ctx (the lane the caller is in)"""
        self.rt.call('uocr_event_record', ev)

    def wait(self, ev):
        self.rt.call('uocr_stream_wait_event', ev)

    def comm_lane(self):
        return self.rt.lane(self.lane)

    def all_reduce(self, array):
        """in place SUM on the communication lane (caller is inside comm_lane())"""
        self.rt.call('uocr_dp_allreduce_sum', array.ptr, array.size, array.code & 0xff)

    def broadcast(self, array, root=0):
        torch.cuda.synchronize()                 # construction time: whatever filled the buffer is done
        with self.comm_lane():
            self.rt.call('uocr_dp_broadcast', array.ptr, array.size, array.code & 0xff, root)
            self.rt.call('uocr_stream_sync')

    def scale(self, array, factor):
        ops.scale_(array, factor)

    def close(self):
        with self.comm_lane():
            self.rt.call('uocr_dp_finalize')
        for ev in self._events:
            self.rt.lib.uocr_event_destroy(ev)
        self._events = []


class _Gloo:
    """torch.distributed on host tensors: CPU tests (storage-only DeviceArrays) and the several-ranks-on-one-card
    rehearsal.  Synchronous, so the event calls are no-ops and ordering is the host's program order."""
    name = 'gloo'

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised (launch with torch.distributed.run)')
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def event(self):
        return None

    def record(self, ev):
        pass

    def wait(self, ev):
        pass

    class _Here:
        def __enter__(self):
            return None

        def __exit__(self, *exc):
            return False

    def comm_lane(self):
        return self._Here()

    def all_reduce(self, array):
        t = array.t
        if t.is_cuda:
            torch.cuda.synchronize()             # (rehearsal on one card: the other lanes' gradients too)
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast(self, array, root=0):
        t = array.t
        if t.is_cuda:
            host = t.cpu()
            dist.broadcast(host, src=root, group=self.group)
            t.copy_(host)
        else:
            dist.broadcast(t, src=root, group=self.group)

    def scale(self, array, factor):
        if array.t.is_cuda:
            ops.scale_(array, factor)
        else:
            array.t.mul_(factor)

    def close(self):
        pass


class DataParallel:
    def __init__(self, models, process_group=None, overlap=True, coalesce=False, backend=None, rendezvous=None,
                 split_buckets=True, split_min_elements=100_000):
        self.models = list(models.values()) if isinstance(models, dict) else list(models)
        if not self.models or any(m.pack is None for m in self.models):
            raise RuntimeError('data parallel needs initialised models with a ParamPack')
        on_gpu = self.models[0].pack.value.t.is_cuda
        if backend is None:
            backend = 'rccl' if on_gpu else 'gloo'
        if backend == 'rccl' and not on_gpu:
            raise RuntimeError('the RCCL backend needs the models on a GPU')
        self.group = process_group
        self.comm = _Rccl(rendezvous, process_group) if backend == 'rccl' else _Gloo(process_group)
        self.backend = backend
        self.rank, self.world = self.comm.rank, self.comm.world
        self.overlap = overlap
        self.coalesce = coalesce
        self.collectives = 0                     # issued so far (tests, bench diagnostics)
        self.profile = False                     # True: keep an event triple per collective (see stats())
        self._prof = []
        self._order = {id(m): i for i, m in enumerate(self.models)}
        self._queue = []                         # (model, "gradient complete" event) of the current step
        self._reduced = {}                       # id(model) -> "reduced" event its lane still has to wait for
        self._rehome()
        self.comm.broadcast(self.flat_value)     # identical replicas: rank 0's weights
        self._ev_ready = {id(m): self.comm.event() for m in self.models}
        self._ev_done = {id(m): self.comm.event() for m in self.models}
        # Within-net overlap: a net whose backward finishes the gradients of a large TAIL of its pack early (the Char
        # net: its dense block holds 84 % of its 801 442 parameters and is differentiated before the three conv layers)
        # reduces that tail as soon as it is final, while the rest of its backward runs: Model.backward calls
        # bucket_hook(model, node) after every node
        self._split = {}                          # id(model) -> (node after whose backward the tail is final, tail offset)
        self._tail_done = {}                      # id(model) -> "tail reduced" event of the current step
        self._ev_tail_ready = {}
        self._ev_tail_done = {}
        if split_buckets and overlap:
            for m in self.models:
                found = self._find_split(m, split_min_elements)
                if found is not None:
                    self._split[id(m)] = found
                    self._ev_tail_ready[id(m)] = self.comm.event()
                    self._ev_tail_done[id(m)] = self.comm.event()
        for model in self.models:
            model.grad_sync = self._sync
            model.bucket_hook = self._bucket if id(model) in self._split else None

    # -- one flat value / gradient buffer for all models ------------------------------------------------------------
    def _rehome(self):
        packs = [m.pack for m in self.models]
        dtypes = {p.dtype for p in packs}
        if len(dtypes) != 1:
            raise RuntimeError(f'data parallel: the models\' parameter dtypes differ: {dtypes}')
        total = sum(p.total for p in packs)
        self.flat_value = CP.zeros((total,), packs[0].dtype)
        self.flat_grad = CP.zeros((total,), packs[0].dtype)
        self._slice = {}
        off = 0
        for model, pack in zip(self.models, packs):
            pack.rebase(self.flat_value, self.flat_grad, off)
            self._slice[id(model)] = (off, off + pack.total)
            off += pack.total

    def _view(self, lo, hi):
        return DeviceArray(self.flat_grad.t[lo:hi])

    def _find_split(self, model, min_elements):
        """(node, offset): once `node` has run in the backward pass, every parameter at pack offsets >= offset has its
        gradient -- the first such point where that tail is at least `min_elements` and half of the pack."""
        pack = model.pack
        where = {id(p): (off, size) for p, off, size in pack.entries}
        plan = model._plan if getattr(model, '_plan', None) is not None else model._toposort()
        done, first = set(), min(off for _, off, _ in pack.entries)
        for node in reversed(plan):
            layer = model.layers.get(node) if hasattr(model.layers, 'get') else None
            params = layer.params() if layer is not None else {}
            if not params:
                continue
            done.update(id(p) for p in params.values())
            lo = pack.total
            for p, off, size in sorted(pack.entries, key=lambda e: -e[1]):      # grow the tail downwards
                if id(p) not in done:
                    break
                lo = off
            tail = pack.total - lo
            if lo > first and tail >= min_elements and 2 * tail >= pack.total:
                return node, lo
        return None

    def split_node(self, model):
        found = self._split.get(id(model))
        return None if found is None else found[0]

    def _bucket(self, model, node):
        """bucket_hook of Model.backward: after the split node, reduce the tail of the model's gradient right away (on
        the communication lane; the call order is the host's program order, the same on every rank)."""
        found = self._split.get(id(model))
        if found is None or node != found[0]:
            return
        if getattr(model, 'group_wgrad', False):
            from .nn.gpu import CP
            CP.runtime().flush_deferred()             # the tail's weight gradients are still only recorded
        if getattr(model, 'side_wgrad', False):
            from .nn.gpu import CP
            CP.runtime().join_side()                  # weight gradients of the tail still on the lane's side stream
        self.tail_ready(model)

    def tail_ready(self, model):
        """The tail [offset, end) of the model's gradient is final on the CURRENT lane: issue its all-reduce."""
        lo_rel = self._split[id(model)][1]
        lo, hi = self._slice[id(model)]
        ready = self.comm.event() if self.profile else self._ev_tail_ready[id(model)]
        self.comm.record(ready)
        with self.comm.comm_lane():
            self.comm.wait(ready)
            begin = self.comm.event() if self.profile else None
            if begin is not None:
                self.comm.record(begin)
            self.comm.all_reduce(self._view(lo + lo_rel, hi))
            self.collectives += 1
            if begin is not None:
                end = self.comm.event()
                self.comm.record(end)
                self._prof.append((ready, begin, end, hi - lo - lo_rel))
            if mean_type_loss(model):
                self.comm.scale(self._view(lo + lo_rel, hi), 1.0 / self.world)
            done = self._ev_tail_done[id(model)]
            self.comm.record(done)
        self._tail_done[id(model)] = done

    # -- the step -----------------------------------------------------------------------------------------------------
    def _sync(self, model):
        """Called by the model right after its backward, from the model's lane: the gradient is complete once
        everything enqueued on this lane so far has run."""
        # Deadlock freedom rests on every rank issuing the same collectives in the same order: ONE grad_sync per model
        # between two waits.  A second one (e.g. a ModelSystem list path that trains a model once per crop with a
        # data-dependent crop count) would give the ranks different collective counts: refuse it here, on every rank.
        if any(m is model for m, _ in self._queue) or id(model) in self._reduced:
            raise RuntimeError(f'data parallel: grad_sync called twice for one model within a step '
                               f'(accumulate the gradients of all crops locally and reduce once)')
        ev = self.comm.event() if self.profile else self._ev_ready[id(model)]     # (profiling: events are read later)
        self.comm.record(ev)
        self._queue.append((model, ev))
        if not (self.overlap and model.defer_grad_sync):
            self.wait(model)

    def flush(self):
        """Issue the collectives of everything queued, on the communication lane, in construction order of
        the models (identical on every rank whatever order the lanes were enqueued in)."""
        if not self._queue:
            return
        queue = sorted(self._queue, key=lambda item: self._order[id(item[0])])
        self._queue = []
        runs = []                                # [(lo, hi, [models])]: neighbours merge when coalescing
        for model, _ in queue:
            lo, hi = self._slice[id(model)]
            if id(model) in self._tail_done:          # its tail went out from bucket_hook already
                hi = lo + self._split[id(model)][1]
            if self.coalesce and runs and runs[-1][1] == lo:
                runs[-1] = (runs[-1][0], hi, runs[-1][2] + [model])
            else:
                runs.append((lo, hi, [model]))
        ready = {id(m): ev for m, ev in queue}
        with self.comm.comm_lane():
            for lo, hi, members in runs:
                for m in members:
                    self.comm.wait(ready[id(m)])
                begin = self.comm.event() if self.profile else None
                if begin is not None:
                    self.comm.record(begin)
                self.comm.all_reduce(self._view(lo, hi))
                self.collectives += 1
                if begin is not None:
                    end = self.comm.event()
                    self.comm.record(end)
                    self._prof.append((ready[id(members[0])], begin, end, hi - lo))
                for m in members:
                    if mean_type_loss(m):
                        mlo, mhi = self._slice[id(m)]
                        if id(m) in self._tail_done:
                            mhi = mlo + self._split[id(m)][1]
                        self.comm.scale(self._view(mlo, mhi), 1.0 / self.world)
                done = self._ev_done[id(members[-1])]
                self.comm.record(done)
                for m in members:
                    self._reduced[id(m)] = done

    def wait(self, model):
        """From the model's lane, before its optimizer tail: the lane waits (on the device) for the reduction."""
        self.flush()
        done = self._reduced.pop(id(model), None)
        if done is not None:
            self.comm.wait(done)
        tail = self._tail_done.pop(id(model), None)
        if tail is not None:
            self.comm.wait(tail)

    def stats(self):
        """Profile mode: per-collective device times from the events kept since `profile` was switched on --
        allreduce = first to last instruction of the collective on the communication lane, lane_wait = from "gradient
        complete" on the net's lane to "reduced" (what the net's optimizer tail waits for).  Synchronises."""
        import ctypes as C
        rows = []
        lib = getattr(getattr(self.comm, 'rt', None), 'lib', None)
        for ready, begin, end, count in self._prof:
            if lib is None or begin is None:
                continue
            a, w = C.c_float(), C.c_float()
            if lib.uocr_event_elapsed_ms_sync(begin, end, C.byref(a)) == 0 and \
                    lib.uocr_event_elapsed_ms_sync(ready, end, C.byref(w)) == 0:
                rows.append((a.value * 1e3, w.value * 1e3, count))
        self._prof = []
        if not rows:
            return None
        return {'collectives_timed': len(rows),
                'allreduce_us_mean': round(sum(r[0] for r in rows) / len(rows), 1),
                'allreduce_us_max': round(max(r[0] for r in rows), 1),
                'lane_wait_us_mean': round(sum(r[1] for r in rows) / len(rows), 1),
                'lane_wait_us_max': round(max(r[1] for r in rows), 1),
                'elements_mean': int(sum(r[2] for r in rows) / len(rows))}

    def replicas_in_sync(self, model, tol=0.0):
        """Debug check: every rank holds the same weights (host-side compare over torch.distributed)."""
        mine = model.pack.value.t.detach().cpu().clone()
        if not dist.is_initialized() or self.world == 1:
            return True
        ref = mine.clone()
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(ref, src=src, group=self.group)
        diff = (mine - ref).abs().max()
        dist.all_reduce(diff, op=dist.ReduceOp.MAX, group=self.group)
        return float(diff.item()) <= tol

    def close(self):
        for model in self.models:
            model.grad_sync = None
            model.bucket_hook = None
        self.comm.close()
