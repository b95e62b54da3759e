"""Data-parallel path on CPU: world_size 2, gloo backend (on GPUs the same queueing / ordering code drives the
RCCL entry points of the C ABI).  Without a GPU the DeviceArrays are storage-only host tensors, so this covers
what is host logic: the packs re-homed into one flat buffer, weight broadcast at construction, the
flat-gradient all-reduce, SUM vs MEAN semantics per loss type, deferred (overlapped) completion, the
rank-independent issue order of the collectives, coalescing into one collective, the `Trainer` epoch loop with
data parallelism, and the replica-consistency check."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results, coalesce):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from univer_ocr_amd.my_model.model import make_char, make_monochrome, make_paragraph
        from univer_ocr_amd.nn.optimizers import Momentum
        from univer_ocr_amd.parallel import DataParallel, mean_type_loss

        np.random.seed(100 + rank)                    # different initial weights on every rank
        mono = make_monochrome((2, 8, 8, 1), Momentum(lr=0.1))
        para = make_paragraph((2, 16, 16, 1), Momentum(lr=0.1))
        char = make_char((1, 32, 8, 1), Momentum(lr=0.1))
        before = mono.pack.value.t.clone()
        views_before = {n: p.value.numpy().copy() for n, p in char.params().items()}
        dp = DataParallel({'Monochrome': mono, 'Paragraph': para, 'Char': char}, overlap=True, coalesce=coalesce)
        out = {'rank': rank, 'backend': dp.backend}
        # 0. the packs now live in ONE flat buffer, parameters are views of it, values survived the move
        out['one_buffer'] = (mono.pack.value.t.data_ptr() == dp.flat_value.t.data_ptr() and
                             char.pack.grad.t.data_ptr() == dp.flat_grad.t[dp._slice[id(char)][0]:].data_ptr())
        char.params()['Char/dense_block/dense_3/w'].value.t[0, 0] = 123.0
        out['views'] = float(dp.flat_value.t[dp._slice[id(char)][0] + [o for p, o, s in char.pack.entries
                                                                      if p is char.params()['Char/dense_block/dense_3/w']][0]]) == 123.0
        if rank == 0:
            out['kept'] = all(np.array_equal(views_before[n][1:], p.value.numpy()[1:]) for n, p in char.params().items())
        # 1. replicas start identical (rank 0's weights), and they were different before
        out['sync0'] = all(dp.replicas_in_sync(m) for m in (mono, para, char))
        out['changed'] = bool((before != mono.pack.value.t).any().item()) if rank != 0 else True
        assert not mean_type_loss(mono) and mean_type_loss(char)
        # 2. Dice nets SUM the gradients, immediate completion
        mono.pack.grad.t.fill_(float(rank + 1))
        mono.defer_grad_sync = False
        mono.grad_sync(mono)
        out['mono_grad'] = float(mono.pack.grad.t[0].item())              # 1 + 2 = 3
        out['after_one'] = dp.collectives
        # 3. deferred completion: nothing is issued until somebody waits; then everything queued goes out in
        # CONSTRUCTION order (Monochrome, Paragraph, Char) although it was queued Char, Monochrome, Paragraph;
        # SoftmaxCE nets average
        for m in (mono, para, char):
            m.defer_grad_sync = True
        char.pack.grad.t.fill_(float(10 * (rank + 1)))
        mono.pack.grad.t.fill_(float(rank + 1))
        para.pack.grad.t.fill_(float(5 * (rank + 1)))
        char.grad_sync(char)
        mono.grad_sync(mono)
        para.grad_sync(para)
        out['queued'] = [m.name if hasattr(m, 'name') else None for m, _ in dp._queue] and len(dp._queue) == 3
        out['not_yet'] = dp.collectives == out['after_one']
        issued = []
        real = dp.comm.all_reduce
        dp.comm.all_reduce = lambda arr: (issued.append(arr.size), real(arr))[1]
        dp.wait(para)
        dp.comm.all_reduce = real
        out['issued_sizes'] = issued
        out['expected_sizes'] = ([mono.pack.total + para.pack.total + char.pack.total] if coalesce else
                                 [mono.pack.total, para.pack.total, char.pack.total])
        dp.wait(char)
        dp.wait(mono)
        out['char_grad'] = float(char.pack.grad.t[0].item())              # (10 + 20) / 2 = 15
        out['para_grad'] = float(para.pack.grad.t[0].item())              # 5 + 10 = 15
        out['mono_grad2'] = float(mono.pack.grad.t[-1].item())            # 3
        out['drained'] = not dp._queue and not dp._reduced
        # 4. a diverged replica is detected
        if rank == 1:
            mono.pack.value.t[0] += 1.0
        out['sync1'] = dp.replicas_in_sync(mono)
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('coalesce', [False, True])
def test_flat_gradient_allreduce_world2(coalesce):
    if torch.cuda.is_available():
        pytest.skip('CPU (gloo) rehearsal of the data-parallel logic')
    world, port = 2, _free_port()
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_worker, args=(world, port, results, coalesce), nprocs=world, join=True)
        results = dict(results)
    assert set(results) == {0, 1}
    assert results[0]['kept']
    for rank, out in results.items():
        assert out['backend'] == 'gloo' and out['one_buffer'] and out['views']
        assert out['sync0'] and out['changed']
        assert out['mono_grad'] == 3.0 and out['after_one'] == 1
        assert out['queued'] and out['not_yet']
        assert out['issued_sizes'] == out['expected_sizes']
        assert out['char_grad'] == 15.0 and out['para_grad'] == 15.0 and out['mono_grad2'] == 3.0
        assert out['drained']
        assert out['sync1'] is False


def _split_worker(rank, world, port, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from univer_ocr_amd.my_model.model import make_char, make_monochrome
        from univer_ocr_amd.nn.optimizers import Momentum
        from univer_ocr_amd.parallel import DataParallel

        mono = make_monochrome((2, 8, 8, 1), Momentum(lr=0.1))
        char = make_char((1, 32, 8, 1), Momentum(lr=0.1))
        dp = DataParallel({'Monochrome': mono, 'Char': char}, overlap=True)
        out = {}
        node = dp.split_node(char)
        lo_rel = dp._split[id(char)][1]
        names = {id(p): n for n, p in char.params().items()}
        tail = sorted(names[id(p)] for p, off, size in char.pack.entries if off >= lo_rel)
        head = sorted(names[id(p)] for p, off, size in char.pack.entries if off < lo_rel)
        out['node'], out['tail'], out['head'] = str(node), tail, head
        out['mono_unsplit'] = dp.split_node(mono) is None and mono.bucket_hook is None
        out['tail_share'] = (char.pack.total - lo_rel) / char.pack.total
        # the hook fires after EVERY node of the backward pass; only the split node issues a collective
        char.defer_grad_sync = mono.defer_grad_sync = True
        char.pack.grad.t.fill_(float(10 * (rank + 1)))
        mono.pack.grad.t.fill_(float(rank + 1))
        for other in reversed([n for n in char._toposort() if n != node]):
            char.bucket_hook(char, other)
        out['quiet'] = dp.collectives
        sizes = []
        real = dp.comm.all_reduce
        dp.comm.all_reduce = lambda arr: (sizes.append(arr.size), real(arr))[1]
        char.bucket_hook(char, node)                                       # tail out NOW: (10 + 20) / 2 = 15
        out['after_tail'] = (dp.collectives, float(char.pack.grad.t[-1].item()), float(char.pack.grad.t[0].item()))
        char.grad_sync(char)                                               # the rest is queued ...
        mono.grad_sync(mono)
        dp.wait(mono)                                                      # ... and goes out in construction order
        dp.wait(char)
        dp.comm.all_reduce = real
        out['sizes'] = sizes
        out['expected'] = [char.pack.total - lo_rel, mono.pack.total, lo_rel]
        out['final'] = (float(char.pack.grad.t[0].item()), float(char.pack.grad.t[-1].item()), float(mono.pack.grad.t[0].item()))
        out['drained'] = not dp._queue and not dp._reduced and not dp._tail_done
        # next step without the hook (a step that does not run the backward through Model.backward): one collective again
        char.pack.grad.t.fill_(float(rank + 1))
        sizes2 = []
        dp.comm.all_reduce = lambda arr: (sizes2.append(arr.size), real(arr))[1]
        char.grad_sync(char)
        dp.wait(char)
        dp.comm.all_reduce = real
        out['sizes2'], out['whole'] = sizes2, [char.pack.total]
        results[rank] = out
    finally:
        dist.destroy_process_group()


def test_gradient_tail_goes_out_early_world2():
    """Within-net overlap (parallel.DataParallel._bucket): the Char net's dense block -- the tail of its pack, 84 % of its
    parameters -- is reduced from Model.backward's bucket_hook as soon as it is final, the conv block with the net's
    regular grad_sync; MEAN semantics on both parts, the same collectives in the same order on both ranks."""
    if torch.cuda.is_available():
        pytest.skip('CPU (gloo) rehearsal of the data-parallel logic')
    world, port = 2, _free_port()
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_split_worker, args=(world, port, results), nprocs=world, join=True)
        results = dict(results)
    assert set(results) == {0, 1}
    assert results[0]['node'] == results[1]['node'] and results[0]['sizes'] == results[1]['sizes']
    for out in results.values():
        assert all('dense' in n for n in out['tail']) and all('conv' in n for n in out['head']), (out['tail'], out['head'])
        assert out['mono_unsplit'] and out['tail_share'] > 0.8
        assert out['quiet'] == 0
        assert out['after_tail'] == (1, 15.0, 10.0 * (out is results[0] and 1 or 2)) or out['after_tail'][:2] == (1, 15.0)
        assert out['sizes'] == out['expected']
        assert out['final'] == (15.0, 15.0, 3.0) and out['drained']
        assert out['sizes2'] == out['whole']


def _rank_mean_worker(rank, world, port, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from univer_ocr_amd.my_model.trainer import Losses, Trainer

        class DP:
            pass
        dp = DP()
        dp.rank, dp.world = rank, world
        trainer = Trainer(None, None, {}, [], [0], data_parallel=dp)
        losses = Losses(['A', 'B'], {'A': 1, 'B': 2})
        losses.reset()
        losses.train({'A': {'output_losses': [1.0 + rank]}, 'B': {'output_losses': [10.0 * (rank + 1), 2.0]}})
        losses.validation({'A': {'output_losses': [3.0 - rank]}, 'B': {'output_losses': [4.0, 8.0 * rank]}})
        losses.normalize(1, 1)
        trainer._rank_mean(losses)
        results[rank] = (losses.train_losses, losses.val_losses)
    finally:
        dist.destroy_process_group()


def test_trainer_averages_epoch_losses_over_ranks():
    """Trainer(data_parallel=...)._rank_mean: every rank ends an epoch with the SAME loss tables (the mean over the
    ranks), so `get_better_weights` and the NaN rollback take the same decisions everywhere."""
    world, port = 2, _free_port()
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_rank_mean_worker, args=(world, port, results), nprocs=world, join=True)
        results = dict(results)
    assert results[0] == results[1]
    train, val = results[0]
    assert train == {'A': [1.5], 'B': [15.0, 2.0]} and val == {'A': [2.5], 'B': [4.0, 4.0]}
