"""Data-parallel semantics on real kernels: two ranks share the one GPU of the test box (gloo
backend, gradients staged through the host -- RCCL refuses two ranks on one device), each trains
on its half of a batch with the overlapped PageTrainer step; the result must equal one process
training on the whole batch:
  * Dice nets: gradients are SUMMED over ranks (their loss sums over the batch),
  * Char (softmax CE): gradients are AVERAGED (its loss divides by the local batch),
  * L2 is applied once, after the all-reduce,
  * replicas hold identical weights after every step."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BATCH, H, W, CW = 4, 32, 64, 16


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _slice(layers, lo, hi, cw):
    out = {}
    for tag, arr in layers.items():
        out[tag] = arr[lo * cw:hi * cw] if tag == 'char_labels' else arr[lo:hi]
    return out


def _worker(rank, world, port, results, graphs, steps, pipelined):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from univer_ocr_amd.my_model.synthetic import make_page_batch
        from univer_ocr_amd.my_model.trainer import PageTrainer
        from univer_ocr_amd.nn import CP
        CP.use_gpu(0)
        CP.set_dtype('float64')
        CP.lazy_losses = graphs
        per = BATCH // world
        trainer = PageTrainer(per, H, W, CW, optimizer='sgd', lr=0.01, seed=5 + rank, overlap=True, graphs=graphs,
                              pipelined=pipelined)
        assert trainer.dp is not None
        layers = make_page_batch(BATCH, H, W, CW, seed=77)
        context = trainer.make_context(_slice(layers, rank * per, (rank + 1) * per, CW))
        for _ in range(steps):
            trainer.step(context)
        trainer.join()
        assert (trainer._captured is not None) == graphs
        ok = all(trainer.dp.replicas_in_sync(m) for m in trainer.models.values())
        weights = {n: p.value.numpy() for m in trainer.models.values() for n, p in m.params().items()}
        results[rank] = (ok, weights)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('graphs,steps,pipelined', [(False, 2, False), (True, 5, False), (False, 4, True), (True, 6, True)])
def test_two_ranks_equal_one_process_on_the_whole_batch(graphs, steps, pipelined):
    """graphs=True: steps 3-5 replay the per-net HIP graphs with the all-reduce issued between them;
    pipelined=True: no per-step join of the lanes (the bench default); both: graph replay on free-running lanes."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    with ctx.Manager() as manager:
        results = manager.dict()
        mp.spawn(_worker, args=(world, port, results, graphs, steps, pipelined), nprocs=world, join=True)
        results = dict(results)
    assert results[0][0] and results[1][0], 'replicas diverged'
    for name in results[0][1]:
        assert np.array_equal(results[0][1][name], results[1][1][name])

    # single process, whole batch, same initial weights (rank 0's seed) -> same result
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float64')
    try:
        trainer = PageTrainer(BATCH, H, W, CW, optimizer='sgd', lr=0.01, seed=5, data_parallel=False)
        context = trainer.make_context(make_page_batch(BATCH, H, W, CW, seed=77))
        for _ in range(steps):
            trainer.step(context)
        for model in trainer.models.values():
            for name, p in model.params().items():
                ref, got = p.value.numpy(), results[0][1][name]
                err = np.max(np.abs(ref - got)) / max(1e-30, np.max(np.abs(ref)))
                assert err <= 1e-11, f'{name}: data-parallel result differs from the single-process one: {err:.2e}'
    finally:
        CP.set_dtype('float32')
