"""The reference's own test entry points, driven through the HIP backend on the GPU:
`test_nn.py test_gradients` (numeric gradient checks, float64) and `test_nn.py test_identity`
(float32 production kernels vs float64 generic kernels on 5x240x320x6), plus one tiny `train_model`
curriculum that writes and re-reads model_weights.json."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_test_gradients_script():
    from univer_ocr_amd.nn.test import test_gradients
    correct, total = test_gradients.main(True)
    assert total >= 35 and correct == total          # reference probe: 35/35 on its NumPy path


def test_test_identity_script():
    from univer_ocr_amd.nn.test import test_identity
    correct, total = test_identity.main(True)
    assert (correct, total) == (10, 10)


def test_train_model_curriculum(tmp_path, monkeypatch):
    from univer_ocr_amd.my_model import train as train_mod
    from univer_ocr_amd.nn import CP
    CP.set_dtype('float32')
    path = tmp_path / 'model_weights.json'
    monkeypatch.setattr(train_mod, 'MODEL_WEIGHTS_FILE_PATH', path)
    results = train_mod.train_model(True, epochs_scale=0.011, batch=1, height=32, width=64)
    assert set(results) == {'TRAIN_MONOCHROME', 'TRAIN_PARAGRAPH', 'TRAIN_PAGE'}
    weights = json.loads(path.read_text())
    assert 'Monochrome/conv_1' in weights and 'Char/dense_block/dense_3' in weights
    assert np.array(weights['Monochrome/conv_1']['w']).shape == (3, 3, 1, 16)
    assert np.array(weights['Char/dense_block/dense_1']['w']).shape == (513, 1024)
    assert all(np.isfinite(np.array(v)).all() for layer in weights.values() for v in layer.values())


def test_page_feeder_uint8_pipeline_matches_direct_copy():
    """uint8 upload on the copy stream + device-side conversion == CP.copy of the float layers."""
    from univer_ocr_amd.my_model.pipeline import PAGE_FEEDS, PageFeeder, to_uint8_layers
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    batches = [to_uint8_layers(make_page_batch(2, 32, 64, 16, seed=s)) for s in (1, 2, 3)]
    feeder = PageFeeder(batches[0])
    feeder.stage(batches[0])
    for i, batch in enumerate(batches):
        ctx = feeder.context()
        if i + 1 < len(batches):
            feeder.stage(batches[i + 1])
        for label, (tag, scale) in PAGE_FEEDS.items():
            expect = batch[tag].astype(np.float32) * np.float32(scale)
            got = CP.asnumpy(ctx[label])
            assert got.dtype == np.float32 and np.allclose(got, expect, rtol=1e-6, atol=0), label


def run_two_ranks(script_args, env, root, timeout):
    """`python -m torch.distributed.run --nproc-per-node 2 <script_args>` on a free rendezvous port of 127.0.0.1.  The port
    is found by binding and closing a socket, so something else can take it before torchrun does: that one failure
    (and only that: EADDRINUSE in stderr) gets a second port."""
    import socket
    import subprocess
    import sys
    out = None
    for _ in range(2):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', str(port), *script_args]
        out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)
        if out.returncode == 0 or not any(k in out.stderr for k in ('EADDRINUSE', 'ddress already in use')):
            break
    return out


def test_bench_two_rank_rehearsal_prints_one_valid_line():
    """`bench.py --gpus 2` launched the way the driver does (torch.distributed.run, one process per rank), with
    UOCR_BENCH_REHEARSAL=1 so that both ranks share the one card through gloo: rendezvous, data-parallel
    trainer, barrier / max-over-ranks timing and the rank-0 JSON line of the multi-rank path."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UOCR_BENCH_REHEARSAL='1')
    out = run_two_ranks([os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '4',
                         '--height', '64', '--width', '128'], env, root, 600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [line for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 3 and rec['config']['global_batch'] == 8
    assert rec['scaling'] == 'weak' and rec['value'] > 0 and 'cpu_baseline' not in rec
    assert all(np.isfinite(v).all() for v in map(np.array, rec['config']['final_losses'].values()))


def test_page_feeder_into_static_graph_inputs():
    """The upload pipeline converting straight into the arrays the HIP graphs read
    (PageFeeder.context(into=PageTrainer.static_inputs())) trains exactly like make_context() on the same
    uint8-quantised pages."""
    from univer_ocr_amd.my_model.pipeline import PageFeeder, to_uint8_layers
    from univer_ocr_amd.my_model.synthetic import make_page_batch
    from univer_ocr_amd.my_model.trainer import PageTrainer
    from univer_ocr_amd.nn import CP
    CP.use_gpu(0)
    CP.set_dtype('float32')
    lazy, CP.lazy_losses = CP.lazy_losses, True
    try:
        raw = [make_page_batch(2, 32, 64, 16, seed=s) for s in (21, 22, 23)]
        u8 = [to_uint8_layers(b) for b in raw]
        quantised = [{t: a.astype(np.float32) * (np.float32(1 / 255) if t in ('image', 'char_lines') else np.float32(1))
                      for t, a in b.items()} for b in u8]
        results = []
        for use_feeder in (False, True):
            trainer = PageTrainer(2, 32, 64, 16, optimizer='sgd', lr=0.01, seed=8, graphs=True, pipelined=True)
            trainer.capture(trainer.make_context(quantised[0]))
            rows = []
            if use_feeder:
                statics = trainer.static_inputs()
                feeder = PageFeeder(u8[0])
                feeder.stage(u8[0])
                for i in range(5):
                    trainer.join()
                    ctx = feeder.context(into=statics)
                    feeder.stage(u8[(i + 1) % 3])
                    losses = trainer.step(ctx)
                    rows.append({n: float(l['output_losses'][0]) for n, l in losses.items()})
            else:
                for i in range(5):
                    losses = trainer.step(trainer.make_context(quantised[i % 3]))
                    rows.append({n: float(l['output_losses'][0]) for n, l in losses.items()})
            trainer.join()
            results.append(rows)
        assert results[0] == results[1]
    finally:
        CP.lazy_losses = lazy


def test_train_py_data_parallel_two_ranks(tmp_path):
    """`torchrun --nproc-per-node 2 train.py`: the reference's step loop (my_model/trainer.py:213-233 ->
    nn/model_system.py:104-118 -> nn/models.py:250-254) trained data-parallel -- every rank its own pages, the
    gradients all-reduced inside Model.train, epoch losses averaged over the ranks, rank 0 writing
    model_weights.json.  Two ranks share the one card here, so the gradient exchange goes through gloo
    (UOCR_DP_BACKEND=gloo); on a node with one GPU per rank the same code path runs the RCCL entry points."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    weights_path = tmp_path / 'model_weights.json'
    env = dict(os.environ, UOCR_DP_BACKEND='gloo', UOCR_WEIGHTS=str(weights_path), UOCR_EPOCHS_SCALE='0.011',
               UOCR_TRAIN_PAGE='32x64', UOCR_DUMP_FINAL_WEIGHTS=str(tmp_path / 'final'))
    out = run_two_ranks([os.path.join(root, 'train.py'), 'True'], env, root, 900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    weights = json.loads(weights_path.read_text())
    assert 'Monochrome/conv_1' in weights and 'Char/dense_block/dense_3' in weights
    # the replicas ended with identical weights (each rank dumps its final ones)
    finals = [np.load(str(tmp_path / f'final.rank{r}.npz')) for r in (0, 1)]
    assert set(finals[0].files) == set(finals[1].files) and len(finals[0].files) > 20
    for key in finals[0].files:
        assert np.array_equal(finals[0][key], finals[1][key]), key
