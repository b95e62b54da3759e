"""`train_model` -- the curriculum driver behind `python train.py` (reference: my_model/train.py:67-289).

Stages (mode, lr, lr_step, epochs) train one net at a time and then all of them, each stage resuming
from and saving to `model_weights.json` (flat {"<layer name>": {"<param>": nested list}} written with
compact separators, train.py:132-141).  The reference's stages that chain nets through the host
crop/rotate code (TRAIN_LINE / TRAIN_CHAR / TRAIN_ALL) are replaced by TRAIN_PAGE, which trains all
four nets on device-resident page and line-strip batches.  Data: seeded synthetic pages
(my_model/synthetic.py); the reference's generate_data needs Windows fonts and stays outside."""
import json
import os
from pathlib import Path

from ..nn.gpu import CP
from ..nn.optimizers import Adam
from ..nn.progress_tracker import ProgressTracker
from .model import Modes, make_context_maker, make_model_system
from .synthetic import SyntheticPages
from .trainer import Trainer

MODEL_WEIGHTS_FILE_PATH = Path(os.environ.get(
    'UOCR_WEIGHTS', Path(__file__).resolve().parent / 'model_weights.json'))


def message(*args):
    print(*args, flush=True)


def load_weights(path=MODEL_WEIGHTS_FILE_PATH):
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        print('No model_weights.json file found')
        return {}


def _ipc_env():
    """RCCL shares device buffers between the ranks' processes through dmabuf IPC (set before the GPU is touched)."""
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def _init_data_parallel():
    """Launched by torch.distributed.run (one rank per GPU)?  Then torch.distributed -- gloo: host-side
    rendezvous only -- carries the RCCL id and the epoch losses; the gradients travel through the C ABI's RCCL
    entry points (parallel.DataParallel).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return 0, 1, None
    _ipc_env()
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    rank = int(os.environ.get('RANK', '0'))
    if not dist.is_initialized():
        dist.init_process_group('gloo', rank=rank, world_size=world)
    return rank, world, int(os.environ.get('LOCAL_RANK', '0'))


def train_model(use_gpu=True, show_progress_bar=False, save_train_progress=False, epochs_scale=None,
                batch=1, height=None, width=None, dp_backend=None):
    if not use_gpu:
        CP.use_cpu()          # raises: the NumPy path is the reference itself
    rank, world, local_rank = _init_data_parallel()
    if height is None or width is None:              # UOCR_TRAIN_PAGE=HxW: page size of the synthetic data set
        height, width = (int(v) for v in os.environ.get('UOCR_TRAIN_PAGE', '256x512').split('x'))
    if local_rank is not None and os.environ.get('UOCR_DP_BACKEND') == 'gloo':
        import torch
        local_rank %= max(1, torch.cuda.device_count())      # rehearsal: the ranks share the cards there are
    CP.use_gpu(local_rank)
    info = CP.runtime().device_info()
    message(f'Using GPU\nname = {info["name"]}\nmultiProcessorCount = {info["cu_count"]}\n'
            f'totalGlobalMem = {info["hbm_bytes"]}\nwarpSize = 64\n')
    tracker = ProgressTracker(lambda *a: None)
    scale = float(os.environ.get('UOCR_EPOCHS_SCALE', '0.02')) if epochs_scale is None else epochs_scale
    stages = [(Modes.TRAIN_MONOCHROME, 0.0015, 0.995, 100), (Modes.TRAIN_PARAGRAPH, 0.0015, 0.995, 100),
              (Modes.TRAIN_PAGE, 0.001, 0.9, 10)]
    # data parallel: every rank trains on its own pages (the batch is sharded by rank: seeds differ)
    train_set = SyntheticPages(batch, height, width, seed=1234 + 1000 * rank, length=4)
    val_set = SyntheticPages(batch, height, width, seed=9999 + 1000 * rank, length=2)
    results = {}
    watchdog = None
    for mode, lr, lr_step, epochs in stages:
        epochs = max(1, int(round(epochs * scale)))
        message(f'Training mode: {mode.name}')
        weights = load_weights()
        optimizer = Adam(lr=lr)
        input_shape = (batch, height, width, 1)
        model_system, models, names = make_model_system(input_shape, optimizer, tracker, weights, mode=mode,
                                                        char_input_shape=(batch, 32, 64, 1))
        message(f'Input shape: {input_shape}; parameters: {sum(m.count_parameters() for m in models.values())}')

        def save_weights(better, models=models):
            merged = load_weights() if os.path.exists(MODEL_WEIGHTS_FILE_PATH) else {}
            for name, model in models.items():
                if name in better:
                    merged.update(model.get_weights())
            with open(MODEL_WEIGHTS_FILE_PATH, 'w') as f:
                json.dump(merged, f, separators=(',', ':'))
        dp = None
        if world > 1:
            from ..parallel import DataParallel
            from ..watchdog import Watchdog
            if watchdog is None:                  # a rank stuck in a collective must not hang the whole job silently
                watchdog = Watchdog(float(os.environ.get('UOCR_STEP_TIMEOUT', '600')), rank, 'train watchdog')
            # rank 0's weights everywhere; UOCR_DP_BACKEND=gloo: several ranks on one card (rehearsal)
            dp = DataParallel(models, overlap=False, backend=dp_backend or os.environ.get('UOCR_DP_BACKEND'))
        trainer = Trainer(model_system, make_context_maker(mode), models, train_set, val_set, tracker,
                          show_progress_bar, optimizer, lr_step, save_weights, data_parallel=dp, watchdog=watchdog)
        results[mode.name] = trainer.train(epochs)
        if dp is not None:
            dp.close()
        final_models = models
    dump = os.environ.get('UOCR_DUMP_FINAL_WEIGHTS')     # tests: every rank's final weights, to compare replicas
    if dump:
        import numpy as np
        np.savez(f'{dump}.rank{rank}.npz', **{n: p.value.numpy() for m in final_models.values()
                                              for n, p in m.params().items()})
    return results
