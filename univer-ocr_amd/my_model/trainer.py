"""Train-step drivers.

`PageTrainer`  -- one step = every net of the model system trained on one device-resident batch
                  (reference step loop: my_model/trainer.py:215-232 -> nn/model_system.py:154-157
                  -> nn/models.py:250-254), with optional data parallelism (parallel.DataParallel)
                  whose per-model gradient all-reduce overlaps the next model's forward/backward.
`Trainer`      -- the reference's epoch loop (my_model/trainer.py:146-296): train / validation
                  passes, lr decay, NaN rollback to the last (then best) weights, best-weights
                  callback.  Host orchestration only.
"""
import numpy as np
import torch.distributed as dist

from ..nn.gpu import CP
from ..nn.optimizers import Adam, Momentum
from .model import CHAR_INPUT_HEIGHT, Modes, make_model_system


def make_optimizer(name, lr):
    if name == 'sgd':
        return Momentum(lr=lr, momentum=0)          # BASELINE config 3 "SGD" (optimizers.py:67-81)
    if name == 'adam':
        return Adam(lr=lr)                          # the reference trainer's choice (train.py:127)
    raise ValueError(f'unknown optimizer {name}')


class PageTrainer:
    def __init__(self, batch, height=256, width=512, char_width=64, optimizer='sgd', lr=0.0015, seed=0,
                 nets=('Monochrome', 'Paragraph', 'Line', 'Char'), data_parallel=None, overlap=True,
                 init='kaiming_normal', fuse=True):
        np.random.seed(seed)                        # kaiming_uniform draws from the NumPy global RNG
        self.batch = batch
        self.optimizer = make_optimizer(optimizer, lr)
        self.model_system, self.models, self.names = make_model_system(
            (batch, height, width, 1), self.optimizer, mode=Modes.TRAIN_PAGE,
            char_input_shape=(batch, CHAR_INPUT_HEIGHT, char_width, 1))
        if init == 'kaiming_normal':
            self.reinit_weights(seed)
        keep = [c for c in self.model_system.components if c.name in nets]
        self.model_system.components = keep
        self.models = {n: m for n, m in self.models.items() if n in nets}
        for model in self.models.values():
            model.enable_fusion(fuse)            # conv + LeakyReLU / Sigmoid as one forward kernel
        self.dp = None
        if data_parallel is None:
            data_parallel = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if data_parallel:
            from ..parallel import DataParallel
            self.dp = DataParallel(self.models, overlap=overlap)
            for model in self.models.values():
                model.defer_grad_sync = overlap

    def reinit_weights(self, seed):
        """Zero-mean He initialisation (the reference's kaiming_normal, initializers.py:16-19) drawn from a
        seeded generator.  The reference's DEFAULT kaiming_uniform is non-negative (U[0,1) scaled), with
        which the Char net's logits reach ~1e6 and plain SGD diverges to NaN within a few steps -- in the
        reference as well; init='reference' keeps that behaviour."""
        rng = np.random.default_rng(seed)
        for model in self.models.values():
            for layer in model.layers.values():
                params = layer.params()
                if not params:
                    continue
                w = params['w']
                fan_in = int(np.prod(w.value.shape[:-1])) + (1 if 'b' in params else 0)
                scale = np.sqrt(2.0 / fan_in)
                for p in params.values():
                    p.value = rng.standard_normal(p.value.shape) * scale

    def make_context(self, layers):
        """Host layer dict (synthetic.make_page_batch) -> device-resident context."""
        mapping = {'monochrome_X': 'image', 'monochrome_y': 'monochrome',
                   'paragraph_X': 'monochrome', 'paragraph_y': 'paragraph',
                   'line_X': 'monochrome', 'line_y': 'line',
                   'char_X': 'char_lines', 'char_y': 'char_labels'}
        cache = {}
        context = {}
        for label, tag in mapping.items():
            if tag not in cache:
                cache[tag] = CP.copy(layers[tag])
            context[label] = cache[tag]
        return context

    def step(self, context):
        """Train every net once.  Returns {net: {'output_losses': [...], 'regularization_loss': r}}."""
        comps = self.model_system.components
        if self.dp is None or not self.dp.overlap:
            self.model_system.train(context)
            return context['losses']
        context['losses'] = {}
        for comp in comps:                          # enqueue fwd+bwd of every net, all-reduces in flight
            comp.selector(context)
            X, y = next(comp.selector.get())
            comp.model.train_begin(X, y)
            comp._publish()
        for comp in comps:                          # then finish: wait, L2, optimizer
            context['losses'][comp.name] = comp.model.train_finish()
        return context['losses']

    def forward(self, context):
        self.model_system.predict({**context})
