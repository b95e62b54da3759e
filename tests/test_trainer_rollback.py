"""Host logic of the epoch loop (`Trainer.train`), without a GPU: learning-rate decay, NaN rollback to the last
weights (up to 9 times with a shrinking lr) and then to the weights held before the first epoch, the error
without an optimizer -- the sequence of reference my_model/trainer.py:258-286 (`reload_attempts`,
`lr *= lr_step ** reload_attempts`, `last_weights` / `best_weights`, `continue` without advancing the epoch).
The models are stand-ins that script `nan_weights()`; the expectation is computed by a literal transcription
of the reference's rules below."""
import numpy as np
import pytest


class ScriptedModel:
    """nan_weights() follows a script (one entry per finished epoch attempt); weights are a counter that every
    train pass increments, so 'which weights were restored' is visible."""

    def __init__(self, name, log, nan_script):
        self.name, self.log, self.nan_script = name, log, list(nan_script)
        self.weights = 0

    def get_outputs_count(self):
        return 1

    def nan_weights(self):
        return self.nan_script.pop(0) if self.nan_script else False

    def get_weights(self):
        return {self.name: {'w': self.weights}}

    def set_weights(self, source):
        self.weights = source[self.name]['w']
        self.log.append(('restore', self.weights))


class ScriptedSystem:
    def __init__(self, model, log):
        self.model, self.log = model, log

    def train(self, context):
        self.model.weights += 1
        context['losses'] = {self.model.name: {'output_losses': [1.0 / (1 + self.model.weights)]}}

    def test(self, context):
        context['losses'] = {self.model.name: {'output_losses': [1.0 / (1 + self.model.weights)]}}


class Pages:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def get(self, index, **kwargs):
        return {}


class Opt:
    def __init__(self, lr):
        self.lr = lr


def reference_sequence(nan_script, num_epochs, lr, lr_step, n_train):
    """The reference's rules, transcribed: returns (lr after every epoch attempt, restore events, final weights)."""
    script = list(nan_script)
    weights, last, best = 0, 0, 0          # best_weights = last_weights = get_weights() before the first epoch
    attempts, epoch = 0, 1
    lrs, restores = [], []
    while epoch <= num_epochs:
        weights += n_train                 # the train pass
        attempts += 1
        lr *= lr_step ** attempts
        lrs.append(lr)
        if script.pop(0) if script else False:
            if attempts < 10:
                weights = last
            else:
                weights = best
                attempts = 0
            restores.append(weights)
            continue
        last = weights
        epoch += 1
        attempts = 0
    return lrs, restores, weights


@pytest.mark.parametrize('nan_script', [
    [False, True, True, False, False],                 # two rollbacks to the last weights in epoch 2
    [False] + [True] * 10 + [False, False],            # the tenth NaN in a row reloads the initial ('best') weights
    [True, False, True, False],
])
def test_nan_rollback_follows_the_reference_sequence(nan_script, capsys):
    from univer_ocr_amd.my_model.trainer import Trainer
    log = []
    model = ScriptedModel('Net', log, nan_script)
    system = ScriptedSystem(model, log)
    opt = Opt(0.01)
    lrs = []

    class Spy(Trainer):
        def _pass(self, dataset, order, sink, train, epoch, stage):
            if train:
                lrs.append(None)
            super()._pass(dataset, order, sink, train, epoch, stage)

    trainer = Spy(system, lambda get, args: get(*args), {'Net': model}, Pages(3), Pages(1), optimizer=opt,
                  learning_rate_step=0.9)
    # record the lr after every attempt: nan_weights() is called right before the decay is applied
    seen = []
    orig = model.nan_weights

    def spy_nan():
        seen.append(None)
        return orig()
    model.nan_weights = spy_nan
    num_epochs = 3
    exp_lrs, exp_restores, exp_weights = reference_sequence(nan_script, num_epochs, 0.01, 0.9, 3)
    trainer.train(num_epochs)
    assert len(seen) == len(exp_lrs)                              # one NaN test per epoch attempt
    assert opt.lr == pytest.approx(exp_lrs[-1], rel=1e-12)
    assert [w for kind, w in log if kind == 'restore'] == exp_restores
    assert model.weights == exp_weights
    out = capsys.readouterr().out
    assert out.count('loading last weights') == sum(1 for i, w in enumerate(exp_restores)) - out.count('last best weights')


def test_nan_without_optimizer_raises_like_the_reference():
    from univer_ocr_amd.my_model.trainer import Trainer
    log = []
    model = ScriptedModel('Net', log, [True])
    trainer = Trainer(ScriptedSystem(model, log), lambda get, args: get(*args), {'Net': model}, Pages(2), Pages(1))
    with pytest.raises(ValueError, match='NaN value found in weights, but no optimizer provided'):
        trainer.train(1)


def test_losses_table_picks_better_weights_like_the_reference():
    """Losses.get_better_weights (reference my_model/trainer.py:98-110): a model is 'better' when the mean of its
    validation losses dropped below its best so far, or when the best is NaN and the new value is not."""
    from univer_ocr_amd.my_model.trainer import Losses
    losses = Losses(['A', 'B'], {'A': 1, 'B': 2})
    losses.reset()
    losses.validation({'A': {'output_losses': [2.0]}, 'B': {'output_losses': [1.0, 3.0]}})
    losses.train({'A': {'output_losses': [2.0]}, 'B': {'output_losses': [1.0, 3.0]}})
    losses.normalize(1, 1)
    assert losses.get_better_weights(1) == ['A', 'B']            # anything beats the initial inf
    losses.next()
    losses.reset()
    losses.validation({'A': {'output_losses': [2.5]}, 'B': {'output_losses': [0.5, 3.0]}})
    losses.train({'A': {'output_losses': [2.5]}, 'B': {'output_losses': [0.5, 3.0]}})
    losses.normalize(1, 1)
    assert losses.get_better_weights(2) == ['B']
    assert losses.best_loss_epoch == {'A': 1, 'B': 2}
    assert losses.val_best_losses['A'] == [2.0] and losses.val_best_losses['B'] == [0.5, 3.0]
    losses.val_best_losses['A'] = [float('nan')]
    assert 'A' in losses.get_better_weights(3)
