"""Train-step drivers.

`PageTrainer`  -- one step = every net of the model system trained on one device-resident batch
                  (reference step loop: my_model/trainer.py:215-232 -> nn/model_system.py:154-157
                  -> nn/models.py:250-254), with optional data parallelism (parallel.DataParallel)
                  whose per-model gradient all-reduce overlaps the next model's forward/backward.
`Trainer`      -- the reference's epoch loop (my_model/trainer.py:146-296): train / validation
                  passes, lr decay, NaN rollback to the last (then best) weights, best-weights
                  callback.  Host orchestration only.
"""
import os

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


class GraphSequence:
    """The graphs of one net's forward + loss + backward, in order (two when its gradient tail is reduced early)."""

    def __init__(self, parts):
        self.parts = list(parts)

    def replay(self):
        for part in self.parts:
            part.replay()


class PageTrainer:
    def __init__(self, batch, height=256, width=512, char_width=64, optimizer='sgd', lr=0.0015, seed=0,
                 nets=('Monochrome', 'Paragraph', 'Line', 'Char'), data_parallel=None, overlap=True,
                 dp_coalesce=False, dp_backend=None,
                 init='kaiming_normal', fuse=True, lanes=True, input_grads=True, graphs=False, eager_nets=(), pipelined=False,
                 snapshot_losses=True,
                 lane_groups=(('Monochrome', 'Paragraph'), ('Line',), ('Char',)), lane_xcds=None, side_wgrad=None,
                 group_wgrad=None):
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
            model.enable_fusion(fuse, windows=os.environ.get('UOCR_WINDOWS_FUSION', '1') != '0')   # conv + LeakyReLU / Sigmoid as one forward kernel
            model.skip_input_grads(not input_grads)   # False: drop the page-input gradient nobody reads
        # side_wgrad: nets whose weight-gradient kernels run on a side stream of their lane (Runtime.side): the dX chain of
        # the backward pass does not wait for them.  Measured (tools/dev/side_ab.sh): the Char net
        # ALONE gains (0.433 -> 0.397 ms/step), but the page step with its three lanes loses badly (0.845 -> 1.18 ms with
        # the Char net forked, 2.3 ms with all nets): every fork adds a stream to the captured graphs and the device
        # time-slices the hardware queues beyond the few it runs at once.  Off by default; results are bit-identical.
        if side_wgrad is None:
            side_wgrad = tuple(n for n in os.environ.get('UOCR_SIDE_WGRAD', '').split(',') if n)
        for name, model in self.models.items():
            model.side_wgrad = CP.has_device() and (name in side_wgrad or 'all' in side_wgrad)
        # group_wgrad: nets whose backward pass runs inside Runtime.defer_wgrad (uocr_wgrad_defer_*): the finish kernels of
        # the weight-gradient producers (five per step in the Paragraph and Line nets) become ONE launch at the end of the
        # pass, and so do the small weight-gradient GEMMs (the Char net's five).  Default: every net
        # (under data parallelism only the Char net: in the one-rank RCCL rehearsal the page nets' late finish launches cost
        # more than they save -- 0.865 ms/step with 'Char', 0.880 with 'all', 0.905 with none; without it 0.835 / 0.822 / 0.852)
        if data_parallel is None:
            data_parallel = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if group_wgrad is None:
            group_wgrad = tuple(n for n in os.environ.get('UOCR_GROUP_WGRAD', 'Char' if data_parallel else 'all').split(',') if n)
        for name, model in self.models.items():
            model.group_wgrad = CP.has_device() and (name in group_wgrad or 'all' in group_wgrad)
        # one stream (lane) per net: the nets are independent until the optimizer step
        self.lanes = None
        if lanes and CP.has_device() and len(self.models) > 1:
            rt = CP.runtime()
            # (high-priority streams for the latency-bound nets were tried: 1.44 -> 2.54 ms/step)
            # Nets of one group share a lane and run one after the other in it.  Default: 3 lanes of about
            # equal length (Monochrome + Paragraph ~ Line ~ Char): the GPU keeps 4 hardware queues running at
            # a time, and with 4 lanes the 5th stream that has work (main in the joined mode, RCCL's stream
            # under data parallelism) gets time-sliced against them (1.3 -> 2.6 ms/step measured)
            self.lanes = {}
            # lane_xcds: one tuple of XCD numbers per lane group -- CU-partitioned lanes (uocr_ctx_create_cu_mask)
            for gi, group in enumerate(lane_groups or [(name,) for name in self.models]):
                members = [name for name in group if name in self.models]
                if members:
                    lane = rt.add_lane(xcds=None if not lane_xcds else lane_xcds[gi])
                    self.lanes.update({name: lane for name in members})
            for name in self.models:                   # nets not named in any group: a lane each
                if name not in self.lanes:
                    self.lanes[name] = rt.add_lane()
        # HIP graphs: each net's forward+loss+backward and its L2+optimizer tail are captured once and
        # replayed (two hipGraphLaunch per net and step instead of ~40 launches); needs lanes
        self.graphs = bool(graphs) and self.lanes is not None
        # pipelined: step() does not end with the main stream waiting for every lane, so a lane starts its
        # next step as soon as ITS previous step is done instead of when the slowest net is; results are
        # the same, losses / outputs are read after their lane's event (DeviceScalar.ready) or join().
        # Works best with one hardware queue per stream: GPU_MAX_HW_QUEUES=8 in the environment before the
        # first GPU call (bench.py sets it; ROCm's default 4 makes two nets share a queue)
        self.pipelined = bool(pipelined) and self.lanes is not None
        # graphs: the captured step writes its losses into static slots that the NEXT replay overwrites; with
        # snapshot_losses the step's last kernel copies them into the next row of a ring of 64 (LossArena) so that the
        # scalars it returns keep the value of THEIR step however late they are read
        self.snapshot_losses = bool(snapshot_losses)
        self._lane_done = {}
        self._event_rings = {}
        self.eager_nets = tuple(eager_nets)          # nets kept out of the graphs (e.g. to time one kernel)
        self._captured = None
        self._eager_override = set()
        self._eager_steps = 0
        self._input_buffers = {}
        self.dp = None
        if data_parallel is None:
            data_parallel = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if data_parallel:
            from ..parallel import DataParallel
            # construction order = the order the collectives of a step are issued in on every rank = the order the
            # nets' backward passes finish in the concurrent step (tools/lane_times.py: Paragraph 0.53 ms, Char 0.84,
            # Monochrome 0.88, Line 0.90), so no reduction waits behind a net that is not ready yet
            order = [n for n in ('Paragraph', 'Char', 'Monochrome', 'Line') if n in self.models]
            order += [n for n in self.models if n not in order]
            self.dp = DataParallel({n: self.models[n] for n in order}, overlap=overlap, coalesce=dp_coalesce,
                                   backend=dp_backend)
            self.dp_backend = self.dp.backend
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
        if self.graphs:
            # graph replay reads fixed addresses: the trainer owns one input buffer per layer tag and
            # every make_context refills them in place (no device-to-device copy in step())
            import torch
            self.join()
            for tag in set(mapping.values()):
                host = np.ascontiguousarray(layers[tag], dtype=CP.dtype)
                if tag not in self._input_buffers:
                    self._input_buffers[tag] = CP.copy(host)
                else:
                    self._input_buffers[tag].t.copy_(torch.from_numpy(host), non_blocking=False)
            return {label: self._input_buffers[tag] for label, tag in mapping.items()}
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
        if self.lanes is not None:
            return self._step_lanes(context)
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

    def _lane_order(self):
        """Enqueue order of the nets: the nets that are long chains of short kernels first (their chain is the
        critical path of the concurrent step), the nets of few long kernels last.  (The order the gradient
        all-reduces are ISSUED in under data parallelism is independent of this: parallel.DataParallel.flush.)"""
        rank = {'Char': 0, 'Paragraph': 1, 'Line': 2, 'Monochrome': 3}
        return sorted(self.model_system.components, key=lambda comp: rank.get(comp.name, 9))

    def _step_lanes(self, context):
        """Every net on its own stream.  The lanes start after everything already queued on the main
        stream (the inputs) and the main stream ends the step by waiting for all of them.  With data
        parallelism each net's all-reduce is issued from its lane right after its backward."""
        if self.graphs:
            if self._captured is None and self._eager_steps >= 2:     # lazy initialisation is over
                self._capture(context)
            if self._captured is not None:
                return self._step_graphs(context)
            self._eager_steps += 1
        rt = CP.runtime()
        start = self._event('start').record()              # (on the main stream: after the inputs queued there)
        context['losses'] = {}
        comps = self._lane_order()
        for comp in comps:
            with rt.lane(self.lanes[comp.name]) as stream:
                start.wait()
                comp.selector(context)
                X, y = next(comp.selector.get())
                if self.pipelined:                     # the caller may drop these arrays before the lane is done
                    X.t.record_stream(stream)
                    y.t.record_stream(stream)
                comp.model.train_begin(X, y)
                comp._publish()
        for comp in comps:
            with rt.lane(self.lanes[comp.name]):
                context['losses'][comp.name] = comp.model.train_finish()
                done = self._event(comp.name).record()
            self._finish_lane(comp.name, done, context['losses'][comp.name])
        return context['losses']

    def _event(self, key):
        """Events of the C ABI, a ring of four per key: an event that is recorded again still stands for "at least
        as late as" to whoever kept it from an earlier step (DeviceScalar.ready)."""
        ring = self._event_rings.setdefault(key, [[], 0])
        if len(ring[0]) < 4:
            ring[0].append(CP.runtime().event())
        ring[1] = (ring[1] + 1) % 4
        return ring[0][min(ring[1], len(ring[0]) - 1)]

    def _finish_lane(self, name, done, losses):
        if not self.pipelined:
            done.wait()                                    # the main stream waits for the lane
            return
        self._lane_done[name] = done
        from ..nn.gpu import DeviceScalar
        for value in list(losses['output_losses']) + [losses['regularization_loss']]:
            if isinstance(value, DeviceScalar):
                value.ready = done

    def join(self):
        """Pipelined mode: make the current stream wait for every lane (before reading outputs or
        overwriting inputs on it)."""
        for done in self._lane_done.values():
            done.wait()
        self._lane_done = {}

    # -- HIP-graph replay of the step ----------------------------------------------------------------
    def _capture(self, context):
        """Record, per net, graph A = forward + loss + backward and graph B = L2 + optimizer + gradient
        reset on the net's lane through the C ABI (uocr_graph_begin_capture / end_capture; replay =
        uocr_graph_launch).  The gradient all-reduce of data parallelism stays outside the graphs (issued
        eagerly between A and B).  The device arrays of `context` become the static inputs of the graphs; the loss
        slots (consecutive in one LossArena per net) and the published outputs are static outputs.  PyTorch only
        provides the memory: one MemPool per net keeps what a graph's kernels read and write apart from every later
        allocation."""
        import torch
        from ..nn.gpu import DeviceScalar, LossArena
        if not CP.lazy_losses:
            raise RuntimeError('PageTrainer(graphs=True) needs CP.lazy_losses = True: a loss read with float() '
                               'inside the step is a host sync, which a HIP graph cannot contain')
        rt = CP.runtime()
        captured = {}
        for comp in self.model_system.components:
            model = comp.model
            if comp.name in self.eager_nets:
                continue
            sync, model.grad_sync = model.grad_sync, None         # collectives are not captured
            hook, model.bucket_hook = model.bucket_hook, None     # (replays reduce the whole buffer at once)
            comp.selector(context)
            labels = (comp.selector.X_label, comp.selector.y_label)
            X, y = (self._static_input(label, context[label]) for label in labels)
            pool = torch.cuda.MemPool()
            # data parallel, a net whose gradient tail is final early (parallel.DataParallel.split_node: the Char net's
            # dense block): graph A is cut there, so that the replay can reduce the tail while the rest of A runs
            split = self.dp.split_node(model) if self.dp is not None else None
            parts = []
            try:
                with rt.lane(self.lanes[comp.name]):
                    # the loss slots and their snapshot ring live as long as the graphs: allocated BEFORE the first capture
                    # (what is allocated in the pool between or during captures may be handed out again to a later capture)
                    ring = self.snapshot_losses and model.has_fused_tail()
                    with torch.cuda.use_mem_pool(pool):
                        arena = CP.loss_arena = LossArena(16)
                        if ring:
                            arena.arm()
                    cap = [rt.capture(pool)]
                    cap[0].__enter__()

                    def cut(_model, node, cap=cap, parts=parts, split=split):
                        if node == split:
                            rt.flush_deferred()           # (recorded weight gradients of the tail: into THIS graph)
                            cap[0].__exit__(None, None, None)
                            parts.append(cap[0].graph)
                            cap[0] = rt.capture(pool)
                            cap[0].__enter__()
                    model.bucket_hook = cut if split is not None else None
                    try:
                        model.train_begin(X, y)
                        comp._publish()
                    finally:
                        model.bucket_hook = None
                        cap[0].__exit__(None, None, None)
                    parts.append(cap[0].graph)
                    begin = GraphSequence(parts)
                    pending = model._pending_losses
                    with rt.capture(pool) as finish:
                        if ring:                              # the optimizer tail itself snapshots the loss slots
                            rt.set_loss_snapshot(arena)
                        try:
                            losses = model.train_finish()
                        finally:
                            rt.set_loss_snapshot(None)
            finally:
                CP.loss_arena = None
            slots = pending + [losses['regularization_loss']]
            for value in slots:
                if not isinstance(value, DeviceScalar):
                    raise RuntimeError('graph capture: a loss was materialised on the host')
            captured[comp.name] = dict(begin=begin, finish=finish, arena=arena,
                                       slot_index=[arena.index_of(v.t) for v in slots],
                                       output_losses=[v.t for v in pending],
                                       regularization_loss=losses['regularization_loss'].t,
                                       prediction=context.get(comp.selector.pred_label))
            model.grad_sync = sync
            model.bucket_hook = hook                              # (eager steps of this net keep reducing the tail early)
        self._captured = captured

    def capture(self, context):
        """Run the eager steps still needed for lazy initialisation, then record the graphs now (step()
        would do it on its third call)."""
        assert self.graphs, 'PageTrainer was built without graphs=True'
        while self._captured is None:
            self.step(context)
        return self

    def set_eager_nets(self, names):
        """Run these nets with eager launches from now on although they are captured (and back: an empty tuple) --
        e.g. to bracket one of their kernels with events for a few steps; results are the same."""
        self.join()
        self._eager_override = set(names)

    def static_inputs(self):
        """{context label: the array the captured graphs read}, or None before capture.  Filling these in
        place (after join()) and passing them to step() avoids the copy into the statics."""
        return dict(self._statics) if self._captured is not None else None

    def _static_input(self, label, array):
        """The array the graphs read for `label`: the trainer's own input buffer if the caller uses
        make_context(), otherwise a private clone (a foreign array is never written to)."""
        if not hasattr(self, '_statics'):
            self._statics, self._clones = {}, {}
        if any(array is own for own in self._input_buffers.values()):
            static = array
        else:
            static = self._clones.get(id(array))
            if static is None:
                static = self._clones[id(array)] = array.copy()
        self._statics[label] = static
        return static

    def _step_graphs(self, context):
        from ..nn.gpu import DeviceArray, DeviceScalar
        rt = CP.runtime()
        comps = self._lane_order()
        self.optimizer.refresh_hyper()                   # lr / betas changed since the last step? (device array)
        if self.pipelined and any(context[label] is not static for label, static in self._statics.items()):
            self.join()                                      # the copies below overwrite what the lanes read
        copied = set()
        for label, static in self._statics.items():          # new batch -> the graphs' static inputs
            fresh = context[label]
            if fresh is not static and (id(fresh), id(static)) not in copied:
                rt.call('uocr_d2d', static.ptr, fresh.ptr, static.nbytes)
                copied.add((id(fresh), id(static)))
        start = self._event('start').record()
        captured = {n: e for n, e in self._captured.items() if n not in self._eager_override}
        for comp in comps:
            entry = captured.get(comp.name)
            with rt.lane(self.lanes[comp.name]):
                start.wait()
                if entry is None:                             # eager net
                    comp.selector(context)
                    X, y = next(comp.selector.get())
                    comp.model.train_begin(X, y)
                    comp._publish()
                    continue
                parts = entry['begin'].parts
                parts[0].replay()
                if len(parts) == 2:                           # the gradient tail is final: its all-reduce starts now
                    comp.model.grad_sync.__self__.tail_ready(comp.model)
                    parts[1].replay()
                if comp.model.grad_sync is not None:
                    comp.model.grad_sync(comp.model)          # RCCL all-reduce of the (rest of the) flat gradient
        context['losses'] = {}
        for comp in comps:
            entry, model = captured.get(comp.name), comp.model
            snap = None
            with rt.lane(self.lanes[comp.name]):
                if entry is None:
                    context['losses'][comp.name] = model.train_finish()
                else:
                    if model.grad_sync is not None and model.defer_grad_sync:
                        model.grad_sync.__self__.wait(model)
                    entry['finish'].replay()
                    if self.snapshot_losses:                  # the row of the ring the replayed optimizer tail wrote to,
                        arena = entry['arena']                # or (no fused tail) one 8-byte-per-loss copy on the lane
                        snap = arena.next_row() if arena.ring is not None else arena.snapshot()
                done = self._event(comp.name).record()
            if entry is None:
                self._finish_lane(comp.name, done, context['losses'][comp.name])
                continue
            if entry['prediction'] is not None:
                context[comp.selector.pred_label] = entry['prediction']
            if snap is not None:
                values = [DeviceScalar(snap.t[i]) for i in entry['slot_index']]
            else:                                             # aliases of the static slots: read before the next step()
                values = [DeviceScalar(t) for t in entry['output_losses']] + [DeviceScalar(entry['regularization_loss'])]
            context['losses'][comp.name] = {'output_losses': values[:-1], 'regularization_loss': values[-1]}
            self._finish_lane(comp.name, done, context['losses'][comp.name])
        return context['losses']

    def forward(self, context):
        """Forward pass of every net (reference: ModelSystem.predict -> Model.predict = forward, models.py:270-271).
        With lanes every net runs on its own stream; with graphs=True (and CP.lazy_losses) the nets' forward
        passes are captured once and replayed (inputs = the arrays of the first call's context / the statics).
        Predictions land in context['<net>_pred']."""
        import torch
        if self.lanes is None:
            self.model_system.predict({**context})
            return context
        rt = CP.runtime()
        # (one graph per net, replayed on the net's lane.  ONE multi-stream graph for the whole step -- the lanes
        # joining the capture through events -- was measured and lost: 0.30 ms per step against 0.17, the
        # branches of a replayed graph do not overlap the way free-running streams do)
        if self.graphs and getattr(self, '_fwd_graphs', None) is None:
            self._fwd_graphs, self._fwd_inputs, self._fwd_preds = {}, {}, {}
            for comp in self.model_system.components:                 # warm-up (lazy initialisation), then capture
                comp.model.predict(context[comp.selector.X_label])
            torch.cuda.synchronize()
            for comp in self.model_system.components:
                with rt.lane(self.lanes[comp.name]):
                    X = context[comp.selector.X_label]
                    with rt.capture(torch.cuda.MemPool()) as graph:
                        pred = comp.model.predict(X)[0]
                    self._fwd_graphs[comp.name], self._fwd_inputs[comp.name], self._fwd_preds[comp.name] = graph, X, pred
        start = self._event('fwd_start').record()
        done = {}
        for comp in self._lane_order():
            with rt.lane(self.lanes[comp.name]):
                start.wait()
                X = context[comp.selector.X_label]
                if self.graphs:
                    static = self._fwd_inputs[comp.name]
                    if X is not static:
                        rt.call('uocr_d2d', static.ptr, X.ptr, static.nbytes)
                    self._fwd_graphs[comp.name].replay()
                    context[comp.selector.pred_label] = self._fwd_preds[comp.name]
                else:
                    context[comp.selector.pred_label] = comp.model.predict(X)[0]
                done[comp.name] = self._event('fwd_' + comp.name).record()
        for comp in self.model_system.components:
            done[comp.name].wait()
        return context


class Losses:
    """Per-model running losses of one epoch (reference: my_model/trainer.py:10-125)."""

    def __init__(self, model_names, outputs_cnts):
        self.model_names, self.outputs_cnts = model_names, outputs_cnts
        inf = float('inf')
        self.train_prev_losses = self._new(inf)
        self.val_best_losses = self._new(inf)
        self.val_prev_losses = self._new(inf)
        self.train_losses = self.val_losses = None
        self.best_loss_epoch = {name: 0 for name in model_names}

    def _new(self, value):
        return {name: [value] * self.outputs_cnts[name] for name in self.model_names}

    def reset(self):
        self.train_losses, self.val_losses = self._new(0.0), self._new(0.0)

    def _add(self, table, update):
        for name in self.model_names:
            for i in range(self.outputs_cnts[name]):
                table[name][i] += float(update[name]['output_losses'][i])

    def train(self, update):
        self._add(self.train_losses, update)

    def validation(self, update):
        self._add(self.val_losses, update)

    def normalize(self, n_train, n_val):
        for name in self.model_names:
            self.train_losses[name] = [v / n_train for v in self.train_losses[name]]
            self.val_losses[name] = [v / n_val for v in self.val_losses[name]]

    def get_better_weights(self, epoch):
        better = []
        for name in self.model_names:
            new, best = np.mean(self.val_losses[name]), np.mean(self.val_best_losses[name])
            if new < best or (not np.isnan(new) and np.isnan(best)):
                better.append(name)
                self.val_best_losses[name] = self.val_losses[name]
                self.best_loss_epoch[name] = epoch
        return better

    def next(self):
        self.train_prev_losses, self.val_prev_losses = self.train_losses, self.val_losses

    def print(self, left_margin=0):
        pad = ' ' * left_margin
        for title, now, prev in (('Train loss', self.train_losses, self.train_prev_losses),
                                 ('Validation loss', self.val_losses, self.val_prev_losses)):
            cells = []
            for name in self.model_names:
                vals = ' '.join(f'{v: .6f}' for v in now[name])
                diffs = ' '.join(f'{a - b:+.6f}' for a, b in zip(now[name], prev[name]))
                cells.append(f'{name}: {vals} ({diffs})')
            print(pad + f'{title + ":":18s}' + ' | '.join(cells))


class Trainer:
    """Epoch loop of the reference (my_model/trainer.py:128-296): a validation pre-pass, then per epoch a
    train pass and a validation pass, learning-rate decay, NaN rollback (reload the last weights up to
    10 times with a shrinking lr, then the best ones) and a callback with the models whose
    validation loss improved.  `make_context_func(dataset.get, (index,))` builds the device context."""

    def __init__(self, model_system, make_context_func, models, train_dataset, validation_dataset,
                 progress_tracker=None, show_progress_bar=False, optimizer=None, learning_rate_step=0.995,
                 save_weights_func=None, save_pictures_func=None, data_parallel=None, watchdog=None):
        from ..nn.progress_tracker import BaseProgressTracker
        self.model_system, self.make_context_func, self.models = model_system, make_context_func, models
        self.train_dataset, self.validation_dataset = train_dataset, validation_dataset
        self.progress_tracker = progress_tracker or BaseProgressTracker()
        self.show_progress_bar = show_progress_bar
        self.optimizer, self.learning_rate_step = optimizer, learning_rate_step
        self.save_weights_func, self.save_pictures_func = save_weights_func, save_pictures_func
        # data parallel (parallel.DataParallel over `models`, or None): every rank runs this loop on its own
        # shard of the datasets; the gradient all-reduce happens inside Model.train (grad_sync), so the weights
        # -- and with them the NaN test and the rollback -- are identical on every rank; the epoch losses are
        # averaged over the ranks so that every rank also takes the same "better weights" decisions, and only
        # rank 0 writes model_weights.json
        self.dp = data_parallel
        self.watchdog = watchdog                     # watchdog.Watchdog or None: beaten after every iteration

    def _rank_mean(self, losses):
        """Average the normalised epoch losses over the ranks (host side, a few floats per epoch)."""
        if self.dp is None or self.dp.world == 1:
            return
        import torch
        for table in (losses.train_losses, losses.val_losses):
            names = sorted(table)
            flat = torch.tensor([v for n in names for v in table[n]], dtype=torch.float64)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=getattr(self.dp, 'group', None))
            flat /= self.dp.world
            it = iter(flat.tolist())
            for n in names:
                table[n] = [next(it) for _ in table[n]]

    def _weights(self):
        return {name: w for model in self.models.values() for name, w in model.get_weights().items()}

    def _pass(self, dataset, order, losses_sink, train, epoch, stage):
        for i, idx in enumerate(order):
            self.progress_tracker.reset()
            context = self.make_context_func(dataset.get, (idx,))
            (self.model_system.train if train else self.model_system.test)(context)
            losses_sink(context['losses'])
            if self.save_pictures_func is not None:
                self.save_pictures_func(epoch, stage, i, context)
            self.progress_tracker.message(f'{stage}_iteration', {'current': i + 1, 'total': len(order)})
            if self.watchdog is not None:
                self.watchdog.beat(f'epoch {epoch} {stage} iteration {i + 1}')

    def train(self, num_epochs):
        from random import shuffle
        names = list(self.models)
        losses = Losses(names, {n: m.get_outputs_count() for n, m in self.models.items()})
        n_train, n_val = len(self.train_dataset), len(self.validation_dataset)
        assert n_val > 0, 'Validation dataset must have at least 1 element'
        print('Precomputing losses')
        losses.reset()
        self._pass(self.validation_dataset, list(range(n_val)), losses.validation, False, 0, 'precomputing')
        losses.next()
        best_weights = last_weights = self._weights()
        reload_attempts, epoch = 0, 1
        train_order, val_order = list(range(n_train)), list(range(n_val))
        while epoch <= num_epochs:
            print(f'Epoch {epoch}/{num_epochs}:' + (f'  lr = {self.optimizer.lr}' if self.optimizer else ''))
            self.progress_tracker.message('epoch', {'current': epoch, 'total': num_epochs})
            losses.reset()
            shuffle(train_order)
            self._pass(self.train_dataset, train_order, losses.train, True, epoch, 'train')
            shuffle(val_order)
            self._pass(self.validation_dataset, val_order, losses.validation, False, epoch, 'validation')
            losses.normalize(n_train, n_val)
            self._rank_mean(losses)
            has_nan = any(model.nan_weights() for model in self.models.values())
            if self.optimizer is not None:
                reload_attempts += 1
                self.optimizer.lr *= self.learning_rate_step ** reload_attempts
                if has_nan:
                    source = last_weights if reload_attempts < 10 else best_weights
                    print('NaN value found in weights, loading ' +
                          ('last weights' if reload_attempts < 10 else 'last best weights'))
                    for model in self.models.values():
                        model.set_weights(source)
                    if reload_attempts >= 10:
                        reload_attempts = 0
                    continue
            elif has_nan:
                raise ValueError('NaN value found in weights, but no optimizer provided. Provide optimizer and '
                                 'learning_rate_step, so learning rate could be decreased to try avoiding NaN values')
            losses.print(left_margin=2)
            better = losses.get_better_weights(epoch)
            if better and self.save_weights_func and (self.dp is None or self.dp.rank == 0):
                print('  Saving weights for ' + ', '.join(better))
                self.save_weights_func(better)
            # (`best_weights` stays what it was before the first epoch, exactly as in the reference, which
            # never reassigns it: my_model/trainer.py:184, 264-268)
            last_weights = self._weights()
            epoch += 1
            reload_attempts = 0
            losses.next()
        return losses.val_best_losses, losses.best_loss_epoch
