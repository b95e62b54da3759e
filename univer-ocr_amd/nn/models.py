"""Graph executor (reference: nn/models.py:7-502): `Model(layers, relations, loss)` and
`Sequential(list)` with forward / backward / compute_loss_and_gradients / train / test / predict /
params / get_weights / set_weights, nested models flattened to 'parent/child' layer names.

The reference walks the DAG with memoised recursion on every call.  This executor compiles the DAG
once (`initialize`) into a topologically ordered plan, so forward is a flat loop of asynchronous
kernel launches on one HIP stream; parameters of the whole model are packed into one flat buffer
(layers.ParamPack), which turns `clear_grads`, `regularize`, `update_grads`, `nan_weights` and the
data-parallel gradient all-reduce into one launch each.

Semantics kept from the reference:
  * gradients are cleared when the forward pass starts (models.py:188) and again after the
    optimizer step in `train` (:250-254);
  * gradients of a layer consumed by several layers are summed (:218);
  * `layers_outputs[k]` / `input_grads[k]` hold the model outputs / input gradients of the last call;
  * `compute_loss_and_gradients` returns {'output_losses': [...], 'regularization_loss': r}.
"""
import numpy as np

from . import ops
from .gpu import CP, DeviceScalar
from .help_func import make_list_if_not
from .layers import BaseLayer, ParamPack
from .losses import SoftmaxCrossEntropy
from .progress_tracker import track_method


class BaseModel(BaseLayer):
    def compute_loss_and_gradients(self, X, y):
        raise NotImplementedError()

    def train(self, X, y):
        raise NotImplementedError()

    def test(self, X, y):
        raise NotImplementedError()

    def predict(self, X):
        raise NotImplementedError()


def _expand(layers, relations, prefix=''):
    """Flatten nested models (models.py:109-158).  Returns (leaf_layers, relations) where every
    relation value is a list of sources; a source is an int (model input) or a leaf layer name."""
    relations = {dst: list(make_list_if_not(srcs)) for dst, srcs in relations.items()}
    leaves = {}
    outputs_of = {}              # sub-model name -> {out_id: [sources]} in the parent's namespace
    for name, layer in layers.items():
        if not isinstance(layer, Model):
            leaves[name] = layer
            continue
        sub_leaves, sub_rel = layer.layers, layer.relations
        parent_inputs = relations[name]

        def translate(src, _name=name, _inputs=parent_inputs):
            return _inputs[src] if isinstance(src, int) else f'{_name}/{src}'
        outputs_of[name] = {}
        for dst, srcs in sub_rel.items():
            mapped = [translate(s) for s in srcs]
            if isinstance(dst, int):
                outputs_of[name][dst] = mapped
            else:
                relations[f'{name}/{dst}'] = mapped
        for sub_name, sub_layer in sub_leaves.items():
            leaves[f'{name}/{sub_name}'] = sub_layer
        del relations[name]

    def resolve(src):
        """A reference to a sub-model means its output(s); they may in turn point at other sub-models."""
        if isinstance(src, tuple) and len(src) > 1 and src[0] in outputs_of:
            found = []
            for out_id in src[1:]:
                for s in outputs_of[src[0]][out_id]:
                    found.extend(resolve(s))
            return found
        if isinstance(src, str) and src in outputs_of:
            found = []
            for out_id in sorted(outputs_of[src]):
                for s in outputs_of[src][out_id]:
                    found.extend(resolve(s))
            return found
        return [src]

    flat = {}
    for dst, srcs in relations.items():
        out = []
        for s in srcs:
            out.extend(resolve(s))
        flat[dst] = out
    return leaves, flat


class Model(BaseModel):
    def __init__(self, layers, relations, loss=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if not isinstance(layers, dict):
            raise TypeError(f'layers argument must be dict, found: {type(layers).__name__}')
        if not isinstance(relations, dict):
            raise TypeError(f'relations argument must be dict, found: {type(relations).__name__}')
        self.ravelled_layers = layers
        self.ravelled_relations = relations
        self.outputs_count = max(k for k in relations if isinstance(k, int)) + 1
        self.loss = SoftmaxCrossEntropy() if loss is None else loss
        self.layers_outputs = {}
        self.input_grads = {}
        self._skipped_input_grads = False
        self.relations_backward = {}
        self.is_initialized = False
        self._plan = None
        self._pack = None
        self.grad_sync = None            # set by parallel.DataParallel
        self.defer_grad_sync = False     # True: the all-reduce is waited for in train_finish()
        self._pending_losses = None
        self.bucket_hook = None
        self.side_wgrad = False          # weight-gradient kernels on the lane's side stream (Runtime.side)
        self.group_wgrad = False         # small weight-gradient GEMMs of a backward pass as one launch (Runtime.defer_wgrad)
        self._receptive_fields = {}
        self.layers, self.relations = None, None
        self.unravel_model()

    # -- structure ---------------------------------------------------------------------------------
    def unravel_model(self):
        if self.layers is None:
            self.layers, self.relations = _expand(self.ravelled_layers, self.ravelled_relations)
            all_ints = [s for srcs in self.relations.values() for s in srcs if isinstance(s, int)]
            self.inputs_count = max(all_ints) + 1 if all_ints else 0
        for name, layer in self.layers.items():
            layer._set_name(name)

    def get_leaf_layers(self):
        return self.layers

    def __getitem__(self, key):
        return self.layers[key]

    def _toposort(self):
        order, state = [], {}

        def visit(node):
            if state.get(node) == 2:
                return
            if state.get(node) == 1:
                raise RecursionError(f'Looped on {node} layer, check relations')
            state[node] = 1
            for src in self.relations[node]:
                if not isinstance(src, int):
                    visit(src)
            state[node] = 2
            if not isinstance(node, int):
                order.append(node)
        for out in range(self.outputs_count):
            visit(out)
        return order

    def initialize(self, input_shapes):
        """models.py:55-107: propagate shapes, initialise lazily-shaped layers, record consumers."""
        input_shapes = make_list_if_not(input_shapes)
        self.input_shapes = input_shapes
        order = self._toposort()
        shapes = {}
        self.relations_backward = {}
        for node in order + list(range(self.outputs_count)):
            in_shapes = []
            for i, src in enumerate(self.relations[node]):
                in_shapes.append(input_shapes[src] if isinstance(src, int) else shapes[src])
                self.relations_backward.setdefault(src, {})[node] = i
            if isinstance(node, int):
                continue
            layer = self.layers[node]
            if not layer.is_initialized:
                layer.initialize(in_shapes)
            out = layer.get_output_shapes(in_shapes)
            shapes[node] = out[0] if isinstance(out, list) else out
        never = [n for n in self.layers if n not in shapes]
        if never:
            print(f'These layers have never been visited: {never}')
        self._plan = order
        self._build_pack()
        self.is_initialized = True

    def _build_pack(self):
        params = [p for _, p in sorted(self.params().items())]
        dtypes = {p.value.dtype for p in params}
        self._pack = ParamPack(params) if params and len(dtypes) == 1 else None
        self._reg_ranges = None

    @property
    def pack(self):
        return self._pack

    # -- forward / backward ----------------------------------------------------------------------------
    @track_method('forward')
    def forward(self, inputs):
        inputs = [ops.as_device(x) for x in make_list_if_not(inputs)]
        if not self.is_initialized:
            self.initialize_from_X(inputs)
        self.clear_param_grads()                      # models.py:188 (per layer there, once here)
        outputs = {}
        fused_conv, fused_act = self._fusion_maps()
        pairs = self._active_pairs(inputs)
        pair_first = {first: second for second, (first, _, _) in pairs.items()}
        ups = self._active_ups()
        up_nodes = {up for up, _ in ups.values()}
        wins = self._active_windows()
        win_nodes = {n for fw, flat, _ in wins.values() for n in (fw, flat)}
        for node in self._plan:
            if node in pair_first or node in up_nodes or node in win_nodes:   # computed inside the kernel of its consumer
                outputs[node] = None
                continue
            if node in wins:                          # windows + flatten + dense on the conv feature map
                fw = wins[node][0]
                src = self.relations[fw][0]
                x = inputs[src] if isinstance(src, int) else outputs[src]
                if x.shape[3] % 32 == 0:
                    act = self.layers[fused_conv[node]] if node in fused_conv else None
                    outputs[node] = self.layers[node].forward_windows(x, self.layers[fw].width, act)
                    continue
                for part in (fw, wins[node][1]):      # channel count the implicit GEMM does not take
                    out = self.layers[part].forward([x])
                    x = outputs[part] = out[0] if isinstance(out, list) else out
                wins = {k: v for k, v in wins.items() if k != node}
            if node in ups:                           # upsample + conv on the low-res tensor
                src = self.relations[ups[node][0]][0]
                x_low = inputs[src] if isinstance(src, int) else outputs[src]
                act = self.layers[fused_conv[node]] if node in fused_conv else None
                outputs[node] = self.layers[node].forward_up(x_low, act)
                continue
            if node in fused_act:                     # activation absorbed into its producing conv
                outputs[node] = outputs[fused_act[node]]
                continue
            if node in pairs:
                first, act_a, act_b = pairs[node]
                src = self.relations[first][0]
                x = inputs[src] if isinstance(src, int) else outputs[src]
                outputs[node] = self.layers[node].forward_pair(
                    x, self.layers[first], self.layers[act_a], None if act_b is None else self.layers[act_b])
                continue
            args = [inputs[s] if isinstance(s, int) else outputs[s] for s in self.relations[node]]
            if node in fused_conv:
                out = self.layers[node].forward_fused(args, self.layers[fused_conv[node]])
            else:
                out = self.layers[node].forward(args)
            outputs[node] = out[0] if isinstance(out, list) else out
        for k in range(self.outputs_count):
            src = self.relations[k][0]
            outputs[k] = inputs[src] if isinstance(src, int) else outputs[src]
        self.layers_outputs = outputs
        self._pairs_used = pairs
        self._ups_used = ups
        self._wins_used = wins
        return [outputs[k] for k in range(self.outputs_count)]

    def _active_pairs(self, inputs):
        """The pair kernels exist in float32 only; other dtypes run the layers one by one."""
        pairs = getattr(self, '_pairs', {})
        if pairs and all(self.layers[n].w.value.dtype == np.float32 for n in pairs):
            return pairs
        return {}

    @track_method('backward')
    def backward(self, grads):
        if self.group_wgrad and CP.has_device():
            with CP.runtime().defer_wgrad():
                return self._backward_pass(grads)
        rt = CP.runtime() if self.side_wgrad and CP.has_device() else None
        if rt is None:
            return self._backward_pass(grads)
        # weight gradients go to the side stream of this lane: the chain of dX kernels does not wait for them;
        # they are back before anything reads a parameter gradient (here, and in front of a bucket hook)
        rt.side_on = True
        try:
            return self._backward_pass(grads)
        finally:
            rt.side_on = False
            rt.join_side()

    def _backward_pass(self, grads):
        grads = [ops.as_device(g) for g in make_list_if_not(grads)]
        grads_mem = {}

        def incoming(node):
            parts = []
            for dst, i in self.relations_backward.get(node, {}).items():
                parts.append(grads[dst] if isinstance(dst, int) else grads_mem[dst][i])
            total = parts[0]
            for extra in parts[1:]:
                total = ops.add(total, extra)         # models.py:218
            return total

        fused_conv, fused_act = self._fusion_maps()
        pairs = getattr(self, '_pairs_used', {})
        pair_first = {first: second for second, (first, _, _) in pairs.items()}
        ups = getattr(self, '_ups_used', {})
        up_nodes = {up for up, _ in ups.values()}
        wins = getattr(self, '_wins_used', {})
        win_nodes = {n for fw, flat, _ in wins.values() for n in (fw, flat)}
        for node in reversed(self._plan):
            if node not in self.relations_backward:
                continue
            if node in wins:                          # dW and dX w.r.t. the conv feature map in one go
                fw, flat, act_in = wins[node]
                act = self.layers[fused_conv[node]] if node in fused_conv else None
                dx = self.layers[node].backward_windows(
                    incoming(node), self.layers[fw].width, act,
                    act_grad_applied=act is not None and self._act_folded(fused_conv[node]),
                    input_activation=None if act_in is None else self.layers[act_in])
                grads_mem[node] = [None]
                grads_mem[flat] = [None]
                grads_mem[fw] = [dx]
            elif node in win_nodes:
                grads_mem.setdefault(node, [None])
            elif node in ups:                           # dW, and dX w.r.t. the LOW-RES input, in one go
                up_node, act_in = ups[node]
                act = self.layers[fused_conv[node]] if node in fused_conv else None
                dx_low = self.layers[node].backward_up(
                    incoming(node), act, act_grad_applied=act is not None and self._act_folded(fused_conv[node]),
                    input_activation=None if act_in is None else self.layers[act_in])
                grads_mem[node] = [None]
                grads_mem[up_node] = [dx_low]
            elif node in up_nodes:
                grads_mem.setdefault(node, [None])
            elif node in pairs:                         # dW of both convs and dX of the first in one kernel
                first, act_a, act_b = pairs[node]
                folded = act_b is None or act_b in getattr(self, '_loss_folded', ())    # Sigmoid' already in the loss gradient
                dx = self.layers[node].backward_pair(
                    incoming(node), self.layers[first], self.layers[act_a], None if folded else self.layers[act_b])
                grads_mem[node] = [None]
                grads_mem[first] = [dx]
            elif node in pair_first or (node in fused_act and fused_act[node] in pair_first):
                grads_mem.setdefault(node, [None])
            elif node in fused_act:                   # its gradient is applied inside a conv's backward
                grads_mem[node] = [incoming(node)]
            elif node in fused_conv or node in self._fusion[2]:
                act = self.layers[fused_conv[node]] if node in fused_conv else None
                folded = node in fused_conv and self._act_folded(fused_conv[node])
                in_act = self._fusion[2].get(node)
                grads_mem[node] = make_list_if_not(self.layers[node].backward_fused(
                    incoming(node), act, act_grad_applied=folded,
                    input_activation=None if in_act is None else self.layers[in_act]))
            else:
                grads_mem[node] = make_list_if_not(self.layers[node].backward(incoming(node)))
            if self.bucket_hook is not None:          # data parallel: part of the gradient may be final now
                self.bucket_hook(self, node)
        if self._skipped_input_grads:
            self.input_grads = {}
            return []
        self.input_grads = {key: incoming(key) for key in range(self.inputs_count)
                            if key in self.relations_backward}
        return [self.input_grads[k] for k in range(self.inputs_count)]

    def skip_input_grads(self, on=True):
        """Training never reads the gradient w.r.t. the model's inputs (the reference computes it in
        every backward, models.py:226-230, and only gradient_check.py:152 looks at it).  With this on,
        a Convolutional2D fed directly and only by model inputs skips its dX kernel and
        `input_grads` stays empty; parameter gradients, losses and updates are unchanged."""
        from .layers import Convolutional2D
        firsts = [n for n in self._plan if all(isinstance(s, int) for s in self.relations[n])
                  and isinstance(self.layers[n], Convolutional2D)]
        for n in firsts:
            self.layers[n].needs_input_grad = not on
        self._skipped_input_grads = bool(on)
        return self

    # -- conv + activation fusion (graph level; reference: none -- every layer is its own pass) -----
    def enable_fusion(self, on=True, pairs=True, windows=True):
        """Run every Convolutional2D whose ONLY consumer is a LeakyRelu(alpha > 0) / Sigmoid as one
        kernel with the activation in the epilogue.  Results are the same tensors the unfused graph
        produces for the activation layers; the conv's pre-activation output is not materialised
        (layers_outputs[conv] then aliases the activation output)."""
        self.fuse_activations = bool(on)
        self.fuse_pairs = bool(pairs)        # also run conv(1->16)+LeakyReLU+conv(16->1) blocks as one kernel
        self.fuse_windows = bool(windows)    # and windows + flatten + dense as one implicit GEMM
        self._fusion = None
        return self

    def _fusion_maps(self):
        if not getattr(self, 'fuse_activations', False):
            self._fusion = ({}, {}, {}, set())
            self._pairs = {}
            self._ups = {}
            self._wins = {}
            return {}, {}
        if self._fusion is None:
            from .layers import Convolutional2D, FullyConnected, LeakyRelu, Sigmoid
            fused_conv, fused_act = {}, {}           # ("conv": Convolutional2D or FullyConnected)
            for node in self._plan:
                layer = self.layers[node]
                consumers = self.relations_backward.get(node, {})
                if not isinstance(layer, (Convolutional2D, FullyConnected)) or len(consumers) != 1:
                    continue
                (dst, _), = consumers.items()
                if isinstance(dst, int) or self.relations[dst] != [node]:
                    continue
                act = self.layers[dst]
                ok = isinstance(act, Sigmoid) or (isinstance(act, LeakyRelu) and act.alpha > 0)
                if ok and type(act) in (Sigmoid, LeakyRelu):
                    fused_conv[node] = dst
                    fused_act[dst] = node
            # a fused activation whose ONLY consumer is a conv: that conv's dx kernel multiplies by the
            # activation's derivative in its epilogue (input_of[conv] = act), and the producing conv
            # skips its own activation-gradient pass (folded)
            input_of, folded = {}, set()
            for act_node in fused_act:
                consumers = self.relations_backward.get(act_node, {})
                if len(consumers) != 1:
                    continue
                (dst, _), = consumers.items()
                if not isinstance(dst, int) and isinstance(self.layers[dst], (Convolutional2D, FullyConnected)) and \
                        self.relations[dst] == [act_node]:
                    input_of[dst] = act_node
                    folded.add(act_node)
            self._fusion = (fused_conv, fused_act, input_of, folded)
            self._pairs = self._find_pairs(fused_conv, input_of) if getattr(self, 'fuse_pairs', True) else {}
            self._ups = self._find_ups(fused_act) if getattr(self, 'fuse_pairs', True) else {}
            self._wins = self._find_windows(fused_act) if getattr(self, 'fuse_windows', True) else {}
        return self._fusion[0], self._fusion[1]

    def _find_windows(self, fused_act):
        """Conv2DToBatchedFixedWidthed feeding only a Flatten feeding only a FullyConnected -- the bridge
        between the conv block and the dense block of the Char net (my_model/model.py:250-304) -- runs as one
        implicit GEMM on the conv feature map (ops.windows_dense_fwd): the 8x larger windows tensor and its
        gradient are never built.  Returns {dense: (windows node, flatten node, fused activation that feeds only
        the windows layer, or None)}; that activation's backward is folded into the dx epilogue."""
        from .layers import Conv2DToBatchedFixedWidthed, Flatten, FullyConnected

        def only_consumer(node, kind):
            consumers = self.relations_backward.get(node, {})
            if len(consumers) != 1:
                return None
            (dst, _), = consumers.items()
            if isinstance(dst, int) or self.relations[dst] != [node] or not isinstance(self.layers[dst], kind):
                return None
            return dst

        wins = {}
        for node in self._plan:
            if not isinstance(self.layers[node], Conv2DToBatchedFixedWidthed):
                continue
            flat = only_consumer(node, Flatten)
            dense = only_consumer(flat, FullyConnected) if flat is not None else None
            if dense is None:
                continue
            src = self.relations[node]
            act_in = None
            if len(src) == 1 and src[0] in fused_act and len(self.relations_backward.get(src[0], {})) == 1:
                act_in = src[0]
            wins[dense] = (node, flat, act_in)
        return wins

    def _active_windows(self):
        """float32 with channel counts the MFMA implicit GEMM takes (the generic conv kernels would be slower
        than the three separate layers)."""
        wins = getattr(self, '_wins', {})
        out = {}
        for dense, v in wins.items():
            layer = self.layers[dense]
            if layer.is_initialized and layer.w.value.dtype == np.float32 and layer.n_output % 32 == 0 and \
                    layer.n_input % (32 * self.layers[v[0]].width) == 0:
                out[dense] = v
        return out

    def _find_ups(self, fused_act):
        """Upsample2D(2) feeding only a 5x5 / stride 1 / padding 2 Convolutional2D with 4->4 or 1->1 channels -- the
        decoder blocks of the Line and Paragraph nets (my_model/model.py:138-247) -- runs as one op on the low-res tensor (csrc/conv_up.hip);
        the upsampled tensor is never built.  Returns {conv: (upsample node, fused activation that feeds only
        this upsample, or None)}; that activation's backward is folded into the op's dx epilogue."""
        from .layers import Convolutional2D, Upsample2D
        ups = {}
        for node in self._plan:
            layer = self.layers[node]
            consumers = self.relations_backward.get(node, {})
            if not isinstance(layer, Upsample2D) or tuple(layer.scale_factor) != (2, 2) or len(consumers) != 1:
                continue
            (dst, _), = consumers.items()
            if isinstance(dst, int) or self.relations[dst] != [node]:
                continue
            conv = self.layers[dst]
            if not (isinstance(conv, Convolutional2D) and conv.kernel_size == (5, 5) and conv.stride == (1, 1)
                    and conv.padding == (2, 2) and conv.padding_value == 0
                    and (conv.in_channels, conv.out_channels) in ((4, 4), (1, 1))):
                continue
            src = self.relations[node]
            act_in = None
            if len(src) == 1 and src[0] in fused_act and len(self.relations_backward.get(src[0], {})) == 1:
                act_in = src[0]
            ups[dst] = (node, act_in)
        return ups

    def _active_ups(self):
        """float32 only, like the pair kernels."""
        ups = getattr(self, '_ups', {})
        return {conv: v for conv, v in ups.items() if self.layers[conv].w.value.dtype == np.float32}

    def _act_folded(self, act_node):
        """Is the backward of this fused activation applied by its consumer (a conv's dx kernel, or the
        loss kernel for an output Sigmoid)?"""
        return act_node in self._fusion[3] or act_node in getattr(self, '_loss_folded', ()) or \
            any(a == act_node for _, a in getattr(self, '_ups_used', {}).values()) or \
            any(a == act_node for _, _, a in getattr(self, '_wins_used', {}).values())

    def _foldable_output_sigmoid(self, key):
        """Model output `key` = a Sigmoid fused into its conv (or into a pair kernel) and consumed by nothing else: its
        backward can move into the loss-gradient kernel (an elementwise, HBM-bound kernel that reads the prediction
        anyway; the pair backward is bound by vector issue and saves the loads of y and three instructions per
        position: 181 -> 167 us at 8 x 1024 x 2048 in float16)."""
        from .layers import Sigmoid
        node = self.relations[key][0]
        fused_act = self._fusion[1] if self._fusion else {}
        if isinstance(node, int) or node not in fused_act or not isinstance(self.layers[node], Sigmoid):
            return None
        if len(self.relations_backward.get(node, {})) != 1:
            return None
        return node

    def _find_pairs(self, fused_conv, input_of):
        """conv3x3(1->16, pad 1) + LeakyReLU feeding only conv3x3(16->1, pad 1) [+ Sigmoid] -- the
        Monochrome block (my_model/model.py:108-135) -- runs as ONE forward and ONE backward kernel
        (csrc/conv_pair.hip) that never writes the 16-channel activation or its gradient to HBM.
        Returns {second conv: (first conv, its LeakyReLU, the second conv's fused activation or None)}."""
        from .layers import LeakyRelu, Sigmoid
        pairs = {}
        from .layers import Convolutional2D
        for conv_b, act_a in input_of.items():
            conv_a = self._fusion[1][act_a]
            a, b, act = self.layers[conv_a], self.layers[conv_b], self.layers[act_a]
            if not isinstance(act, LeakyRelu) or not (isinstance(a, Convolutional2D) and isinstance(b, Convolutional2D)):
                continue
            if not 0.0 <= act.alpha <= 1.0:               # the fused kernels take LeakyReLU as max(z, alpha z)
                continue
            same = all(l.kernel_size == (3, 3) and l.stride == (1, 1) and l.padding == (1, 1) for l in (a, b))
            if not (same and (a.in_channels, a.out_channels, b.in_channels, b.out_channels) == (1, 16, 16, 1)
                    and b.padding_value == 0):
                continue
            act_b = fused_conv.get(conv_b)
            if act_b is not None and not isinstance(self.layers[act_b], Sigmoid):
                continue
            pairs[conv_b] = (conv_a, act_a, act_b)
        return pairs

    def _loss_func(self, key):
        return self.loss[key] if isinstance(self.loss, list) else self.loss

    def _forward_loss_backward(self, X, y):
        predicted = self.forward(make_list_if_not(X))
        y = make_list_if_not(y)
        losses, gradients = [], []
        self._loss_folded = set()
        for key in range(self.outputs_count):
            func, act_node = self._loss_func(key), self._foldable_output_sigmoid(key)
            if act_node is not None and getattr(func, 'folds_sigmoid', False):
                # the loss kernel writes the gradient w.r.t. the INPUT of the fused output Sigmoid
                loss, grad = func(predicted[key], ops.as_device(y[key]), out_act='sigmoid')
                self._loss_folded.add(act_node)
            else:
                loss, grad = func(predicted[key], ops.as_device(y[key]))
            losses.append(loss)
            gradients.append(grad)
        self.backward(gradients)
        self._loss_folded = set()
        if self.grad_sync is not None:
            self.grad_sync(self)                      # data parallel: RCCL all-reduce of pack.grad
        return losses

    def compute_loss_and_gradients(self, X, y):
        losses = self._forward_loss_backward(X, y)
        if self.grad_sync is not None and self.defer_grad_sync:
            self.grad_sync.__self__.wait(self)
        return {'output_losses': losses, 'regularization_loss': self.regularize()}

    def train(self, X, y):
        """models.py:250-254: compute_loss_and_gradients, update_grads, clear_grads."""
        self.train_begin(X, y)
        return self.train_finish()

    # two-phase train step: lets a data-parallel driver overlap this model's gradient all-reduce
    # with the next model's forward/backward (parallel.DataParallel, my_model/trainer.py)
    def train_begin(self, X, y):
        self._pending_losses = self._forward_loss_backward(X, y)

    def train_finish(self):
        if self.grad_sync is not None and self.defer_grad_sync:
            self.grad_sync.__self__.wait(self)
        fused = self._fused_tail()
        if fused is not None:                         # regularize + update + clear_grads in one pass
            losses = {'output_losses': self._pending_losses, 'regularization_loss': fused}
            self._pending_losses = None
            self.input_grads = {}
            return losses
        losses = {'output_losses': self._pending_losses, 'regularization_loss': self.regularize()}
        self._pending_losses = None
        self.update_grads()
        self.clear_grads()
        return losses

    def _fused_tail(self):
        """The three calls that end a train step (models.py:252-254: regularize via compute_loss_and_gradients,
        update_grads, clear_grads) as ONE kernel over the flat pack when the whole model is trained by one
        Momentum or Adam optimizer and has at most 4 L1/L2 ranges; None = not applicable, run them one by one."""
        plan = self._fused_tail_plan()
        if plan is None:
            return None
        optimizer, pack, ranges = plan
        return optimizer.update_pack_fused(pack, ranges)

    def _fused_tail_plan(self):
        from .optimizers import Adam, Momentum
        pack = self._pack
        if pack is None or not self.trainable or not all(layer.trainable for layer in self.layers.values()):
            return None
        optimizer = pack.same_optimizer()
        if type(optimizer) not in (Momentum, Adam):
            return None
        ranges = self._regularizer_ranges()
        if len(ranges) > 4 or any(key[0] not in ('l1', 'l2') for key, _, _ in ranges):
            return None
        return optimizer, pack, ranges

    def has_fused_tail(self):
        """True when train_finish ends in ONE fused kernel (which can also snapshot the loss slots: LossArena)."""
        return self._fused_tail_plan() is not None

    def test(self, X, y):
        predicted = self.forward(make_list_if_not(X))
        y = make_list_if_not(y)
        return {'output_losses': [self._loss_func(k).value_only(predicted[k], ops.as_device(y[k]))
                                  for k in range(self.outputs_count)]}

    def predict(self, X):
        return self.forward(X)

    # -- parameters --------------------------------------------------------------------------------------
    def params(self):
        return {f'{lname}/{pname}': param
                for lname, layer in self.layers.items() for pname, param in layer.params().items()}

    def clear_param_grads(self):
        if self._pack is not None:
            self._pack.zero_grad()
        else:
            for layer in self.layers.values():
                layer.clear_grads()

    def clear_grads(self):
        self.clear_param_grads()
        self.input_grads = {}

    def update_grads(self):
        if not self.trainable:
            return
        pack = self._pack
        if pack is not None and all(layer.trainable for layer in self.layers.values()):
            optimizer = pack.same_optimizer()
            if optimizer is not None:
                optimizer.update_pack(pack)           # one fused launch for the whole model
                return
        for layer in self.layers.values():
            layer.update_grads()

    def regularize(self):
        """models.py:472-476.  All regularised parameters add their loss into one device slot;
        one host sync at the end (none with CP.lazy_losses)."""
        layers = [layer for layer in self.layers.values() if layer.regularizer is not None and layer.params()]
        if not layers:
            return 0
        if self._pack is not None:
            # neighbouring parameters with the same regulariser are one range of the flat pack
            # (alignment gaps hold zeros and contribute nothing): one launch per range; the first
            # range overwrites the loss slot, so the slot needs no zero-fill launch
            slot = CP.loss_slot()
            for i, ((kind, strength), lo, hi) in enumerate(self._regularizer_ranges()):
                ops.regularize(kind, self._pack.view_of(self._pack.value, lo, hi - lo, (hi - lo,)),
                               self._pack.view_of(self._pack.grad, lo, hi - lo, (hi - lo,)), strength, slot, i > 0)
        else:
            slot = CP.zeros((1,), np.float64)
            for layer in layers:
                layer.regularize(slot)
        return DeviceScalar(slot.t) if CP.lazy_losses else float(slot.t.item())

    def _regularizer_ranges(self):
        if self._reg_ranges is None:
            owner = {id(p): layer for layer in self.layers.values() for p in layer.params().values()}
            ranges = []
            entries = self._pack.entries
            for idx, (p, off, size) in enumerate(entries):
                end = entries[idx + 1][1] if idx + 1 < len(entries) else self._pack.total
                reg = owner[id(p)].regularizer
                if reg is None:
                    continue
                key = (reg.kind, reg.reg_strength)
                if ranges and ranges[-1][0] == key and ranges[-1][2] == off:
                    ranges[-1][2] = end
                else:
                    ranges.append([key, off, end])
            self._reg_ranges = ranges
        return self._reg_ranges

    def get_weights(self):
        """models.py:455-459 / layers.py:120-121: {layer: {param: nested lists}}.  With a ParamPack the whole
        model comes down in ONE device-to-host copy of the flat buffer (the reference does one `tolist()`
        = one transfer per parameter)."""
        if self._pack is not None:
            host = self._pack.value.numpy()
            where = {id(p): (off, size) for p, off, size in self._pack.entries}
            weights = {}
            for name, layer in self.layers.items():
                params = layer.params()
                if params and all(id(p) in where for p in params.values()):
                    weights[name] = {pname: host[where[id(p)][0]:where[id(p)][0] + where[id(p)][1]]
                                     .reshape(p.value.shape).tolist() for pname, p in params.items()}
                elif params:
                    weights[name] = layer.get_weights()
            return weights
        weights = {name: layer.get_weights() for name, layer in self.layers.items()}
        return {name: w for name, w in weights.items() if w != {}}

    def set_weights(self, weights):
        for name, layer in self.layers.items():
            layer_weights = weights.get(name)
            if layer_weights is not None:
                layer.set_weights(layer_weights)

    def nan_weights(self):
        if self._pack is not None:
            return ops.has_nan(self._pack.value)
        return any(layer.nan_weights() for layer in self.layers.values())

    def count_parameters(self):
        return sum(layer.count_parameters() for layer in self.layers.values())

    # -- shapes --------------------------------------------------------------------------------------------
    def get_all_output_shapes(self, input_shapes):
        input_shapes = make_list_if_not(input_shapes)

        def clean(shapes):
            return [tuple(int(v) for v in s) for s in make_list_if_not(shapes)]
        per_layer, extra = {}, {}
        for node in self._plan if self._plan is not None else self._toposort():
            ins = []
            for src in self.relations[node]:
                s = input_shapes[src] if isinstance(src, int) else per_layer[src]
                ins.append(s[0] if isinstance(s, list) else s)
            out, more = self.layers[node].get_all_output_shapes(ins)
            per_layer[node] = clean(out)
            extra.update({f'{node}/{k}': clean(v) for k, v in more.items()})
        result = []
        for k in range(self.outputs_count):
            src = self.relations[k][0]
            result.append(input_shapes[src] if isinstance(src, int) else per_layer[src][0])
        extra.update(per_layer)
        return clean(result), extra

    def get_output_shapes(self, input_shapes):
        return self.get_all_output_shapes(input_shapes)[0]

    def get_outputs_count(self):
        return self.outputs_count

    def is_fully_convolutional(self):
        return all(layer.is_fully_convolutional() for layer in self.layers.values())

    def changes_receptive_field(self):
        return any(layer.changes_receptive_field() for layer in self.layers.values())

    # -- receptive fields (models.py:340-432) -------------------------------------------------------------------
    def _field_of(self, node, axis, position, memo):
        """{input_key: set of input positions} that position `position` of node's output sees."""
        key = (node, axis, position)
        if key in memo:
            return memo[key]
        points = {0: {position}} if isinstance(node, int) else \
            self.layers[node]._get_receptive_field(axis, position, 0)
        result = {k: set() for k in range(self.inputs_count)}
        for src_id, src in enumerate(self.relations[node]):
            pts = points.get(src_id, points.get(0, set())) if not isinstance(node, int) else points[0]
            if isinstance(src, int):
                result[src].update(pts)
                continue
            for p in pts:
                for in_key, found in self._field_of(src, axis, p, memo).items():
                    result[in_key].update(found)
        memo[key] = result
        return result

    def _get_receptive_field(self, axis, position, output_id):
        return self._field_of(self.relations[output_id][0], axis, position, {})

    def get_receptive_fields(self):
        assert self.is_initialized, 'The model must be initialized before calling this method'
        assert self.is_fully_convolutional(), \
            'This method is only available for Fully Convolutional Networks (FCN)'
        memo, result = {}, {}
        for name, layer in self.layers.items():
            if not layer.changes_receptive_field():
                continue
            fy, fx = self._field_of(name, 0, 0, memo), self._field_of(name, 1, 0, memo)
            result[name] = {}
            for in_id in fy:
                ys, xs = fy[in_id], fx[in_id]
                if not ys or not xs:
                    continue
                result[name][f'input {in_id}'] = {
                    'cnt': (len(ys), len(xs)),
                    'y': (min(ys), max(ys)), 'x': (min(xs), max(xs)),
                    'is_solid_y': len(ys) == max(ys) - min(ys) + 1,
                    'is_solid_x': len(xs) == max(xs) - min(xs) + 1,
                }
        for layer in self.layers.values():
            layer._clear_receptive_fields_info()
        return result

    def init_progress_tracker(self, progress_tracker, model_name='model'):
        if self.name is None:
            self.name = model_name
        self.progress_tracker = progress_tracker
        self.progress_tracker.register_layer(self.name)
        for layer in self.layers.values():
            layer.init_progress_tracker(progress_tracker, None)


class Sequential(Model):
    """models.py:487-502: layer i is named '<i>_<ClassName>' and fed by layer i-1."""

    def __init__(self, layers, *args, **kwargs):
        if not isinstance(layers, list):
            raise TypeError(f'layers argument must be list, found: {type(layers).__name__}')
        named, relations, prev = {}, {}, 0
        for i, layer in enumerate(layers):
            name = f'{i}_{type(layer).__name__}'
            named[name] = layer
            relations[name] = prev
            prev = name
        relations[0] = prev
        super().__init__(layers=named, relations=relations, *args, **kwargs)
