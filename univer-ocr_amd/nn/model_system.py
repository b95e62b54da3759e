"""Model system: an ordered list of components that read and write one shared `context` dict.

Same public names and context conventions as the reference's nn/model_system.py.  A ModelComponent
runs one model on the arrays its selector picks out of the context, files the losses under
context['losses'][name] (predict: the raw prediction under context['prediction'][name]) and hands
the model outputs back to the selector; function components are the reference's host stages (crop,
rotate, label) and plug in unchanged."""

_SLOT_OF_MODE = {'train': 'losses', 'test': 'losses', 'predict': 'prediction'}


# ---- selectors: where a component finds X / y and where its prediction goes -----------------------
class BaseSelector:
    def __init__(self):
        self.context = None

    def __call__(self, context):
        """Bind to the context of the current step."""
        self.context = context

    def get(self):
        raise NotImplementedError()

    def get_X(self):
        raise NotImplementedError()

    def put(self, pred):
        raise NotImplementedError()


class StringSelector(BaseSelector):
    """One sample per step, addressed by three context keys."""

    def __init__(self, X_label, y_label, pred_label):
        BaseSelector.__init__(self)
        self.X_label = X_label
        self.y_label = y_label
        self.pred_label = pred_label

    def get(self):
        ctx = self.context
        yield ctx[self.X_label], ctx[self.y_label]

    def get_X(self):
        yield self.context[self.X_label]

    def put(self, pred):
        self.context[self.pred_label] = pred


class IterableSelector(StringSelector):
    """The context keys hold lists of samples; predictions are collected in a list."""

    def get(self):
        ctx = self.context
        for pair in zip(ctx[self.X_label], ctx[self.y_label]):
            yield pair

    def get_X(self):
        for X in self.context[self.X_label]:
            yield X

    def put(self, pred):
        collected = self.context.get(self.pred_label)
        if collected is None:
            collected = self.context[self.pred_label] = []
        collected.append(pred)


# ---- components ------------------------------------------------------------------------------------
class BaseComponent:
    def train(self, context):
        raise NotImplementedError()

    def test(self, context):
        raise NotImplementedError()

    def predict(self, context):
        raise NotImplementedError()


class RawFunctionComponent(BaseComponent):
    """A host stage: the same `func(context)` in every mode."""

    def __init__(self, func):
        self.func = func

    def __call__(self, context):
        self.func(context)

    def train(self, context):
        self(context)

    def test(self, context):
        self(context)

    def predict(self, context):
        self(context)


class WrappedFunctionComponent(RawFunctionComponent):
    """A host stage given as a plain function of context entries; its result is stored under `name`."""

    def __init__(self, name, func, *args_labels, **kwargs_labels):
        RawFunctionComponent.__init__(self, func)
        self.name = name
        self.args_labels = args_labels
        self.kwargs_labels = kwargs_labels

    def __call__(self, context):
        positional = [context[label] for label in self.args_labels]
        named = {key: context[label] for key, label in self.kwargs_labels.items()}
        context[self.name] = self.func(*positional, **named)


def _file_losses(store, name, losses):
    """First sample of a step creates the entry; further samples extend its lists / add its scalars."""
    if name not in store:
        store[name] = losses
        return
    entry = store[name]
    for key in losses:
        entry[key] += losses[key]


class ModelComponent(BaseComponent):
    def __init__(self, name, model, selector, delist_result=False):
        self.name = name
        self.model = model
        self.selector = selector
        self.delist_result = delist_result

    def _publish(self):
        model = self.model
        outputs = [model.layers_outputs[index] for index in range(model.outputs_count)]
        self.selector.put(outputs[0] if self.delist_result else outputs)

    def _supervised(self, context, step):
        self.selector(context)
        for X, y in self.selector.get():
            _file_losses(context['losses'], self.name, step(X, y))
            self._publish()

    def train(self, context):
        self._supervised(context, self.model.train)

    def test(self, context):
        self._supervised(context, self.model.test)

    def predict(self, context):
        self.selector(context)
        for X in self.selector.get_X():
            context['prediction'][self.name] = self.model.predict(X)
            self._publish()


class ModelSystem:
    def __init__(self, components):
        assert isinstance(components, list)
        assert all(isinstance(c, BaseComponent) for c in components)
        self.components = components

    def _each(self, mode, context):
        context[_SLOT_OF_MODE[mode]] = {}
        for component in self.components:
            getattr(component, mode)(context)

    def train(self, context):
        self._each('train', context)

    def test(self, context):
        self._each('test', context)

    def predict(self, context):
        self._each('predict', context)
