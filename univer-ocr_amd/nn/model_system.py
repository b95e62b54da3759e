"""Model system: a list of components run in order over a shared `context` dict
(reference: nn/model_system.py:1-167, same class names and context keys).

train/test put per-component losses into context['losses'][name], predict puts the raw prediction
into context['prediction'][name]; selectors say where a component reads X / y and writes its
prediction.  The crop/rotate stages between the nets are host code in the reference and plug in
here unchanged as RawFunctionComponent / WrappedFunctionComponent.
"""


class BaseComponent:
    def train(self, context):
        raise NotImplementedError()

    def test(self, context):
        raise NotImplementedError()

    def predict(self, context):
        raise NotImplementedError()


class RawFunctionComponent(BaseComponent):
    def __init__(self, func):
        self.func = func

    def __call__(self, context):
        self.func(context)

    train = test = predict = __call__


class WrappedFunctionComponent(RawFunctionComponent):
    def __init__(self, name, func, *args_labels, **kwargs_labels):
        super().__init__(func)
        self.name, self.args_labels, self.kwargs_labels = name, args_labels, kwargs_labels

    def __call__(self, context):
        args = [context[label] for label in self.args_labels]
        kwargs = {key: context[label] for key, label in self.kwargs_labels.items()}
        context[self.name] = self.func(*args, **kwargs)

    train = test = predict = __call__


class BaseSelector:
    def __init__(self):
        self.context = None

    def __call__(self, context):
        self.context = context

    def get(self):
        raise NotImplementedError()

    def get_X(self):
        raise NotImplementedError()

    def put(self, pred):
        raise NotImplementedError()


class StringSelector(BaseSelector):
    def __init__(self, X_label, y_label, pred_label):
        super().__init__()
        self.X_label, self.y_label, self.pred_label = X_label, y_label, pred_label

    def get(self):
        yield self.context[self.X_label], self.context[self.y_label]

    def get_X(self):
        yield self.context[self.X_label]

    def put(self, pred):
        self.context[self.pred_label] = pred


class IterableSelector(StringSelector):
    def get(self):
        yield from zip(self.context[self.X_label], self.context[self.y_label])

    def get_X(self):
        yield from self.context[self.X_label]

    def put(self, pred):
        self.context.setdefault(self.pred_label, []).append(pred)


class ModelComponent(BaseComponent):
    def __init__(self, name, model, selector, delist_result=False):
        self.name, self.model, self.selector, self.delist_result = name, model, selector, delist_result

    def _publish(self):
        result = [self.model.layers_outputs[k] for k in range(self.model.outputs_count)]
        self.selector.put(result[0] if self.delist_result else result)

    def _accumulate(self, context, losses):
        seen = context['losses'].get(self.name)
        if seen is None:
            context['losses'][self.name] = losses
        else:
            for key, value in losses.items():     # lists concatenate, scalars add (model_system.py:109-111)
                seen[key] += value

    def train(self, context):
        self.selector(context)
        for X, y in self.selector.get():
            self._accumulate(context, self.model.train(X, y))
            self._publish()

    def test(self, context):
        self.selector(context)
        for X, y in self.selector.get():
            self._accumulate(context, self.model.test(X, y))
            self._publish()

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

    def train(self, context):
        context['losses'] = {}
        for component in self.components:
            component.train(context)

    def test(self, context):
        context['losses'] = {}
        for component in self.components:
            component.test(context)

    def predict(self, context):
        context['prediction'] = {}
        for component in self.components:
            component.predict(context)
