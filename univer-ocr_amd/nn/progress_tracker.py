"""Per-layer timing hooks with the public names of the reference's nn/progress_tracker.py
(`Event`, `BaseProgressTracker`, `ProgressTracker`, `track_method`, `track_function`).

Every layer's forward/backward is wrapped by `track_method`; with the default (null) tracker that
is two no-op calls.  Kernels are launched asynchronously, so the wall time of a layer call is only
meaningful with `ProgressTracker(sync=True)`, which synchronises the HIP stream when an event stops
(the reference synchronises after every custom kernel anyway: layers/convolutional.py:192,278).
The summary handed to the handler keeps the reference's dictionary keys (name, done, started,
stopped, time, counter) because the web UI renders them."""
import functools
from collections import defaultdict
from datetime import datetime, timedelta

_FIELDS = ('name', 'done', 'started', 'stopped', 'time', 'counter')


class Event:
    """Accumulated wall time and call count of one (layer, phase) pair."""

    def __init__(self, name):
        self.name = name
        self.reset()

    def reset(self):
        self.done = False
        self.started = self.stopped = self.time = None
        self.counter = 0

    def start(self):
        self.done = False
        self.started = datetime.now()

    def stop(self):
        self.stopped = datetime.now()
        self.time = (self.time or timedelta(0)) + (self.stopped - self.started)
        self.counter += 1
        self.done = True

    def to_dict(self):
        return {field: getattr(self, field) for field in _FIELDS}


class BaseProgressTracker:
    """Null object: accepts every call of the interface and records nothing."""

    def __init__(self, *args, **kwargs):
        pass

    def get_summary(self):
        return {}

    def _ignore(self, *args, **kwargs):
        return None

    register_layer = start_tracking = stop_tracking = message = reset = _ignore


class ProgressTracker(BaseProgressTracker):
    def __init__(self, handler=print, sync=False):
        self.handler = handler
        self.sync = sync
        self.layers = defaultdict(dict)          # layer name -> {phase: Event}

    def register_layer(self, name):
        self.layers[name] = {}

    def get_summary(self):
        return {layer: [ev.to_dict() for ev in phases.values()] for layer, phases in self.layers.items()}

    def _event(self, layer, phase):
        phases = self.layers[layer]
        if phase not in phases:
            phases[phase] = Event(phase)
        return phases[phase]

    def start_tracking(self, name, event):
        self._event(name, event).start()
        self.handler(event, self.get_summary())

    def stop_tracking(self, name, event):
        if self.sync:
            from .gpu import CP
            CP.runtime().synchronize()
        self._event(name, event).stop()
        self.handler(event, self.get_summary())

    def message(self, message, data=None):
        self.handler(message, data)

    def reset(self):
        self.handler('reset')
        for phases in self.layers.values():
            for ev in phases.values():
                ev.reset()


def _tracked(func, tracker_of, name_of, phase):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        tracker = tracker_of(args)
        if type(tracker) is BaseProgressTracker:      # the do-nothing default: no bookkeeping on the hot path
            return func(*args, **kwargs)
        name = name_of(args)
        tracker.start_tracking(name, phase)
        try:
            return func(*args, **kwargs)
        finally:
            tracker.stop_tracking(name, phase)
    return wrapper


def track_method(event):
    """Decorator for layer methods: uses `self.progress_tracker` and `self.name`."""
    def decorator(func):
        return _tracked(func, lambda a: a[0].progress_tracker, lambda a: a[0].name, event)
    return decorator


def track_function(name, event, progress_tracker):
    """Decorator for free functions (the host stages of a model system)."""
    if progress_tracker is None:
        return lambda func: func
    progress_tracker.register_layer(name)

    def decorator(func):
        return _tracked(func, lambda a: progress_tracker, lambda a: name, event)
    return decorator
