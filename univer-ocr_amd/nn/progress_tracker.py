"""Per-layer timing hooks (reference: nn/progress_tracker.py:5-126, same public names).

`track_method('forward'/'backward')` wraps every layer call; with the default BaseProgressTracker
it costs two no-op calls.  Kernels are asynchronous, so wall time per layer is only meaningful with
`ProgressTracker(sync=True)`, which synchronises the HIP stream at every stop (the reference
synchronises after every custom kernel anyway: convolutional.py:192,278)."""
from datetime import datetime
from functools import wraps


class Event:
    def __init__(self, name):
        self.name = name
        self.reset()

    def reset(self):
        self.done, self.started, self.stopped, self.time, self.counter = False, None, None, None, 0

    def start(self):
        self.done, self.started = False, datetime.now()

    def stop(self):
        self.stopped = datetime.now()
        elapsed = self.stopped - self.started
        self.time = elapsed if self.time is None else self.time + elapsed
        self.done = True
        self.counter += 1

    def to_dict(self):
        return {key: getattr(self, key) for key in ('name', 'done', 'started', 'stopped', 'time', 'counter')}


class BaseProgressTracker:
    def __init__(self, *args, **kwargs):
        pass

    def register_layer(self, name):
        pass

    def get_summary(self):
        return {}

    def start_tracking(self, name, event):
        pass

    def stop_tracking(self, name, event):
        pass

    def message(self, message, data=None):
        pass

    def reset(self):
        pass


class ProgressTracker(BaseProgressTracker):
    def __init__(self, handler=print, sync=False):
        self.layers = {}
        self.handler = handler
        self.sync = sync

    def register_layer(self, name):
        self.layers[name] = {}

    def get_summary(self):
        return {name: [ev.to_dict() for ev in events.values()] for name, events in self.layers.items()}

    def start_tracking(self, name, event):
        events = self.layers.setdefault(name, {})
        if event not in events:
            events[event] = Event(event)
        events[event].start()
        self.handler(event, self.get_summary())

    def stop_tracking(self, name, event):
        if self.sync:
            from .gpu import CP
            CP.runtime().synchronize()
        self.layers[name][event].stop()
        self.handler(event, self.get_summary())

    def message(self, message, data=None):
        self.handler(message, data)

    def reset(self):
        self.handler('reset')
        for events in self.layers.values():
            for ev in events.values():
                ev.reset()


def track_method(event):
    def decorator(func):
        @wraps(func)
        def wrapper(self, *args, **kwargs):
            tracker = self.progress_tracker
            tracker.start_tracking(self.name, event)
            result = func(self, *args, **kwargs)
            tracker.stop_tracking(self.name, event)
            return result
        return wrapper
    return decorator


def track_function(name, event, progress_tracker):
    if progress_tracker is None:
        return lambda func: func
    progress_tracker.register_layer(name)

    def decorator(func):
        @wraps(func)
        def wrapper(*args, **kwargs):
            progress_tracker.start_tracking(name, event)
            result = func(*args, **kwargs)
            progress_tracker.stop_tracking(name, event)
            return result
        return wrapper
    return decorator
