"""Array backend of the nn framework on MI355X.

Mirrors the reference's backend switch `CP` (web_app/components/nn/gpu.py:5-29): `CP.cp`,
`CP.is_gpu_used`, `CP.use_gpu()`, `CP.copy(obj)` (host -> device), `CP.asnumpy(obj)` (device ->
host).  Where the reference binds `CP.cp` to CuPy and launches numba.cuda kernels, this backend
owns a `Runtime` (one HIP context of libuniver_hip.so per process = per GPU) and `DeviceArray`s
whose memory comes from PyTorch-ROCm's caching allocator (plumbing only: no torch op computes
anything on the hot path).

There is NO host compute path: `CP.use_cpu()` exists for API compatibility and raises, and every
op raises `HipError` when no MI355X is present.  Without a GPU, DeviceArrays can still be created
(storage-only, on the host) so that model construction, weight I/O and the data-parallel
bucketing logic are testable; any kernel call then fails loudly.
"""
import os

import numpy as np
# One hardware queue per stream (main, the PageTrainer lanes, copy / RCCL streams): ROCm's default of 4 per
# process makes streams share queues and run one after the other.  Read by the HIP runtime when it starts,
# i.e. at the first torch.cuda call after this import; an explicit setting in the environment wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch

from ..hip import lib as hiplib
from ..hip.lib import HipError

_TORCH_DT = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
             np.dtype(np.float16): torch.float16,
             np.dtype(np.uint8): torch.uint8, np.dtype(np.int32): torch.int32}
_CODE = {torch.float32: hiplib.F32, torch.float64: hiplib.F64, torch.float16: hiplib.F16}
_NP_DT = {v: k for k, v in _TORCH_DT.items()}


class Runtime:
    """One HIP context (stream + workspace) of libuniver_hip.so; all kernels of the process go
    through `call`.  The stream is a dedicated torch stream made current, so torch's allocator,
    H2D/D2H copies, HIP-graph capture and RCCL collectives are ordered with the kernels."""

    def __init__(self, device_index=None):
        if not torch.cuda.is_available():
            raise HipError('no HIP device visible: the univer-ocr MI355X backend has no host fallback')
        self.lib = hiplib.get_lib()
        if device_index is None:
            device_index = int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count())
        self.device_index = device_index
        self.device = torch.device('cuda', device_index)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        torch.cuda.set_stream(self.stream)
        import ctypes as C
        handle = C.c_void_p()
        ws = int(os.environ.get('UOCR_WORKSPACE_MB', '256')) << 20
        rc = self.lib.uocr_ctx_create(device_index, ws, C.byref(handle))
        if rc != 0:
            raise HipError(f'uocr_ctx_create(device={device_index}) failed with code {rc}')
        self.ctx = handle
        self._fn = {}
        self.launches = 0
        self.call('uocr_ctx_set_stream', C.c_void_p(self.stream.cuda_stream))
        self._lanes = [(self.ctx, self.stream)]      # lane 0 = the main stream
        self._sides = {}                             # lane ctx -> its side stream (weight gradients), made on demand
        self._deferred = {}                          # lane ctx -> arrays held for its open deferred weight-gradient group
        self.side_on = False                         # Model.backward turns it on for models with side_wgrad

    # -- lanes: extra (context, stream) pairs so independent models run concurrently ---------------
    def add_lane(self, workspace_mb=256, priority=0, xcds=None):
        """A further HIP context of the library with its own stream and workspace.  The four my_model
        nets are independent, so each trains on its own lane: the latency-bound kernels of the small
        nets run under the bandwidth-bound kernels of the large ones.
        xcds: an iterable of XCD numbers (0..7) -- the lane's stream is created by the library on those XCDs only
        (uocr_ctx_create_cu_mask); torch merely wraps it (ExternalStream) for its allocator's bookkeeping."""
        import ctypes as C
        handle = C.c_void_p()
        if xcds is not None:
            cus = self.device_info()['cu_count']
            words = (cus + 31) // 32
            mask = (C.c_uint32 * words)()
            for i in range(cus):
                if i % 8 in set(xcds):
                    mask[i // 32] |= 1 << (i % 32)
            rc = self.lib.uocr_ctx_create_cu_mask(self.device_index, int(workspace_mb) << 20, mask, words, C.byref(handle))
            if rc != 0:
                raise HipError(f'uocr_ctx_create_cu_mask (lane on XCDs {sorted(set(xcds))}) failed with code {rc}')
            stream = torch.cuda.ExternalStream(self.lib.uocr_ctx_get_stream(handle), device=self.device)
            self._lanes.append((handle, stream))
            return len(self._lanes) - 1
        rc = self.lib.uocr_ctx_create(self.device_index, int(workspace_mb) << 20, C.byref(handle))
        if rc != 0:
            raise HipError(f'uocr_ctx_create (lane) failed with code {rc}')
        stream = torch.cuda.Stream(device=self.device, priority=priority)   # -1 = high
        main = self.ctx
        self.ctx = handle
        self.call('uocr_ctx_set_stream', C.c_void_p(stream.cuda_stream))
        self.ctx = main
        self._lanes.append((handle, stream))
        return len(self._lanes) - 1

    def lane(self, index):
        return _Lane(self, index)

    def lane_stream(self, index):
        return self._lanes[index][1]

    def call(self, name, *args):
        fn = self._fn.get(name)
        if fn is None:
            fn = self._fn[name] = getattr(self.lib, name)
        rc = fn(self.ctx, *args)
        self.launches += 1
        if rc != 0:
            msg = self.lib.uocr_last_error(self.ctx)
            raise HipError(f'{name} failed ({rc}): {msg.decode() if msg else "?"}')

    def use_stream(self, torch_stream):
        """Route kernels to another torch stream (used while capturing a HIP graph)."""
        import ctypes as C
        self.call('uocr_ctx_set_stream', C.c_void_p(torch_stream.cuda_stream))

    def set_option(self, key, value):
        """Kernel selection knobs of the C ABI: 'mfma' (0 never / 1 auto / 2 whenever eligible),
        'fast_paths' (0 generic kernels only / 1 shape-specialised).  Applied to every lane."""
        current = self.ctx
        try:
            for handle in [h for h, _ in self._lanes] + [s.handle for s in self._sides.values()]:
                self.ctx = handle
                self.call('uocr_ctx_set_option', key.encode(), int(value))
        finally:
            self.ctx = current
        self._options = getattr(self, '_options', {})
        self._options[key] = int(value)              # (replayed on contexts made later: side streams)

    def set_loss_snapshot(self, arena):
        """From now on the fused optimizer tails launched on the CURRENT lane end by copying `arena`'s slots into the next
        row of its ring (LossArena.arm); None switches it off."""
        if arena is None:
            self.call('uocr_ctx_set_loss_snapshot', None, 0, None, 0, None)
        else:
            self.call('uocr_ctx_set_loss_snapshot', arena.array.ptr, arena.array.shape[0], arena.ring.ptr, arena.ring_len,
                      arena.counter.ptr)

    # -- deferred weight gradients (uocr_wgrad_defer_begin / _flush): one launch for the small GEMMs of a backward pass --
    def defer_wgrad(self):
        """`with rt.defer_wgrad():` -- the weight-gradient GEMMs issued inside on the current lane that are too small
        to fill the chip are recorded by the library and run as ONE grid (+ one reduction) at the end of the block, or
        earlier at `flush_deferred()`.  The arrays they read are held until then (`keep`)."""
        return _DeferScope(self)

    def keep(self, *arrays):
        """Hold `arrays` until the deferred weight gradients of the current lane have been flushed (no-op otherwise)."""
        held = self._deferred.get(self.ctx.value)
        if held is not None:
            held.extend(arrays)

    def flush_deferred(self):
        """Run what has been recorded on the current lane now and keep recording (a gradient bucket is due)."""
        held = self._deferred.get(self.ctx.value)
        if held is not None:
            self.call('uocr_wgrad_defer_flush', 1)
            held.clear()

    # -- side stream of a lane: kernels nobody waits for until the end of the backward pass (weight gradients) --------
    def side(self, *keep):
        """`with rt.side(x, dy):` -- the C-ABI calls inside go to the side stream of the current lane, ordered behind
        everything enqueued on the lane so far (device-side event); the lane does not wait for them until
        `join_side()`.  `keep`: the arrays the side kernels read -- held until the join so that the allocator cannot
        hand their memory to a later kernel of the lane.  No-op (the calls stay on the lane) unless `side_on`."""
        return _SideScope(self, keep)

    def join_side(self):
        """The current lane waits (on the device) for its side stream."""
        side = self._sides.get(self.ctx.value)
        if side is None or not side.dirty:
            return
        lane = self.ctx
        self.ctx = side.handle
        side.join_ev.record()
        self.ctx = lane
        side.join_ev.wait()
        side.dirty = False
        side.keep.clear()

    def synchronize(self):
        self.call('uocr_stream_sync')

    # -- events and HIP graphs of the C ABI (include/univer_hip.h) ------------------------------------------------
    def event(self):
        return Event(self)

    def capture(self, pool=None):
        """`with rt.capture(pool) as graph:` records the C-ABI calls made inside on the CURRENT ctx's stream into a
        HIP graph (uocr_graph_begin_capture / uocr_graph_end_capture); `graph.replay()` launches it on the ctx that
        is current then.  Arrays allocated inside come from `pool` (a torch.cuda.MemPool: memory only -- it keeps the
        graph's buffers away from every other allocation for as long as the pool lives)."""
        return _Capture(self, pool)

    def device_info(self):
        import ctypes as C
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int(), C.c_size_t()
        self.call('uocr_device_info', name, 256, C.byref(cus), C.byref(hbm))
        return {'name': name.value.decode(), 'cu_count': cus.value, 'hbm_bytes': hbm.value}


class Event:
    """HIP event of the C ABI: record() on the current ctx's stream, wait() = the current ctx's stream waits for it
    on the device, synchronize() = the host waits."""

    __slots__ = ('rt', 'handle')

    def __init__(self, rt):
        import ctypes as C
        self.rt, self.handle = rt, C.c_void_p()
        if rt.lib.uocr_event_create(C.byref(self.handle)) != 0:
            raise HipError('uocr_event_create failed')

    def record(self):
        self.rt.call('uocr_event_record', self.handle)
        return self

    def wait(self):
        self.rt.call('uocr_stream_wait_event', self.handle)

    def synchronize(self):
        if self.rt.lib.uocr_event_synchronize(self.handle) != 0:
            raise HipError('uocr_event_synchronize failed')

    def __del__(self):
        try:
            if self.handle:
                self.rt.lib.uocr_event_destroy(self.handle)
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass


class Graph:
    """An instantiated HIP graph of the C ABI; keeps its memory pool alive."""

    __slots__ = ('rt', 'handle', 'pool')

    def __init__(self, rt, handle, pool):
        self.rt, self.handle, self.pool = rt, handle, pool

    def replay(self):
        self.rt.call('uocr_graph_launch', self.handle)

    def __del__(self):
        try:
            if self.handle:
                self.rt.lib.uocr_graph_destroy(self.handle)
        except Exception:   # noqa: BLE001
            pass


class _Capture:
    def __init__(self, rt, pool):
        self.rt, self.pool, self.graph = rt, pool, None

    def __enter__(self):
        self._mem = torch.cuda.use_mem_pool(self.pool) if self.pool is not None else None
        if self._mem is not None:
            self._mem.__enter__()
        torch.cuda.synchronize()
        self.rt.call('uocr_graph_begin_capture')
        self.graph = Graph(self.rt, None, self.pool)
        return self.graph

    def __exit__(self, exc_type, exc, tb):
        import ctypes as C
        handle = C.c_void_p()
        try:
            self.rt.join_side()                      # a forked side stream must be back before the capture ends
            self.rt.call('uocr_graph_end_capture', C.byref(handle))
            self.graph.handle = handle
        except HipError:
            if exc_type is None:
                raise
        finally:
            if self._mem is not None:
                self._mem.__exit__(exc_type, exc, tb)
        return False


class LossArena:
    """Consecutive float64 loss slots in ONE device array: the loss kernels of a captured step write side by side.
    Per-step snapshots: `arm()` gives the arena a ring of `ring_len` rows; with Runtime.set_loss_snapshot(arena) in force
    the step's last kernel (the fused optimizer tail) copies the slots into the next row by itself
    (uocr_ctx_set_loss_snapshot), and `next_row()` names that row on the host -- no copy launch.  A row is overwritten
    `ring_len` steps later."""

    def __init__(self, capacity=16, ring_len=64):
        self.array = CP.empty((capacity,), np.float64)
        self.used = 0
        self.ring_len = ring_len
        self.ring = self.counter = None
        self.rows = 0                                # optimizer tails launched since arm() (the device counts the same)

    def take(self):
        if self.used >= self.array.shape[0]:
            raise HipError('loss arena full')
        slot = DeviceArray(self.array.t[self.used:self.used + 1])
        self.used += 1
        return slot

    def index_of(self, tensor):
        return (tensor.data_ptr() - self.array.ptr) // 8

    def snapshot(self):
        """a copy of the used slots, made on the current ctx's stream (one uocr_d2d)"""
        out = CP.empty((self.used,), np.float64)
        CP.runtime().call('uocr_d2d', out.ptr, self.array.ptr, 8 * self.used)
        return out

    def arm(self):
        """allocate the ring and its device counter (not inside a graph capture: the counter is zeroed once)"""
        capacity = self.array.shape[0]
        self.ring = CP.empty((self.ring_len * capacity,), np.float64)
        self.counter = CP.zeros((1,), np.int32)
        self.rows = 0
        return self

    def next_row(self):
        """the ring row the optimizer tail that was just launched writes: a view, valid for ring_len further steps"""
        capacity = self.array.shape[0]
        k = self.rows % self.ring_len
        self.rows += 1
        return DeviceArray(self.ring.t[k * capacity:(k + 1) * capacity])


class _DeferScope:
    def __init__(self, rt):
        self.rt = rt

    def __enter__(self):
        self.key = self.rt.ctx.value
        if self.key in self.rt._deferred:
            raise HipError('defer_wgrad: a deferred group is already open on this lane')
        self.rt.call('uocr_wgrad_defer_begin')
        self.rt._deferred[self.key] = []

    def __exit__(self, exc_type, exc, tb):
        held = self.rt._deferred.pop(self.key)
        try:
            self.rt.call('uocr_wgrad_defer_flush', 0)
        except HipError:
            if exc_type is None:
                raise
        finally:
            held.clear()
        return False


class _Side:
    """Side stream of one lane: its own uocr_ctx (stream + workspace)."""

    def __init__(self, rt):
        import ctypes as C
        self.handle = C.c_void_p()
        rc = rt.lib.uocr_ctx_create(rt.device_index, int(os.environ.get('UOCR_SIDE_WORKSPACE_MB', '128')) << 20,
                                    C.byref(self.handle))
        if rc != 0:
            raise HipError(f'uocr_ctx_create (side stream) failed with code {rc}')
        self.stream = torch.cuda.Stream(device=rt.device)
        lane = rt.ctx
        rt.ctx = self.handle
        rt.call('uocr_ctx_set_stream', C.c_void_p(self.stream.cuda_stream))
        for key, value in getattr(rt, '_options', {}).items():
            rt.call('uocr_ctx_set_option', key.encode(), value)
        rt.ctx = lane
        self.fork_ev, self.join_ev = Event(rt), Event(rt)
        self.keep, self.dirty = [], False


class _SideScope:
    def __init__(self, rt, keep):
        self.rt, self.keep = rt, keep

    def __enter__(self):
        rt = self.rt
        self.lane = None
        if not rt.side_on:
            return
        side = rt._sides.get(rt.ctx.value)
        if side is None:
            side = rt._sides[rt.ctx.value] = _Side(rt)
        side.fork_ev.record()                      # on the lane
        self.lane = rt.ctx
        rt.ctx = side.handle
        side.fork_ev.wait()                        # the side stream waits for the lane up to here
        side.keep.extend(self.keep)
        side.dirty = True

    def __exit__(self, *exc):
        if self.lane is not None:
            self.rt.ctx = self.lane
        return False


class _Lane:
    """`with runtime.lane(i):` -- kernels, allocations and copies inside go to lane i's stream."""

    def __init__(self, rt, index):
        self.rt, self.index = rt, index

    def __enter__(self):
        self.prev_ctx, self.prev_stream = self.rt.ctx, torch.cuda.current_stream()
        self.rt.ctx, stream = self.rt._lanes[self.index]
        torch.cuda.set_stream(stream)
        return stream

    def __exit__(self, *exc):
        self.rt.ctx = self.prev_ctx
        torch.cuda.set_stream(self.prev_stream)
        return False


class DeviceArray:
    """N-d C-contiguous array in HBM (NHWC for images).  Only what the framework needs: shape /
    dtype / reshape (a view) / copy / host round trip / `+` for fan-out gradient sums
    (models.py:218).  All arithmetic happens in libuniver_hip.so."""

    __slots__ = ('t', 'gscale')
    __array_priority__ = 100

    def __init__(self, tensor, gscale=0):
        self.t = tensor
        # float16 mode: log2 of the power-of-two factor this array carries when it is an activation GRADIENT
        # (set by the loss kernels, handed on by every backward op, removed by the bwd_weight kernels:
        # include/univer_hip.h, UOCR_F16_SCALED).  0 for everything else.
        self.gscale = gscale

    # -- metadata ---------------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def dtype(self):
        return _NP_DT[self.t.dtype]

    @property
    def size(self):
        return self.t.numel()

    @property
    def ndim(self):
        return self.t.dim()

    @property
    def nbytes(self):
        return self.t.numel() * self.t.element_size()

    @property
    def ptr(self):
        return self.t.data_ptr()

    @property
    def code(self):
        code = _CODE.get(self.t.dtype)
        if code is None:
            raise HipError(f'{self.t.dtype} arrays have no kernel dtype code')
        return code | (self.gscale << 8) if code == hiplib.F16 else code

    def __len__(self):
        return self.t.shape[0]

    def __repr__(self):
        return f'DeviceArray(shape={self.shape}, dtype={self.dtype}, device={self.t.device})'

    # -- views / copies -----------------------------------------------------------------------
    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return DeviceArray(self.t.view(*[int(s) for s in shape]), self.gscale)

    def copy(self):
        return DeviceArray(self.t.clone(), self.gscale)

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def tolist(self):
        return self.numpy().tolist()

    def set(self, host):
        """Overwrite the contents with a host array of the same shape (keeps the storage)."""
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.shape != self.shape:
            raise ValueError(f'shape mismatch: {host.shape} != {self.shape}')
        self.t.copy_(torch.from_numpy(host))
        return self

    def __add__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return self
        if not isinstance(other, DeviceArray):
            return NotImplemented
        return CP.ops.add(self, other)

    __radd__ = __add__


class _Random:
    """CP.cp.random.* of the test scripts (test_gradients.py:65-70): host RNG, then H2D."""

    @staticmethod
    def randn(*shape):
        return CP.copy(np.random.randn(*shape))

    @staticmethod
    def rand(*shape):
        return CP.copy(np.random.rand(*shape))

    @staticmethod
    def uniform(low=0.0, high=1.0, size=None):
        return CP.copy(np.random.uniform(low, high, size))


class _ArrayModule:
    """The slice of the NumPy/CuPy module API the framework and its test scripts use on `CP.cp`."""
    random = _Random()
    ndarray = DeviceArray

    @staticmethod
    def zeros(shape, dtype=None):
        return CP.zeros(shape, dtype)

    @staticmethod
    def ones(shape, dtype=None):
        return CP.full(shape, 1.0, dtype)

    @staticmethod
    def zeros_like(a):
        return CP.zeros(a.shape, a.dtype)

    @staticmethod
    def array(obj, dtype=None):
        return CP.copy(obj, dtype)

    asarray = array

    @staticmethod
    def reshape(a, shape):
        return a.reshape(shape)

    @staticmethod
    def copy(a):
        return a.copy()


class CP:
    """Backend switch with the reference's interface (nn/gpu.py:5-29)."""
    cp = _ArrayModule()
    is_gpu_used = True
    dtype = np.dtype(os.environ.get('UOCR_DTYPE', 'float32'))
    lazy_losses = False          # True: losses stay on the device until float() is called
    loss_arena = None            # a LossArena: loss slots come from it (graph capture, my_model/trainer.py)
    f16_grad_scale_log2 = None   # float16 mode: None = the loss kernels pick the gradient scale, int k = 2^k
    _runtime = None
    ops = None                   # set by nn/ops.py (kernel wrappers)

    @staticmethod
    def use_cpu():
        raise NotImplementedError(
            'CP.use_cpu(): the MI355X backend ships no host compute path. The NumPy mode of the '
            'reference (nn/gpu.py:9-12) is the reference itself; tests compare against oracle/.')

    @staticmethod
    def use_gpu(device_index=None):
        CP.is_gpu_used = True
        CP.runtime(device_index)

    @staticmethod
    def set_dtype(dtype):
        """float32 (production), float64 (the reference's type: parity mode) or float16 = binary16 ACTIVATIONS
        with float32 parameters / gradients / accumulation (BASELINE configs[4], include/univer_hip.h UOCR_F16)."""
        dt = np.dtype(dtype)
        if dt not in (np.dtype(np.float32), np.dtype(np.float64), np.dtype(np.float16)):
            raise HipError(f'unsupported compute dtype {dt}')
        CP.dtype = dt

    @staticmethod
    def param_dtype():
        """dtype of parameters, their gradients and optimizer state: float32 master copies in float16 mode."""
        return np.dtype(np.float32) if CP.dtype == np.dtype(np.float16) else CP.dtype

    @staticmethod
    def runtime(device_index=None):
        if CP._runtime is None:
            CP._runtime = Runtime(device_index)
        return CP._runtime

    @staticmethod
    def has_device():
        return torch.cuda.is_available()

    @staticmethod
    def storage_device():
        if CP._runtime is not None:
            return CP._runtime.device
        if torch.cuda.is_available():
            return CP.runtime().device
        return torch.device('cpu')      # storage-only mode (no kernels can run)

    @staticmethod
    def loss_slot():
        """one float64 device slot a loss / regularisation kernel writes its value to"""
        if CP.loss_arena is not None:
            return CP.loss_arena.take()
        return CP.empty((1,), np.float64)

    # -- allocation -----------------------------------------------------------------------------
    @staticmethod
    def empty(shape, dtype=None):
        dt = CP.dtype if dtype is None else np.dtype(dtype)
        shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(int(s) for s in shape)
        return DeviceArray(torch.empty(shape, dtype=_TORCH_DT[dt], device=CP.storage_device()))

    @staticmethod
    def zeros(shape, dtype=None):
        """cupy.zeros: allocation (torch's caching allocator) + uocr_memset_zero -- no torch kernel runs."""
        out = CP.empty(shape, dtype)
        if out.t.is_cuda:
            if out.size:
                CP.runtime().call('uocr_memset_zero', out.ptr, out.nbytes)
        else:
            out.t.zero_()                   # storage-only mode (no GPU): host memory
        return out

    @staticmethod
    def full(shape, value, dtype=None):
        out = CP.empty(shape, dtype)
        if out.t.is_cuda and out.t.dtype in _CODE:
            if out.size:
                CP.runtime().call('uocr_fill', _CODE[out.t.dtype], out.ptr, float(value), out.size)
        else:
            out.t.fill_(value)
        return out

    # -- host <-> device (gpu.py:19-29) ------------------------------------------------------------
    @staticmethod
    def copy(obj, dtype=None):
        """Host array (or DeviceArray) -> new DeviceArray in the compute dtype."""
        dt = CP.dtype if dtype is None else np.dtype(dtype)
        if isinstance(obj, DeviceArray):
            if obj.dtype == dt:
                return obj.copy()
            return DeviceArray(obj.t.to(_TORCH_DT[dt]))
        host = np.ascontiguousarray(np.asarray(obj), dtype=dt)
        return DeviceArray(torch.from_numpy(host).to(CP.storage_device()))

    @staticmethod
    def asnumpy(obj):
        if isinstance(obj, DeviceArray):
            return obj.numpy()
        return np.asarray(obj)


class DeviceScalar:
    """A float64 loss value that lives in a device slot until somebody needs the number.  The
    reference converts every loss with float() right away (losses.py:25,42,57,73 -- one host sync
    each); with CP.lazy_losses the sync happens once, when the caller formats / adds the value."""

    __slots__ = ('t', '_value', 'ready')

    def __init__(self, tensor, ready=None):
        self.t = tensor          # 0-d or 1-element float64 tensor
        self._value = None
        self.ready = ready       # event of the stream that writes the slot, when that is not the reader's stream

    def __float__(self):
        if self._value is None:
            if self.ready is not None:
                self.ready.synchronize()
            self._value = float(self.t.item())
        return self._value

    def __add__(self, other):
        return float(self) + float(other)

    __radd__ = __add__

    def __sub__(self, other):
        return float(self) - float(other)

    def __rsub__(self, other):
        return float(other) - float(self)

    def __mul__(self, other):
        return float(self) * float(other)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return float(self) / float(other)

    def __lt__(self, other):
        return float(self) < float(other)

    def __gt__(self, other):
        return float(self) > float(other)

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))
