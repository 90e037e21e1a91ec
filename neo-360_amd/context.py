"""Per-device library context and tensor marshalling helpers."""
import ctypes
import os
import weakref

import torch

from . import _lib

_contexts = {}


def _require_gpu(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise _lib.NeoError(
            "%s is on %s: the neo360_amd path runs only on a ROCm device (there is no CPU fallback)" % (name, t.device))


def f32(t, name="tensor"):
    """Dense fp32 device tensor (copies only if needed)."""
    _require_gpu(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def f64(t, name="tensor"):
    _require_gpu(t, name)
    if t.dtype != torch.float64:
        t = t.double()
    return t.contiguous()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream_of(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Context:
    """Owns one `neo_ctx` (packed weights, scene features, workspaces) on one device."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.NeoError("neo360_amd needs a ROCm device, got %s" % self.device)
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self.lib.neo_ctx_create(self.index, ctypes.byref(h)))
        self.handle = h
        self.uploaded = {}   # slot key -> fingerprint of the parameters packed on the device
        # the context owns device memory (packed weights, scene features, grow-only workspaces: ~5 GB after a 640x480
        # NeO-360 frame): release it when this object is collected, not at interpreter exit
        self._finalizer = weakref.finalize(self, Context._destroy, self.lib, h)

    @staticmethod
    def _destroy(lib, handle):
        try:
            lib.neo_ctx_destroy(handle)
        except Exception:       # interpreter shutdown: the driver frees everything anyway
            pass

    def stream(self):
        return stream_of(self.device)

    def poll_flags(self):
        flags = ctypes.c_uint32(0)
        _lib.check(self.lib.neo_ctx_poll_flags(self.handle, ctypes.byref(flags), self.stream()))
        return flags.value

    def post_flags(self):
        """Enqueue a read-and-clear of the assertion word on the current stream; no synchronisation."""
        _lib.check(self.lib.neo_ctx_post_flags(self.handle, self.stream()))

    def take_flags(self, wait=False):
        """OR of the posted reads that have completed (wait=False) / of all posted reads (wait=True: blocks)."""
        flags, pending = ctypes.c_uint32(0), ctypes.c_int(0)
        _lib.check(self.lib.neo_ctx_take_flags(self.handle, 1 if wait else 0, self.stream(), ctypes.byref(flags),
                                               ctypes.byref(pending)))
        return flags.value

    def sync_count(self):
        """Blocking waits (stream / event synchronisations) the flag calls of this context have issued."""
        n = ctypes.c_uint64(0)
        _lib.check(self.lib.neo_ctx_sync_count(self.handle, ctypes.byref(n)))
        return n.value

    def stream_waits(self):
        """Cross-stream ordering waits this context has inserted (calls arriving on a stream other than the previous call's)."""
        n = ctypes.c_uint64(0)
        _lib.check(self.lib.neo_ctx_stream_waits(self.handle, ctypes.byref(n)))
        return n.value

    def set_lane(self, lane):
        """Scratch lane (0 / 1) the next calls of this context use (neo_ctx_set_lane)."""
        if getattr(self, "_lane", 0) != lane:
            _lib.check(self.lib.neo_ctx_set_lane(self.handle, int(lane)))
            self._lane = lane

    def set_ray_grid(self, width, first_ray=0):
        """Pixel-grid hint for the next whole-frame renders (neo_ctx_set_ray_grid); width 0 removes it."""
        key = (int(width), int(first_ray) if width else 0)
        if getattr(self, "_ray_grid", (0, 0)) != key:
            _lib.check(self.lib.neo_ctx_set_ray_grid(self.handle, key[0], key[1]))
            self._ray_grid = key

    def set_precision(self, mode):
        """'f32' (exact fp32 MFMA) or 'f16x3' (fp16 MFMA, hi/lo-split operands, fp32-equivalent)."""
        code = {"f32": 0, "f16x3": 1, 0: 0, 1: 1}[mode]
        _lib.check(self.lib.neo_ctx_set_precision(self.handle, code))

    def set_timing(self, enable):
        _lib.check(self.lib.neo_ctx_set_timing(self.handle, 1 if enable else 0))

    def read_timing(self):
        ms, n, pts, fl = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
        _lib.check(self.lib.neo_ctx_read_timing(self.handle, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(pts),
                                                ctypes.byref(fl)))
        return ms.value, n.value, pts.value, fl.value

    KERNEL_NAMES = {0: "unspecified", 1: "k_tp_mlp_hp", 2: "k_tp_mlp_hpp", 3: "k_tp_mlp_h", 4: "k_tp_mlp",
                    5: "k_mip_mlp_h<256> (proposal MLP)", 6: "k_mip_mlp_h<1024> (NeRF MLP, fused)",
                    7: "k_mip_ipe_h + 8 x k_mip_gemm_h + k_mip_mlp_h<tail> (NeRF MLP, layer by layer)", 8: "k_mip_mlp",
                    9: "k_pix_mlp"}

    def read_spans(self):
        """Every timed evaluator launch since set_timing(True), in order: [(ms, kernel name, points, algorithmic flops)]."""
        n = ctypes.c_int(0)
        _lib.check(self.lib.neo_ctx_read_spans(self.handle, 0, None, None, None, None, ctypes.byref(n)))
        cap = n.value
        ms, kid = (ctypes.c_double * cap)(), (ctypes.c_int * cap)()
        pts, fl = (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
        _lib.check(self.lib.neo_ctx_read_spans(self.handle, cap, ms, kid, pts, fl, ctypes.byref(n)))
        return [(ms[i], self.KERNEL_NAMES.get(kid[i], str(kid[i])), pts[i], fl[i]) for i in range(min(cap, n.value))]

    def close(self):
        if self.handle:
            self._finalizer.detach()
            self.lib.neo_ctx_destroy(self.handle)
            self.handle = None


_SIDE_STREAMS = {}


class CallOverlap:
    """Lets consecutive whole-chunk calls of one module overlap on the device (round 6, VERDICT r5 task 4).

    The reference renders a frame as 300 `model(chunk)` calls of 1024 rays (neo360/model.py:861-907).  On one stream each
    call's four evaluator launches end on a partly filled machine and the next call cannot start before the last workgroup of
    this one has finished.  Here call i runs on side stream i % 2 with scratch lane i % 2 of the context, so call i + 1's
    kernels fill the CUs call i's tail leaves idle.  Stream semantics towards the CALLER are unchanged:

    * fork: the side stream waits for an event of the caller's stream.  A fresh event is recorded at every call, EXCEPT when
      the call's ray tensors are views of the very tensor objects (weak identity), at the very versions, of the call the last
      fresh event was recorded for, on the same caller stream and with no `out=` write of this library in between - then
      everything these rays depend on was already enqueued before that event and it is reused.  (A fresh event at call
      i + 1 would sit behind the join of call i and serialise the two.)  Any converted / copied input takes a fresh event.
    * join: before the call returns, the caller's stream waits for the call's completion event, so anything enqueued on it
      afterwards sees the outputs - exactly as if the kernels had run there.
    * memory: outputs are allocated under the side stream and `record_stream`-ed to the caller's stream, inputs are
      `record_stream`-ed to the side stream: the caching allocator will not hand a block to another stream while the other
      side may still touch it.
    """

    def __init__(self, device):
        self.device = device
        self.lanes = max(1, min(4, int(os.environ.get("NEO360_LANES", "2"))))      # calls in flight (library scratch lanes: up to 4)
        # ONE set of side streams per device for every module of the process: HIP multiplexes streams onto a few hardware queues
        # in creation order, and two side streams that land on one queue do not overlap at all (measured: a module whose private
        # pair collided ran its chunk loop at 0.85 of the frame rate where another module's pair gave 0.98)
        pool = _SIDE_STREAMS.setdefault(torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), [])
        while len(pool) < self.lanes:
            pool.append(torch.cuda.Stream(device))
        self.streams = pool[:self.lanes]
        self.done = [torch.cuda.Event() for _ in range(self.lanes)]
        self.calls = 0
        self.fresh_forks = 0
        self._fork_ev = None
        self._fork_key = None
        self._fork_bases = ()

    def _key(self, raw, conv, cur):
        bases, ks = [], []
        for a, b in zip(raw, conv):
            if a is not b:
                return None, ()
            base = a._base if a._base is not None else a
            bases.append(base)
            ks.append((base._version, a.untyped_storage().data_ptr()))
        return (tuple(ks), cur.cuda_stream, _lib.write_epoch), tuple(bases)

    def begin(self, raw, conv):
        """-> (side stream, lane, caller's stream).  The side stream is ordered behind everything the inputs depend on."""
        cur = torch.cuda.current_stream(self.device)
        lane = self.calls % self.lanes
        self.calls += 1
        side = self.streams[lane]
        key, bases = self._key(raw, conv, cur)
        reuse = (key is not None and key == self._fork_key and len(bases) == len(self._fork_bases)
                 and all(r() is b for r, b in zip(self._fork_bases, bases)))
        if not reuse:
            ev = torch.cuda.Event()
            ev.record(cur)
            self._fork_ev, self._fork_key = ev, key
            self._fork_bases = tuple(weakref.ref(b) for b in bases)
            self.fresh_forks += 1
        side.wait_event(self._fork_ev)
        for t in conv:
            t.record_stream(side)
        return side, lane, cur

    def end(self, lane, cur, outputs):
        """Completion event on the side stream; the caller's stream waits for it; outputs may be used (and freed) there."""
        ev = self.done[lane]
        ev.record(self.streams[lane])
        cur.wait_event(ev)
        for t in outputs:
            t.record_stream(cur)


def get_context(device):
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ctx = _contexts.get(idx)
    if ctx is None:
        ctx = _contexts[idx] = Context(torch.device("cuda", idx))
    return ctx


def new_context(device):
    """A private context (own packed weights / scene), e.g. one per nn.Module."""
    return Context(device)
