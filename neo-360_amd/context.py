"""Per-device library context and tensor marshalling helpers."""
import ctypes
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
