"""ctypes binding of libneo360_hip.so (include/neo360_hip.h).

This is the binding a maintainer of the reference would add:
plain pointers and sizes, tensors passed as `tensor.data_ptr()`.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# $NEO360_HIP_LIB points the loader at another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("NEO360_HIP_LIB") or os.path.join(_HERE, "lib", "libneo360_hip.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float


class MipLevelOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("rgb", "sdist", "weights", "rgbdens")]


class TpTrainOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("rgb", "fg_weights", "bg_weights", "fg_tvals", "bg_tvals", "bg_acc",
                                               "fg_rgbsigma", "bg_rgbsigma")]


class TpLevelOut(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("rgb", "fg_rgb", "bg_rgb", "fg_acc", "bg_lambda", "depth")]


# name -> (restype, argtypes); must list every symbol the header declares
SIGNATURES = {
    "neo_abi_version": (_i, []),
    "neo_last_error": (ctypes.c_char_p, []),
    "neo_ctx_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "neo_ctx_destroy": (_i, [_vp]),
    "neo_ctx_poll_flags": (_i, [_vp, ctypes.POINTER(ctypes.c_uint32), _vp]),
    "neo_ctx_post_flags": (_i, [_vp, _vp]),
    "neo_ctx_take_flags": (_i, [_vp, _i, _vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(_i)]),
    "neo_ctx_sync_count": (_i, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "neo_ctx_stream_waits": (_i, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "neo_ctx_set_lane": (_i, [_vp, _i]),
    "neo_ctx_set_ray_grid": (_i, [_vp, _i, ctypes.c_long]),
    "neo_ctx_set_precision": (_i, [_vp, _i]),
    "neo_linspace_host": (None, [_f, _f, _i, c_float_p]),
    "neo_raygen": (_i, [_vp, _i, _i, _f, c_float_p, _vp, _vp, _vp, _vp, _vp]),
    "neo_raygen_range": (_i, [_vp, _i, _i, _f, c_float_p, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "neo_aabb_multi": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _vp, _vp, _i,
                            _vp, _vp, _vp, _vp, _vp]),
    "neo_aabb_intersect": (_i, [_vp, ctypes.POINTER(ctypes.c_double), _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "neo_intersect_sphere": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "neo_pos_enc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "neo_resample": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "neo_composite": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "neo_vanilla_upload_mlp": (_i, [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "neo_vanilla_mlp": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "neo_vanilla_render": (_i, [_vp, _vp, _vp, _vp, _i, _f, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "neo_tp_upload_mlp": (_i, [_vp, _i, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "neo_tp_set_scene": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _f, _f, _vp]),
    "neo_tp_set_preproject": (_i, [_vp, _i]),
    "neo_tp_mlp": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, c_float_p, _i, _f, _f, _f, _vp, _vp]),
    "neo_tp_render": (_i, [_vp, _vp, _vp, _vp, _i, _i, c_float_p, _i, _f, _f, _f, _i, _i, _i,
                           ctypes.POINTER(TpLevelOut), ctypes.POINTER(TpLevelOut), _vp]),
    "neo_enc_upload": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "neo_enc_floorplans": (_i, [_vp, _vp, _i, _i, _i, _f, _f, c_float_p, _f, _f, _f, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "neo_rand_uniform": (_i, [_vp, ctypes.c_uint64, ctypes.c_uint32, _i, _i, _vp, _vp]),
    "neo_tp_sample_level0": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "neo_resample_u": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "neo_composite_backward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "neo_distloss": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "neo_tp_train_points": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, c_float_p, _i, _vp, _vp, _vp]),
    "neo_tp_activate": (_i, [_vp, _vp, _vp, _vp, _f, ctypes.c_long, _vp, _vp]),
    "neo_tp_activate_backward": (_i, [_vp, _vp, _vp, _vp, _f, ctypes.c_long, _vp, _vp, _vp, _vp]),
    "neo_tp_gather_map": (_i, [_vp, _vp, ctypes.c_long, _i, _vp, ctypes.c_long, c_float_p, _i, _f, _f, _f, _vp, _vp]),
    "neo_tp_gather_map_backward": (_i, [_vp, ctypes.c_long, _i, _vp, ctypes.c_long, c_float_p, _i, _f, _f, _f, _vp, _vp, _vp]),
    "neo_transpose": (_i, [_vp, _vp, ctypes.c_long, _i, _i, _vp, _vp]),
    "neo_tp_gather_map_slice": (_i, [_vp, _vp, ctypes.c_long, ctypes.c_long, _i, _vp, ctypes.c_long, c_float_p, _i, _f, _f, _f, _vp, _vp]),
    "neo_tp_gather_map_slice_backward": (_i, [_vp, ctypes.c_long, ctypes.c_long, _i, _vp, ctypes.c_long, c_float_p, _i, _f, _f, _f, _vp, _vp, _vp]),
    "neo_pix_gather_map": (_i, [_vp, _vp, ctypes.c_long, _i, _vp, ctypes.c_long, c_float_p, _i, _f, _f, _f, _vp, _vp]),
    "neo_pix_gather_map_backward": (_i, [_vp, ctypes.c_long, _i, _vp, ctypes.c_long, c_float_p, _i, _f, _f, _f, _vp, _vp, _vp]),
    "neo_tp_mlp_train_forward_pre": (_i, [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _i, ctypes.c_long, _vp, _vp,
                                          _vp, _vp]),
    "neo_tp_mlp_train_backward_pre": (_i, [_vp, _i, ctypes.POINTER(_vp), _vp, _vp, _vp, _i, ctypes.c_long, _vp, _vp, _vp,
                                           ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, _vp, _vp]),
    "neo_train_chain_mode": (_i, [_i]),
    "neo_pix_mlp_train_tape_floats": (ctypes.c_long, [_i, ctypes.c_long]),
    "neo_pix_mlp_train_forward_pre": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, _vp, _i, ctypes.c_long, _vp, _vp, _vp, _vp]),
    "neo_pix_mlp_train_backward_pre": (_i, [_vp, ctypes.POINTER(_vp), _vp, _vp, _i, ctypes.c_long, _vp, _vp, _vp,
                                            ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, _vp]),
    "neo_mip_mlp_train_tape_floats": (ctypes.c_long, [_i, _i, _i, ctypes.c_long, _i]),
    "neo_mip_mlp_train_forward": (_i, [_vp, _i, _i, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, ctypes.c_long, _i, _vp, _vp, _vp]),
    "neo_mip_mlp_train_backward": (_i, [_vp, _i, _i, _i, ctypes.POINTER(_vp), _vp, _vp, ctypes.c_long, _i, _vp, _vp, _vp,
                                        ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "neo_linear_forward": (_i, [_vp, ctypes.c_long, _i, _i, _vp, ctypes.c_long, _vp, ctypes.c_long, _vp, _i, _i, _vp, ctypes.c_long, _vp]),
    "neo_linear_input_grad": (_i, [_vp, ctypes.c_long, _i, _i, _vp, ctypes.c_long, _vp, ctypes.c_long, _i, _vp, ctypes.c_long, _vp]),
    "neo_linear_weight_grad": (_i, [_vp, _i, _i, ctypes.c_long, _vp, ctypes.c_long, _vp, ctypes.c_long, _vp, ctypes.c_long, _vp, _vp]),
    "neo_tp_gather": (_i, [_vp, _vp, ctypes.c_long, c_float_p, _i, _f, _f, _f, _vp, _vp, _vp]),
    "neo_tp_gather_backward": (_i, [_vp, _vp, ctypes.c_long, c_float_p, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "neo_tp_mlp_train_tape_floats": (ctypes.c_long, [_i, ctypes.c_long]),
    "neo_tp_mlp_train_forward": (_i, [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _i, ctypes.c_long, _vp, _vp,
                                      _vp, _vp]),
    "neo_tp_mlp_train_backward": (_i, [_vp, _i, ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _i, ctypes.c_long, _vp, _vp, _vp,
                                       ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, _vp, _vp]),
    "neo_vanilla_mlp_train_tape_floats": (ctypes.c_long, [ctypes.c_long]),
    "neo_vanilla_mlp_train_forward": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, ctypes.c_long, _vp, _vp, _vp, _vp]),
    "neo_vanilla_mlp_train_backward": (_i, [_vp, ctypes.POINTER(_vp), _vp, _vp, ctypes.c_long, _vp, _vp, _vp, ctypes.POINTER(_vp),
                                            ctypes.POINTER(_vp), _vp, _vp, _vp]),
    "neo_tp_render_train": (_i, [_vp, _vp, _vp, _vp, _i, _i, c_float_p, _i, _f, _f, _f, _i, _i, _i, ctypes.c_uint64,
                                 ctypes.POINTER(TpTrainOut), ctypes.POINTER(TpTrainOut), _vp]),
    "neo_pix_upload_mlp": (_i, [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "neo_pix_set_scene": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _vp]),
    "neo_pix_set_preproject": (_i, [_vp, _i]),
    "neo_mip_set_layered": (_i, [_vp, _i]),
    "neo_pix_mlp": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, c_float_p, _i, _f, _f, _f, _vp, _vp]),
    "neo_pix_render": (_i, [_vp, _vp, _vp, _vp, _i, _i, c_float_p, _i, _f, _f, _f, _f, _f, _i, _i, _i,
                            _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "neo_mip_upload_mlp": (_i, [_vp, _i, _i, _i, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp]),
    "neo_mip_resample": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _f, _f, _vp, _vp, _vp]),
    "neo_mip_mlp": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "neo_mip_composite": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "neo_mip_resample_u": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _f, _f, _vp, _vp, _vp]),
    "neo_mip_encode": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "neo_mip_composite_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "neo_mip_render": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _i, _i, ctypes.POINTER(MipLevelOut), _vp]),
    "neo_ctx_set_timing": (_i, [_vp, _i]),
    "neo_ctx_read_timing": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i),
                                 ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "neo_ctx_read_spans": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_double),
                                ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i)]),
}

_lock = threading.Lock()
_lib = None


class NeoError(RuntimeError):
    pass


class NeoRangeError(NeoError):
    """The range guard of the split-fp16 arithmetic tripped (device flag bit 1): the results of that call are invalid.
    render.render_rays_test catches exactly this and re-renders the frame on the exact fp32 kernels.
    `static_operand` (flag bit 2): the operand that left the range is a packed weight or an uploaded feature map, i.e.
    every frame on the same (weights, scene) pair trips again."""
    static_operand = False


def load():
    """dlopen the in-tree library and attach prototypes.  Raises if it is missing:
    there is deliberately no fallback path."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NeoError(
                "libneo360_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(expected at %s)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("NEO360_TRAIN_CHAIN") in ("0", "1"):          # A/B of the fused training chain (neo_train_chain_mode)
            lib.neo_train_chain_mode(int(os.environ["NEO360_TRAIN_CHAIN"]))
        _lib = lib
        return lib


def check(rc):
    if rc != 0:
        msg = load().neo_last_error()
        raise NeoError("libneo360_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def linspace(start, end, steps):
    """Host helper (no GPU): the library's restatement of torch.linspace fp32."""
    buf = (ctypes.c_float * steps)()
    load().neo_linspace_host(start, end, steps, buf)
    return list(buf)


# Bumped by every library call that overwrites a caller's EXISTING tensor through its data pointer (the `out=` forms of ops.py):
# such a write does not advance the tensor's autograd version counter, and models.CallOverlap keys its fork points on versions.
write_epoch = 0


def note_external_write():
    global write_epoch
    write_epoch += 1
