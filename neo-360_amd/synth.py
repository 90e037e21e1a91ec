"""Seeded synthetic inputs owned by the build.

Everything is derived from a counter-based integer hash (splitmix64), never
from a torch / numpy RNG stream, so the golden-fixture generator, the tests
and bench.py regenerate bit-identical weights, feature planes and cameras on
any machine and any library version.  Fixtures therefore only need to store
expected OUTPUTS.
"""
import math

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream_id(tag):
    """Stable 64-bit id of a string tag (FNV-1a)."""
    h = 0xCBF29CE484222325
    for ch in tag.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _uniform01_range(seed, tag, lo, hi):
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) ^ np.uint64(_stream_id(tag)))
        idx = np.arange(lo, hi, dtype=np.uint64)
        bits = _splitmix64(base + idx * np.uint64(0xD1342543DE82EF95))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _in_pieces(fn, n, piece=1 << 22):
    """fn(lo, hi) over [0, n) in pieces on a few threads (numpy releases the GIL): values depend on the index only,
    so the result is identical to one call - full-size feature maps (118 M values) take seconds instead of half a minute."""
    if n <= piece:
        return fn(0, n)
    from concurrent.futures import ThreadPoolExecutor
    import os
    bounds = [(lo, min(lo + piece, n)) for lo in range(0, n, piece)]
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        parts = list(ex.map(lambda b: fn(*b), bounds))
    return np.concatenate(parts)


def uniform01(seed, tag, n):
    """n float64 values in [0,1), a pure function of (seed, tag, index)."""
    return _in_pieces(lambda lo, hi: _uniform01_range(seed, tag, lo, hi), n)


def uniform(seed, tag, shape, lo, hi):
    n = int(np.prod(shape))
    v = lo + (hi - lo) * uniform01(seed, tag, n)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def normal(seed, tag, shape, std=1.0):
    n = int(np.prod(shape))
    m = (n + 1) // 2

    def pairs(lo, hi):                                   # Box-Muller on index pairs: (cos half, sin half)
        u1 = _uniform01_range(seed, tag + "/a", lo, hi)
        u2 = _uniform01_range(seed, tag + "/b", lo, hi)
        r = np.sqrt(-2.0 * np.log(1.0 - u1))
        return np.stack([(std * (r * np.cos(2 * math.pi * u2))).astype(np.float32),
                         (std * (r * np.sin(2 * math.pi * u2))).astype(np.float32)])

    if m <= (1 << 22):
        both = pairs(0, m)
    else:
        from concurrent.futures import ThreadPoolExecutor
        import os
        piece = 1 << 22
        bounds = [(lo, min(lo + piece, m)) for lo in range(0, m, piece)]
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            both = np.concatenate(list(ex.map(lambda b: pairs(*b), bounds)), axis=1)
    v = np.concatenate([both[0], both[1]])[:n]
    return torch.from_numpy(v.reshape(shape))


# ----------------------------------------------------------------------------
# weights: same distributions as the reference's initialisers
# ----------------------------------------------------------------------------

def _linear(seed, sd, name, fan_out, fan_in, xavier=True, gain=1.0):
    """xavier_uniform_ weight (bound sqrt(6/(in+out))) or torch's default
    Linear init (bound 1/sqrt(in)); bias always the default U(+-1/sqrt(in))."""
    wb = math.sqrt(6.0 / (fan_in + fan_out)) if xavier else 1.0 / math.sqrt(fan_in)
    sd[name + ".weight"] = uniform(seed, name + ".weight", (fan_out, fan_in), -wb, wb) * gain
    bb = 1.0 / math.sqrt(fan_in)
    sd[name + ".bias"] = uniform(seed, name + ".bias", (fan_out,), -bb, bb)


def vanilla_mlp_state(seed, prefix, sd=None, density_gain=1.0):
    """Parameters of one vanilla NeRFMLP under reference key names
    (vanilla_nerf/model.py:44-98): 8x256 trunk, skip at 4, 63/27-d encodings."""
    sd = {} if sd is None else sd
    width, pos, view = 256, 63, 27
    _linear(seed, sd, prefix + "pts_linears.0", width, pos)
    for i in range(1, 8):
        _linear(seed, sd, prefix + "pts_linears.%d" % i, width, width + pos if i == 5 else width)
    _linear(seed, sd, prefix + "views_linear.0", 128, width + view, xavier=False)
    _linear(seed, sd, prefix + "bottleneck_layer", width, width)
    _linear(seed, sd, prefix + "density_layer", 1, width, gain=density_gain)
    _linear(seed, sd, prefix + "rgb_layer", 3, 128)
    return sd


def vanilla_state(seed=0, density_gain=1.0):
    sd = {}
    vanilla_mlp_state(seed, "coarse_mlp.", sd, density_gain)
    vanilla_mlp_state(seed, "fine_mlp.", sd, density_gain)
    return sd


def nerfpp_mlp_state(seed, prefix, sd=None, input_ch=3, local=512, world=128, density_gain=1.0):
    """Parameters of one NeRFPPMLP (neo360/model.py:37-108): 4x128 trunk,
    skip at 2, input = posenc(10 deg) + local + world."""
    sd = {} if sd is None else sd
    width, cond = 128, 64
    pos = (2 * 10 + 1) * input_ch + local + world
    _linear(seed, sd, prefix + "pts_linears.0", width, pos)
    for i in range(1, 4):
        _linear(seed, sd, prefix + "pts_linears.%d" % i, width, width + pos if i == 3 else width)
    _linear(seed, sd, prefix + "views_linear.0", cond, width + 27, xavier=False)
    _linear(seed, sd, prefix + "views_linear.1", cond, cond)
    _linear(seed, sd, prefix + "bottleneck_layer", width, width)
    _linear(seed, sd, prefix + "density_layer", 1, width, gain=density_gain)
    _linear(seed, sd, prefix + "rgb_layer", 3, cond)
    return sd


def pixelnerf_mlp_state(seed, prefix, sd=None, latent=512, density_gain=1.0):
    """Parameters of one PixelNeRF NeRFMLP (vanilla_nerf/model_pixel.py:35-94): 4x128 trunk on
    [posenc 63 | latent 512] (the skip never fires at depth 4), bottleneck, 2x128 view branch."""
    sd = {} if sd is None else sd
    width = 128
    _linear(seed, sd, prefix + "pts_linears.0", width, 63 + latent)
    for i in range(1, 4):
        _linear(seed, sd, prefix + "pts_linears.%d" % i, width, width)
    _linear(seed, sd, prefix + "views_linear.0", width, width + 27, xavier=False)
    _linear(seed, sd, prefix + "views_linear.1", width, width)
    _linear(seed, sd, prefix + "bottleneck_layer", width, width)
    _linear(seed, sd, prefix + "density_layer", 1, width, gain=density_gain)
    _linear(seed, sd, prefix + "rgb_layer", 3, width)
    return sd


def pixelnerf_state(seed=0, density_gain=1.0):
    sd = {}
    for name in ("coarse_mlp.", "fine_mlp."):
        pixelnerf_mlp_state(seed, name, sd, density_gain=density_gain)
    return sd


def nerf_tp_state(seed=0, density_gain=1.0):
    sd = {}
    for name, ch in (("fg_coarse_mlp.", 3), ("fg_fine_mlp.", 3), ("bg_coarse_mlp.", 4), ("bg_fine_mlp.", 4)):
        nerfpp_mlp_state(seed, name, sd, input_ch=ch, density_gain=density_gain)
    return sd


def _kaiming_linear(seed, sd, name, fan_out, fan_in, kaiming=True, gain=1.0):
    """kaiming_uniform_ (a=0: bound sqrt(6/in)) or torch's default Linear init (1/sqrt(in))."""
    wb = math.sqrt(6.0 / fan_in) if kaiming else 1.0 / math.sqrt(fan_in)
    sd[name + ".weight"] = uniform(seed, name + ".weight", (fan_out, fan_in), -wb, wb) * gain
    bb = 1.0 / math.sqrt(fan_in)
    sd[name + ".bias"] = uniform(seed, name + ".bias", (fan_out,), -bb, bb)


def mip360_state(seed=0, density_gain=1.0, weight_gain=1.0):
    """Parameters of MipNeRF360 under reference key names (mipnerf360/model.py:30-233):
    mlps.0/.1 = PropMLP (4x256, no rgb branch), mlps.2 = NeRFMLP (8x1024, skip at 4).
    `weight_gain` < 1 tames the 1024-wide kaiming init so random-init densities stay O(1)."""
    from .geopoly import icosahedron_basis
    sd = {}
    pos = 504
    for lvl, (depth, width, rgb) in enumerate(((4, 256, False), (4, 256, False), (8, 1024, True))):
        pre = "mlps.%d." % lvl
        sd[pre + "pos_basis_t"] = icosahedron_basis()
        _kaiming_linear(seed, sd, pre + "pts_linear.0", width, pos, gain=weight_gain)
        for i in range(1, depth):
            _kaiming_linear(seed, sd, pre + "pts_linear.%d" % i, width, width + pos if i == 5 else width, gain=weight_gain)
        _kaiming_linear(seed, sd, pre + "density_layer", 1, width, gain=density_gain * weight_gain)
        if rgb:
            _kaiming_linear(seed, sd, pre + "bottleneck_layer", 256, width, gain=weight_gain)
            _kaiming_linear(seed, sd, pre + "views_linear.0", 128, 256 + 27, gain=weight_gain)
            _kaiming_linear(seed, sd, pre + "rgb_layer", 3, 128, gain=weight_gain)
    return sd


# ----------------------------------------------------------------------------
# cameras / rays / scene features
# ----------------------------------------------------------------------------

def look_at_origin(azimuth_deg, radius=0.6, height=0.3):
    """c2w (4,4) fp32 of a camera on an orbit, looking at the origin, z up;
    camera convention of the reference's rays: x right, y up, looking along -z."""
    a = math.radians(azimuth_deg)
    eye = np.array([radius * math.cos(a), radius * math.sin(a), height])
    back = eye / np.linalg.norm(eye)
    right = np.cross(np.array([0.0, 0.0, 1.0]), back)
    right /= np.linalg.norm(right)
    up = np.cross(back, right)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, back, eye
    return torch.from_numpy(m.astype(np.float32))


def source_views(nv=3, W=640, H=480):
    az = [0.0, 115.0, 229.0, 300.0, 60.0][:nv]
    poses = torch.stack([look_at_origin(a) for a in az])
    focal = torch.full((nv,), 0.8 * W)
    centre = torch.tensor([[W / 2.0, H / 2.0]]).repeat(nv, 1)
    return poses, focal, centre


def scene_features(seed=0, nv=3, world_ch=128, plane_hw=(120, 160), local_ch=512, latent_hw=(240, 320), std=0.1):
    """Stand-ins for the scene encoder's outputs, reference layout (NCHW)."""
    planes = {k: normal(seed, "plane_" + k, (nv, world_ch) + tuple(plane_hw), std) for k in ("xz", "xy", "yz")}
    latent = normal(seed, "latent", (nv, local_ch) + tuple(latent_hw), std)
    return dict(plane_xz=planes["xz"], plane_xy=planes["xy"], plane_yz=planes["yz"], latent=latent)


def pillar_state(seed=0, latent=512):
    """Parameters of the pillar stage of GridEncoder (neo360/encoder_tp_fusion_conv.py:263-279, :364-373) under the
    reference's key names: depth_fc (518 -> 512 -> 512 -> 512) and the three axis scorers (513 -> 512 -> 1).
    kaiming_normal_ weights (std sqrt(2/fan_in)), biases U(-1e-3, 1e-3) (init_weights_kaiming, :255-260)."""
    sd = {}

    def lin(name, fan_out, fan_in):
        sd[name + ".weight"] = normal(seed, name + ".weight", (fan_out, fan_in), math.sqrt(2.0 / fan_in))
        sd[name + ".bias"] = uniform(seed, name + ".bias", (fan_out,), -1e-3, 1e-3)

    lin("depth_fc.common_branch.0", latent, latent + 6)
    lin("depth_fc.common_branch.2", latent, latent)
    lin("depth_fc.depth_encoder", latent, latent)
    for ax in ("xz", "yz", "xy"):
        lin("pillar_aggregator_%s.0" % ax, latent, latent + 1)
        lin("pillar_aggregator_%s.2" % ax, 1, latent)
    return sd
