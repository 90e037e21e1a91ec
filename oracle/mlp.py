"""Oracle: the two per-point MLPs, evaluated from a flat state_dict.

Test infrastructure (oracle/__init__.py).  `params` is a dict of fp32 tensors
with the reference's state_dict key names (torch.nn.Linear layout: weight
(out,in), bias (out,)); `prefix` selects the sub-module, e.g. "coarse_mlp.".
"""
import torch
import torch.nn.functional as F


def _lin(params, name, x):
    return F.linear(x, params[name + ".weight"], params[name + ".bias"])


def vanilla_mlp(params, prefix, x_enc, dir_enc, depth=8, skip=4):
    """x_enc (B,N,63), dir_enc (B,27) -> raw_rgb (B,N,3), raw_sigma (B,N,1).

    Follows vanilla_nerf/model.py:100-125: 8 ReLU layers of 256 with the
    encoded input re-concatenated after layer index 4; density head on the
    trunk; bottleneck (no activation) + per-ray view encoding tiled over the
    samples -> 128 ReLU -> rgb.
    """
    B, N, Fin = x_enc.shape
    x0 = x_enc.reshape(-1, Fin)
    h = x0
    for i in range(depth):
        h = torch.relu(_lin(params, "%spts_linears.%d" % (prefix, i), h))
        if i % skip == 0 and i > 0:
            h = torch.cat([h, x0], dim=-1)
    raw_sigma = _lin(params, prefix + "density_layer", h).reshape(-1, N, 1)
    bott = _lin(params, prefix + "bottleneck_layer", h)
    cond = torch.tile(dir_enc[:, None, :], (1, N, 1)).reshape(-1, dir_enc.shape[-1])
    v = torch.relu(_lin(params, prefix + "views_linear.0", torch.cat([bott, cond], dim=-1)))
    return _lin(params, prefix + "rgb_layer", v).reshape(-1, N, 3), raw_sigma


def _view_mean(x, nv, npts):
    """neo360/util.py:599-610 combine_interleaved(..., 'average'): rows are
    view-major (v*P + p); mean over v."""
    return x.reshape(-1, nv, npts, x.shape[-1]).mean(dim=1).reshape(-1, x.shape[-1])


def nerfpp_mlp(params, prefix, x_enc, cond_rows, world_feat, local_feat, nv, depth=4, skip=2, combine=3):
    """Late-fusion multi-view MLP (NeRFPPMLP).

    x_enc (NV,P,63|84) encoded camera-frame points; cond_rows (NV*P,27);
    world_feat (NV*P,128); local_feat (NV*P,512).  Returns raw_rgb (P,3),
    raw_sigma (P,1) (caller reshapes to rays x samples).

    Follows neo360/model.py:110-158: input = [enc | local | world]; 4 ReLU
    layers of 128; at layer index 3 the per-view bottleneck is taken and the
    trunk is averaged over views; the input is re-concatenated after layer
    index 2; density from the view-mean trunk; view branch
    [bottleneck | cond] -> 64 -> mean over views -> ReLU -> 64 ReLU -> rgb.
    """
    npts = x_enc.shape[1]
    x0 = torch.cat([x_enc.reshape(-1, x_enc.shape[-1]), local_feat, world_feat], dim=-1)
    h = x0
    bott = None
    for i in range(depth):
        h = torch.relu(_lin(params, "%spts_linears.%d" % (prefix, i), h))
        if i == combine:
            bott = _lin(params, prefix + "bottleneck_layer", h)
            h = _view_mean(h, nv, npts)
        if i % skip == 0 and i > 0:
            h = torch.cat([h, x0], dim=-1)
    raw_sigma = _lin(params, prefix + "density_layer", h)
    y = _lin(params, prefix + "views_linear.0", torch.cat([bott, cond_rows], dim=-1))
    y = torch.relu(_view_mean(y, nv, npts))
    y = torch.relu(_lin(params, prefix + "views_linear.1", y))
    return _lin(params, prefix + "rgb_layer", y), raw_sigma


def pixelnerf_mlp(params, prefix, x_enc, cond_rows, local_feat, nv):
    """PixelNeRF's late-fusion MLP (vanilla_nerf/model_pixel.py:96-131): input = [enc 63 | latent 512];
    4 ReLU layers of 128 (skip_layer=4 never fires); after the last one the per-view bottleneck is
    taken and the trunk averaged over views; density from the mean; view branch
    [bottleneck | cond 27] -> 128 -> mean over views -> ReLU -> 128 ReLU -> rgb.
    x_enc (NV,P,63), cond_rows (NV*P,27), local_feat (NV*P,512) -> raw_rgb (P,3), raw_sigma (P,1)."""
    npts = x_enc.shape[1]
    h = torch.cat([x_enc.reshape(-1, x_enc.shape[-1]), local_feat], dim=-1)
    bott = None
    for i in range(4):
        h = torch.relu(_lin(params, "%spts_linears.%d" % (prefix, i), h))
        if i == 3:
            bott = _lin(params, prefix + "bottleneck_layer", h)
            h = _view_mean(h, nv, npts)
    raw_sigma = _lin(params, prefix + "density_layer", h)
    y = _lin(params, prefix + "views_linear.0", torch.cat([bott, cond_rows], dim=-1))
    y = torch.relu(_view_mean(y, nv, npts))
    y = torch.relu(_lin(params, prefix + "views_linear.1", y))
    return _lin(params, prefix + "rgb_layer", y), raw_sigma


def density_activation(raw):
    """softplus(raw - 1).  vanilla_nerf/model.py:203-204, neo360/model.py:380-381."""
    return F.softplus(raw + (-1.0))


def colour_activation(raw):
    """sigmoid(raw)*(1+2*0.001) - 0.001.  vanilla_nerf/model.py:198-200,
    neo360/model.py:383-385."""
    pad = 0.001
    return torch.sigmoid(raw) * (1 + 2 * pad) - pad
