"""Oracle: the pillar stage of the scene encoder: GridEncoder.forward from the world grid to the
three floor-plans (models/neo360/encoder_tp_fusion_conv.py:472-578), i.e. everything between the ResNet latent and the
floor-plan conv nets.  Test infrastructure (oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

from . import gather


def world_grid(grid_size, side=((-1.0, 1.0), (-1.0, 1.0), (0.0, 1.0))):
    """neo360/util.py:12-26 get_world_grid: (G0*G1*G2, 3), x slowest."""
    axes = [torch.linspace(side[a][0], side[a][1], grid_size[a]) for a in range(3)]
    X, Y, Z = torch.meshgrid(*axes, indexing="ij")
    return torch.stack([X, Y, Z], dim=-1).reshape(-1, 3)


def floorplans(params, latent, image_wh, poses, focal, centre, grid_size=(64, 64, 64)):
    """latent (NV,512,Hf,Wf) = SpatialEncoder output; poses (NV,4,4) c2w; focal (NV,), centre (NV,2) (view 0's are
    used for every view, encoder_tp_fusion_conv.py:491-493).  Returns floorplans_yz (NV,G1,G2,512),
    floorplans_xz (NV,G0,G2,512), floorplans_xy (NV,G0,G1,512) - the tensors the floor-plan conv nets receive
    (before their permute to NCHW, :580-592)."""
    nv = poses.shape[0]
    G0, G1, G2 = grid_size
    wg = world_grid(grid_size).to(latent.dtype)                                  # (NC,3)
    cam = gather.world_to_camera(wg, poses)                                      # (NV,NC,3)   :507
    mask = cam[:, :, 2] < 1e-3                                                   # :509
    dirs = wg[None] - poses[:, None, :3, -1]                                     # :511
    dirs = dirs / torch.norm(dirs + 1e-9, dim=-1)[:, :, None]                    # :512-516
    dirs = dirs * mask[:, :, None]                                               # :517
    # projection with (f, -f) and view 0's intrinsics, then SpatialEncoder.index (encoder_pn.py:101-152)
    f0, c0 = focal[0], centre[0]
    uv = -cam[..., :2] / (cam[..., 2:] + 1e-9)
    uv = uv * torch.stack([f0, -f0]) + c0
    Hf, Wf = latent.shape[-2:]
    scale = gather.latent_scaling(Hf, Wf).to(latent.dtype) / torch.tensor([float(image_wh[0]), float(image_wh[1])], dtype=latent.dtype)
    grid = (uv * scale - 1.0).unsqueeze(2)                                       # (NV,NC,1,2)
    feat = F.grid_sample(latent, grid, align_corners=True, mode="bilinear", padding_mode="zeros")[:, :, :, 0]   # (NV,512,NC)
    x = torch.cat([feat, cam.permute(0, 2, 1), dirs.permute(0, 2, 1)], dim=1).permute(0, 2, 1)                 # (NV,NC,518)
    lin = lambda name, t: F.linear(t, params[name + ".weight"], params[name + ".bias"])
    h = torch.relu(lin("depth_fc.common_branch.0", x))
    h = torch.relu(lin("depth_fc.common_branch.2", h))
    L = lin("depth_fc.depth_encoder", h).reshape(nv, G0, G1, G2, -1)             # :536-541
    w3 = wg.reshape(1, G0, G1, G2, 3).expand(nv, -1, -1, -1, -1)

    def scores(ax, coord):
        t = torch.cat([L, w3[..., coord:coord + 1]], dim=-1)
        return lin("pillar_aggregator_%s.2" % ax, torch.relu(lin("pillar_aggregator_%s.0" % ax, t)))

    w_yz = torch.softmax(scores("yz", 0), dim=1)                                 # over x   :562-574
    w_xz = torch.softmax(scores("xz", 1), dim=2)                                 # over y
    w_xy = torch.softmax(scores("xy", 2), dim=3)                                 # over z
    return (L * w_yz).sum(1), (L * w_xz).sum(2), (L * w_xy).sum(3)
