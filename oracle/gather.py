"""Oracle: camera transforms, projection and bilinear feature lookups.

Test infrastructure (oracle/__init__.py).  `bilinear_zero_pad` restates the
arithmetic of torch.nn.functional.grid_sample(mode="bilinear",
padding_mode="zeros", align_corners=True) — the third-party op the reference
calls at encoder_tp_fusion_conv.py:180-202 and encoder_pn.py:144-150 — with
explicit index math; tests pin it against torch's own grid_sample.
"""
import torch


def world_to_camera(pts, c2w):
    """pts (P,3) world, c2w (NV,4,4) -> (NV,P,3): R^T x - R^T t per view.
    Follows neo360/util.py:52-70 (rot = c2w[:, :3,:3]^T, trans = -rot @ c2w[:, :3,3])."""
    rot = c2w[:, :3, :3].transpose(1, 2)
    trans = -torch.bmm(rot, c2w[:, :3, 3:])
    return torch.matmul(rot[:, None], pts[None, :, :, None])[..., 0] + trans[:, None, :, 0]


def world_to_camera_dirs(dirs, c2w):
    """Directions rotate only.  neo360/util.py:45-49."""
    rot = c2w[:, :3, :3].transpose(1, 2)
    return torch.matmul(rot[:, None], dirs[None, :, :, None])[..., 0]


def project(cam, focal_xy, centre):
    """uv = -xy/(z+1e-9) * (f, -f) + c.  neo360/util.py:92-111 with the
    (f,-f) sign flip applied by the caller at neo360/model.py:242-243."""
    uv = -cam[..., :2] / (cam[..., 2:] + 1e-9)
    return uv * focal_xy + centre


def bilinear_zero_pad(maps, grid):
    """maps (NV,C,H,W), grid (NV,P,2) in [-1,1] (x->W, y->H) -> (NV,P,C).
    align_corners=True: pixel = (g+1)/2*(size-1); four taps; a tap outside the
    map contributes zero."""
    NV, C, H, W = maps.shape
    x = (grid[..., 0] + 1) / 2 * (W - 1)
    y = (grid[..., 1] + 1) / 2 * (H - 1)
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    x1 = x0 + 1
    y1 = y0 + 1
    out = torch.zeros(NV, grid.shape[1], C)
    flat = maps.permute(0, 2, 3, 1).reshape(NV, H * W, C)
    # tap weights in torch's own form: (x1-x)(y1-y), (x-x0)(y1-y), (x1-x)(y-y0), (x-x0)(y-y0)
    for yi, wy in ((y0, y1 - y), (y1, y - y0)):
        for xi, wx in ((x0, x1 - x), (x1, x - x0)):
            ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long()
            tap = torch.gather(flat, 1, idx[..., None].expand(-1, -1, C))
            out = out + tap * (wx * wy * ok)[..., None]
    return out


def triplane_features(pts, plane_xz, plane_xy, plane_yz, c2w):
    """Sum of three plane lookups at each view's camera-frame point, view-major
    (NV*P, C).  Follows encoder_tp_fusion_conv.py:122-209: the camera-frame
    coordinates are used directly as normalised grid coordinates; (x,z)->xz,
    (x,y)->xy, (y,z)->yz, first component indexes width."""
    cam = world_to_camera(pts.reshape(-1, 3), c2w)
    total = (
        bilinear_zero_pad(plane_xz, cam[..., [0, 2]].float())
        + bilinear_zero_pad(plane_xy, cam[..., [0, 1]].float())
        + bilinear_zero_pad(plane_yz, cam[..., [1, 2]].float())
    )
    return total.reshape(-1, total.shape[-1])


def latent_scaling(Hf, Wf):
    """[Wf,Hf]/([Wf,Hf]-1)*2.  encoder_pn.py:204-206."""
    s = torch.tensor([float(Wf), float(Hf)])
    return s / (s - 1) * 2.0


def pixel_aligned_features(pts, latent, c2w, focal, centre, image_wh, flip_y=True):
    """Project into every source view and bilinearly read the feature map,
    view-major (NV*P, C).  Follows neo360/model.py:239-264 +
    encoder_pn.py:101-152: focal/centre of source view 0 for all views;
    g = uv * latent_scaling/image_size - 1.  flip_y=False is the PixelNeRF baseline's
    call (vanilla_nerf/model_pixel.py:203-206), which passes (f, f) without the sign flip."""
    cam = world_to_camera(pts.reshape(-1, 3), c2w)
    f = focal[0].unsqueeze(-1).repeat((1, 2)).clone()
    if flip_y:
        f[..., 1] *= -1.0
    uv = project(cam, f, centre[0].unsqueeze(0))
    scale = latent_scaling(latent.shape[-2], latent.shape[-1]) / torch.as_tensor(image_wh, dtype=torch.float32)
    feats = bilinear_zero_pad(latent, uv * scale - 1.0)
    return feats.reshape(-1, feats.shape[-1])
