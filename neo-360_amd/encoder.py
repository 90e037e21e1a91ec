"""Scene encoder with the pillar stage on the HIP library.

`GridEncoder` mirrors models/neo360/encoder_tp_fusion_conv.py:282-597: same constructor role, `forward(images, poses,
focal, c)` -> (scene_grid_xz, scene_grid_xy, scene_grid_yz), same state_dict keys (`depth_fc.*`,
`pillar_aggregator_{xz,yz,xy}.*`, `floorplan_convnet_{xy,yz,xz}.*`, `spatial_encoder.*`).  What runs where:

  spatial_encoder        the caller's image CNN (the reference's ResNet-34 SpatialEncoder): PyTorch, out of scope
  pillar stage           world grid -> per cell-view [latent | camera xyz | direction] -> depth_fc -> three axis scorers ->
                         softmax-weighted sums = three floor-plans: ONE library call (csrc/pillar.hip), 1.58 MMAC per
                         cell-view x 786,432 cell-views, the part the reference pays 300 x per frame
  floorplan_convnet_*    the reference's 2-D conv stacks on 64 x 64 floor-plans: PyTorch (MIOpen), once per scene

Attach it to `models.NeRF_TP(encoder=GridEncoder(spatial_encoder=...))`: the module runs it once per distinct src_imgs.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .context import f32, ptr
from .models import _HipModule, _fingerprint, _ptr_table


def _kaiming(m):
    """init_weights_kaiming (encoder_tp_fusion_conv.py:255-260): applies to nn.Linear only."""
    if type(m) == nn.Linear:
        nn.init.kaiming_normal_(m.weight)
        nn.init.uniform_(m.bias, -1e-3, 1e-3)


class DepthPillarEncoder(nn.Module):
    """encoder_tp_fusion_conv.py:263-279 (parameter container; evaluated by the library)."""

    def __init__(self, inp_ch, LS):
        super().__init__()
        self.common_branch = nn.Sequential(nn.Linear(inp_ch, LS), nn.ReLU(inplace=True), nn.Linear(LS, LS), nn.ReLU(inplace=True))
        self.depth_encoder = nn.Linear(LS, LS)
        self.common_branch.apply(_kaiming)
        self.depth_encoder.apply(_kaiming)


def _floorplan_convnet():
    """encoder_tp_fusion_conv.py:375-397 (identical for xy / yz / xz)."""
    return nn.Sequential(
        nn.Conv2d(512, 256, 3, stride=2, padding=1), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
        nn.Conv2d(256, 128, 3, stride=2, padding=1), nn.BatchNorm2d(128), nn.ReLU(inplace=True),
        nn.Conv2d(128, 128, 3, stride=1, padding=1), nn.BatchNorm2d(128), nn.ReLU(inplace=True),
        nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
        nn.Conv2d(128, 128, 3, padding=1), nn.BatchNorm2d(128), nn.ReLU(inplace=True),
        nn.Upsample(size=(120, 160), mode="bilinear", align_corners=True),
        nn.Conv2d(128, 128, 3, padding=1))


class GridEncoder(_HipModule):
    LATENT = 512

    def __init__(self, spatial_encoder=None, grid_size=(64, 64, 64), encoder_type="resnet"):
        super().__init__()
        self.grid_size = [int(g) for g in grid_size]
        self.encoder_type = encoder_type
        self.side_lengths = [1, 1, 1]
        if spatial_encoder is not None:
            self.spatial_encoder = spatial_encoder
        self.latent_size = LS = self.LATENT
        self.depth_fc = DepthPillarEncoder(LS + 3 + 3, LS)
        mk = lambda: nn.Sequential(nn.Linear(LS + 1, LS), nn.ReLU(inplace=True), nn.Linear(LS, 1))
        self.pillar_aggregator_xz, self.pillar_aggregator_yz, self.pillar_aggregator_xy = mk(), mk(), mk()
        self.floorplan_convnet_xy, self.floorplan_convnet_yz, self.floorplan_convnet_xz = (_floorplan_convnet() for _ in range(3))
        for m in (self.pillar_aggregator_xz, self.pillar_aggregator_yz, self.pillar_aggregator_xy, self.floorplan_convnet_xy,
                  self.floorplan_convnet_yz, self.floorplan_convnet_xz):
            m.apply(_kaiming)

    def _context(self, device):
        ctx = super()._context(device)
        if getattr(ctx, "_precision", None) != "f16x3":
            raise _lib.NeoError("the pillar stage exists in the split-fp16 arithmetic only (precision 'f16x3')")
        return ctx

    def ordered_layers(self):
        """Upload order fixed by include/neo360_hip.h (neo_enc_upload)."""
        agg = []
        for m in (self.pillar_aggregator_xz, self.pillar_aggregator_yz, self.pillar_aggregator_xy):
            agg += [m[0], m[2]]
        return [self.depth_fc.common_branch[0], self.depth_fc.common_branch[2], self.depth_fc.depth_encoder] + agg

    def _sync_weights(self, ctx):
        layers = self.ordered_layers()
        ws = [f32(l.weight.detach(), "weight") for l in layers]
        bs = [f32(l.bias.detach(), "bias") for l in layers]
        fp = _fingerprint(ws + bs)
        if ctx.uploaded.get("enc") == fp:
            return
        _lib.check(ctx.lib.neo_enc_upload(ctx.handle, _ptr_table(ws), _ptr_table(bs), ctx.stream()))
        ctx.uploaded["enc"] = fp

    @torch.no_grad()
    def floorplans(self, latent, poses, focal, c, image_wh):
        """The pillar stage alone: latent (NV,512,Hf,Wf), poses (NV,4,4) c2w, focal (NV,), c (NV,2) (view 0's are used
        for every view, encoder_tp_fusion_conv.py:491-493), image_wh = (W,H) of the encoded images.  Returns the
        channels-last floor-plans (yz (NV,G1,G2,512), xz (NV,G0,G2,512), xy (NV,G0,G1,512))."""
        latent = f32(latent, "latent")
        dev = latent.device
        ctx = self._context(dev)
        self._sync_weights(ctx)
        NV, C, Hf, Wf = latent.shape
        if C != self.LATENT:
            raise _lib.NeoError("the pillar stage is specialised for the reference's 512-channel latent")
        G0, G1, G2 = self.grid_size
        host = poses.detach().float().cpu().contiguous()
        host_poses = (ctypes.c_float * (16 * NV))(*host.reshape(-1).tolist())
        f0 = float(focal[0])
        cx, cy = (float(x) for x in c[0])
        yz = torch.empty(NV, G1, G2, 512, device=dev)
        xz = torch.empty(NV, G0, G2, 512, device=dev)
        xy = torch.empty(NV, G0, G1, 512, device=dev)
        _lib.check(ctx.lib.neo_enc_floorplans(ctx.handle, ptr(latent), NV, Hf, Wf, float(image_wh[0]), float(image_wh[1]),
                                              host_poses, f0, cx, cy, G0, G1, G2, ptr(yz), ptr(xz), ptr(xy), ctx.stream()))
        if self.poll_flags:
            self._raise_flags(ctx.poll_flags())
        return yz, xz, xy

    def forward(self, images, poses, focal, c):
        """GridEncoder.forward (encoder_tp_fusion_conv.py:472-597): images (NV,3,H,W) -> three (NV,128,120,160) planes
        in the reference's return order (xz, xy, yz).  Leaves the pixel-aligned latent in `spatial_encoder.latent`."""
        NV, _, H, W = images.shape
        self.spatial_encoder(images)
        yz, xz, xy = self.floorplans(self.spatial_encoder.latent, poses, focal, c, (W, H))
        nchw = lambda t: t.permute(0, 3, 1, 2)
        return self.floorplan_convnet_xz(nchw(xz)), self.floorplan_convnet_xy(nchw(xy)), self.floorplan_convnet_yz(nchw(yz))
