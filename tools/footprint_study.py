"""Host-side measurement (no GPU): how many UNIQUE feature texels does a 64-point tile of the NeO-360
evaluator touch per source view?  (decide whether LDS footprint staging can pay.)

Runs the CPU oracle on groups of consecutive rays of the bench frame (same camera / source views as bench.py,
random features: only the geometry and the resampled positions matter), then forms the kernel's tiles
(64 consecutive points of the flattened (ray, sample) order) for all four MLP slots and counts, per tile and
view, the distinct texels among the 64 x 4 bilinear taps of the latent map (240x320) and of each tri-plane
(120x160).  Prints the distribution; writes profiles/r03_tile_footprint.json.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neo360_amd import synth            # noqa: E402
import oracle                           # noqa: E402
from oracle import gather, rays as rays_mod, sampling   # noqa: E402

H, W, NV = 480, 640, 3
GROUPS, RAYS_PER_GROUP = 24, 4


def taps(gx, gy, Wd, Hd):
    x = (gx + 1) / 2 * (Wd - 1)
    y = (gy + 1) / 2 * (Hd - 1)
    x0, y0 = np.floor(x), np.floor(y)
    ids = []
    for yy in (y0, y0 + 1):
        for xx in (x0, x0 + 1):
            ok = (xx >= 0) & (xx <= Wd - 1) & (yy >= 0) & (yy <= Hd - 1)
            ids.append(np.where(ok, yy * Wd + xx, 0).astype(np.int64))
    return np.stack(ids, -1)            # (..., 4)


def main():
    torch.manual_seed(0)
    state = synth.nerf_tp_state(0)
    scene = {k: torch.randn(NV, 128, 120, 160) * 0.1 for k in ("plane_xz", "plane_xy", "plane_yz")}
    scene["latent"] = torch.randn(NV, 512, 240, 320) * 0.1
    scene["image_wh"] = (float(W), float(H))
    poses, focal, centre = synth.source_views(NV, W, H)
    c2w = synth.look_at_origin(40.0)
    ro, vd, rd, _ = rays_mod.camera_rays(rays_mod.pixel_directions(H, W, 0.8 * W), c2w[:3, :4])
    rng = np.random.RandomState(0)
    starts = rng.randint(0, H * W - RAYS_PER_GROUP, GROUPS)
    idx = np.concatenate([np.arange(s, s + RAYS_PER_GROUP) for s in starts])
    batch = dict(rays_o=ro[idx], rays_d=rd[idx], viewdirs=vd[idx], src_poses=poses, src_focal=focal, src_c=centre)
    _, extra = oracle.neo360.render(state, batch, scene, keep=True)
    o, d = batch["rays_o"], batch["rays_d"]
    far = extra[0]["far"]
    res = {}
    for level in range(2):
        for region in ("fg", "bg"):
            tv = extra[level]["fg_t" if region == "fg" else "bg_s"]
            if region == "fg":
                pts = sampling.points_on_rays(tv, o, d)
            else:
                pts = sampling.points_on_rays(far * (1.0 - tv) + 3.0 * tv, o, d)
            B, N, _ = pts.shape
            cam = gather.world_to_camera(pts.reshape(-1, 3), poses)                 # (NV, P, 3)
            f = focal[0].repeat(2).clone()
            f[1] *= -1
            uv = gather.project(cam, f, centre[0][None])
            g = (uv * (gather.latent_scaling(240, 320) / torch.tensor([float(W), float(H)])) - 1.0).numpy()
            camn = cam.numpy()
            maps = {"latent": taps(g[..., 0], g[..., 1], 320, 240),
                    "plane_xz": taps(camn[..., 0], camn[..., 2], 160, 120),
                    "plane_xy": taps(camn[..., 0], camn[..., 1], 160, 120),
                    "plane_yz": taps(camn[..., 1], camn[..., 2], 160, 120)}
            key = "%s_%s" % (region, "coarse" if level == 0 else "fine")
            res[key] = {}
            # tile-views with NO weighted tap at all: latent / all three planes / everything (texel id 0 is the placeholder
            # of invalid taps; a valid tap at texel 0 is possible but rare enough not to matter for this statistic)
            none_lat, none_pl, none_all, nt = 0, 0, 0, 0
            for gi in range(GROUPS):
                lo, hi = gi * RAYS_PER_GROUP * N, (gi + 1) * RAYS_PER_GROUP * N
                for t0 in range(lo, hi - 63, 64):
                    for v in range(NV):
                        a = not maps["latent"][v, t0:t0 + 64].any()
                        b = not (maps["plane_xz"][v, t0:t0 + 64].any() or maps["plane_xy"][v, t0:t0 + 64].any() or maps["plane_yz"][v, t0:t0 + 64].any())
                        none_lat += a; none_pl += b; none_all += (a and b); nt += 1
            res[key]["tile_views_without_weighted_taps"] = dict(latent=none_lat / nt, all_planes=none_pl / nt, everything=none_all / nt)
            print(key, "tile-views without any weighted tap: latent %.3f, all three planes %.3f, both %.3f" % (none_lat / nt, none_pl / nt, none_all / nt))
            for name, t in maps.items():
                counts = []
                for gi in range(GROUPS):                                   # tiles inside a group of consecutive rays
                    lo, hi = gi * RAYS_PER_GROUP * N, (gi + 1) * RAYS_PER_GROUP * N
                    for t0 in range(lo, hi - 63, 64):
                        for v in range(NV):
                            counts.append(len(np.unique(t[v, t0:t0 + 64])))
                c = np.array(counts)
                res[key][name] = dict(mean=float(c.mean()), p50=float(np.percentile(c, 50)), p90=float(np.percentile(c, 90)),
                                      p99=float(np.percentile(c, 99)), max=int(c.max()),
                                      frac_le_32=float((c <= 32).mean()), frac_le_64=float((c <= 64).mean()),
                                      frac_le_96=float((c <= 96).mean()), frac_le_128=float((c <= 128).mean()), tiles=len(c))
                print(key, name, json.dumps(res[key][name]))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r03_tile_footprint.json"), "w") as fh:
        json.dump(dict(note="unique texels among the 64 x 4 bilinear taps of one 64-point tile and source view; bench camera "
                            "(look_at_origin(40 deg)), 3 source views, 128 + 256 samples, %d groups of %d consecutive rays"
                            % (GROUPS, RAYS_PER_GROUP), result=res), fh, indent=1)


if __name__ == "__main__":
    main()
