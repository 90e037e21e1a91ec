"""Host-side (no GPU): how full are the entries of k_tp_mlp_hpp's work list?  An entry = (16-row group, map) of a 64-point
tile-view in which ANY row has a valid bilinear tap; the kernel blends all 16 rows of it.  Counts, per MLP slot of the bench
geometry: entries per tile-view (of 16), rows with a valid tap per entry (of 16), and what a list of single (row, map) pairs
packed 16 to a step would need instead.  Same rays / scene set-up as tools/footprint_study.py."""
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


def valid(gx, gy, Wd, Hd):
    x = (gx + 1) / 2 * (Wd - 1)
    y = (gy + 1) / 2 * (Hd - 1)
    x0, y0 = np.floor(x), np.floor(y)
    any_ok = np.zeros(x.shape, bool)
    for yy in (y0, y0 + 1):
        for xx in (x0, x0 + 1):
            any_ok |= (xx >= 0) & (xx <= Wd - 1) & (yy >= 0) & (yy <= Hd - 1)
    return any_ok


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
    for level in range(2):
        for region in ("fg", "bg"):
            tv = extra[level]["fg_t" if region == "fg" else "bg_s"]
            pts = sampling.points_on_rays(tv, o, d) if region == "fg" else sampling.points_on_rays(far * (1.0 - tv) + 3.0 * tv, o, d)
            B, N, _ = pts.shape
            cam = gather.world_to_camera(pts.reshape(-1, 3), poses)
            f = focal[0].repeat(2).clone()
            f[1] *= -1
            uv = gather.project(cam, f, centre[0][None])
            g = (uv * (gather.latent_scaling(240, 320) / torch.tensor([float(W), float(H)])) - 1.0).numpy()
            camn = cam.numpy()
            ok = np.stack([valid(g[..., 0], g[..., 1], 320, 240), valid(camn[..., 0], camn[..., 2], 160, 120),
                           valid(camn[..., 0], camn[..., 1], 160, 120), valid(camn[..., 1], camn[..., 2], 160, 120)], 0)   # (4 maps, NV, P)
            entries, rows_in, pairs, tiles = [], [], [], 0
            for gi in range(GROUPS):
                lo, hi = gi * RAYS_PER_GROUP * N, (gi + 1) * RAYS_PER_GROUP * N
                for t0 in range(lo, hi - 63, 64):
                    for v in range(NV):
                        m = ok[:, v, t0:t0 + 64].reshape(4, 4, 16)            # map, group, row
                        per = m.sum(-1)                                        # rows with weight per (map, group)
                        listed = per > 0
                        entries.append(int(listed.sum()))
                        rows_in.extend(per[listed].tolist())
                        pairs.append(int(m.sum()))
                        tiles += 1
            e, p = np.array(entries), np.array(pairs)
            r = np.array(rows_in) if rows_in else np.zeros(1)
            steps_now = np.where(e == 0, 0, np.ceil(np.maximum(e, 4) / 3) * 3)           # empty groups keep one entry; padded to x3 (approx.)
            steps_row = np.ceil(p / 16.0)
            print("%s_%s: tile-views %d | entries per tile-view mean %.2f (of 16), empty %.1f %% | rows with a tap per entry mean %.1f of 16 "
                  "(p10 %d, p50 %d, p90 %d) | (row, map) pairs per tile-view mean %.1f -> steps per chunk: now ~%.2f, rows packed 16 to a step %.2f"
                  % (region, "coarse" if level == 0 else "fine", tiles, e.mean(), 100.0 * (e == 0).mean(), r.mean(),
                     np.percentile(r, 10), np.percentile(r, 50), np.percentile(r, 90), p.mean(), steps_now.mean(), steps_row.mean()))


if __name__ == "__main__":
    main()
