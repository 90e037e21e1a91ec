"""Host-side measurement (no GPU), round 6: how much cache-line work does the gather half of k_tp_mlp_hp present to a CU's
texture-address unit, and what would a per-tile-view dedup (each unique texel fetched once, blends from LDS) leave of it?

The energy budget (profiles/r06_energy_budget.log) prices the DIVERGENT part of the gathers - distinct cache lines per load
instruction and the L2 -> L1 traffic behind them - at 22 % of an inside-sphere launch (NEO_TP_ABLATE 256: every tap reads texel 0
of its map, loads kept).  A load instruction of the kernel covers 4 ADJACENT samples x one tap x 256 B; the address unit works
per distinct 128-B line.  Counted here on the bench geometry, per 64-point tile and source view:
  now        sum over the kernel's load instructions of the distinct texels among their 4 rows   (x lines per 256-B piece)
  unique     distinct texels of the whole tile-view                                              (x lines per texel)
  cells      runs of consecutive samples in one bilinear cell x 4 texels (the cheap dedup: no hashing, duplicates between
             neighbouring cells stay)
Same rays / camera / sample positions as tools/footprint_study.py."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from neo360_amd import synth            # noqa: E402
import oracle                           # noqa: E402
from oracle import gather, rays as rays_mod, sampling   # noqa: E402
from footprint_study import taps, H, W, NV, GROUPS, RAYS_PER_GROUP   # noqa: E402


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
    res = {}
    for level in range(2):
        tv = extra[level]["fg_t"]
        pts = sampling.points_on_rays(tv, o, d)
        B, N, _ = pts.shape
        cam = gather.world_to_camera(pts.reshape(-1, 3), poses)
        f = focal[0].repeat(2).clone()
        f[1] *= -1
        uv = gather.project(cam, f, centre[0][None])
        g = (uv * (gather.latent_scaling(240, 320) / torch.tensor([float(W), float(H)])) - 1.0).numpy()
        camn = cam.numpy()
        maps = {"latent": (taps(g[..., 0], g[..., 1], 320, 240), 1024),          # projected latent: 1 KB per texel
                "plane_xz": (taps(camn[..., 0], camn[..., 2], 160, 120), 512),
                "plane_xy": (taps(camn[..., 0], camn[..., 1], 160, 120), 512),
                "plane_yz": (taps(camn[..., 1], camn[..., 2], 160, 120), 512)}
        key = "fg_%s" % ("coarse" if level == 0 else "fine")
        tot = dict(now=0.0, unique=0.0, cells=0.0, requested=0.0)
        per = {}
        ntv = 0
        for name, (t, texel_bytes) in maps.items():
            lines = texel_bytes // 128
            now = uniq = cells = 0
            n = 0
            for gi in range(GROUPS):
                lo, hi = gi * RAYS_PER_GROUP * N, (gi + 1) * RAYS_PER_GROUP * N
                for t0 in range(lo, hi - 63, 64):
                    for v in range(NV):
                        tt = t[v, t0:t0 + 64]                                     # (64 rows, 4 taps)
                        n += 1
                        # the kernel's instruction: rows {4w..4w+3} + 16q, one tap
                        for q in range(4):
                            for w in range(4):
                                rows = tt[16 * q + 4 * w:16 * q + 4 * w + 4]
                                for k in range(4):
                                    now += len(np.unique(rows[:, k]))
                        uniq += len(np.unique(tt))
                        change = np.ones(64, bool)
                        change[1:] = (tt[1:] != tt[:-1]).any(axis=1)
                        cells += 4 * int(change.sum())
            per[name] = dict(now_texels=now / n, unique_texels=uniq / n, cell_run_texels=cells / n, lines_per_texel=lines)
            tot["now"] += now / n * lines
            tot["unique"] += uniq / n * lines
            tot["cells"] += cells / n * lines
            tot["requested"] += 256 * lines
            ntv = n
        res[key] = dict(per_map=per, lines_per_tile_view=tot, tile_views=ntv,
                        ratio_now_over_unique=tot["now"] / tot["unique"], ratio_now_over_cells=tot["now"] / tot["cells"])
        print(key, json.dumps(res[key], indent=1))
    with open(os.path.join(ROOT, "profiles", "r06_gather_linework.json"), "w") as fh:
        json.dump(dict(note=__doc__, result=res), fh, indent=1)


if __name__ == "__main__":
    main()
