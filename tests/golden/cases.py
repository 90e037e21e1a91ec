"""Seeded input builders shared by the golden generator and the tests.

Pure functions of integer seeds (neo360_amd.synth): the fixtures under
tests/golden/*.npz hold only EXPECTED OUTPUTS of the reference; every input is
regenerated here, identically, on any machine.  No reference imports.
"""
import math

import numpy as np
import torch

from neo360_amd import synth

# ---- small scene used by every NeO-360 fixture ----------------------------------
NV = 3
IMG_WH = (64, 48)          # source image size the latents were "encoded" from
PLANE_HW = (12, 16)
LATENT_HW = (24, 32)


def crop_rays(H, W, azimuth=40.0, radius=0.6, height=0.3):
    """Rays of an H x W pinhole camera (focal 0.8 W) on the test orbit, built with
    plain fp64 NumPy so that fixtures do not depend on either implementation of
    ray generation.  Every ray hits the unit sphere (camera is inside it)."""
    c2w = synth.look_at_origin(azimuth, radius, height).double().numpy()
    f = 0.8 * W
    jj, ii = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    cam = np.stack([(ii - W / 2) / f, -(jj - H / 2) / f, -np.ones_like(ii)], -1).reshape(-1, 3)
    d = cam @ c2w[:3, :3].T
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(c2w[:3, 3], d.shape)
    o32 = torch.from_numpy(np.ascontiguousarray(o, dtype=np.float32))
    d32 = torch.from_numpy(d.astype(np.float32))
    return dict(rays_o=o32, rays_d=d32.clone(), viewdirs=d32.clone())


def take(rays, idx):
    return {k: v[idx].contiguous() for k, v in rays.items()}


def neo_batch(rays, nv=NV):
    """Adds the src_* keys of the reference's batch dict (datasets/nerds360_ae.py)."""
    poses, focal, centre = synth.source_views(nv, IMG_WH[0], IMG_WH[1])
    out = dict(rays)
    out.update(src_poses=poses, src_focal=focal, src_c=centre,
               src_imgs=torch.zeros(nv, 3, IMG_WH[1], IMG_WH[0]))
    return out


def small_scene(seed=7, nv=NV):
    sc = synth.scene_features(seed, nv, 128, PLANE_HW, 512, LATENT_HW, std=0.5)
    sc["image_wh"] = (float(IMG_WH[0]), float(IMG_WH[1]))
    return sc


def strided_rays(n, H=48, W=64, **kw):
    """n rays spread over an H x W frame (deterministic stride pattern)."""
    rays = crop_rays(H, W, **kw)
    total = H * W
    idx = torch.from_numpy(((np.arange(n, dtype=np.int64) * 2654435761) % total))
    return take(rays, idx)


# ---- the BASELINE C3 configuration at full size (640x480, 3 views, 128 + 256 samples) -----------------------
FULL_WH = (640, 480)
FULL_PLANE_HW = (120, 160)
FULL_LATENT_HW = (240, 320)


def full_scene(seed=0, nv=NV, std=0.1):
    """Stand-ins for the scene encoder's outputs at the reference's shapes for 640x480 sources (tri-planes
    3 x 128 x 120 x 160, latent 3 x 512 x 240 x 320, N(0, 0.1^2)), from the hash generator: the build container
    (reference run -> fixture g4_neo_full), the GPU tests and bench.py regenerate them bit for bit."""
    sc = synth.scene_features(seed, nv, 128, FULL_PLANE_HW, 512, FULL_LATENT_HW, std=std)
    sc["image_wh"] = (float(FULL_WH[0]), float(FULL_WH[1]))
    return sc


def full_strip_index(n, stride=601, start_row=230):
    """n rays of the 640x480 frame: a stride-601 walk starting at the image centre row (hits every region of the
    frame; as ONE reference chunk when n = 1024)."""
    W, H = FULL_WH
    return (torch.arange(n, dtype=torch.int64) * stride + start_row * W) % (H * W)


def full_batch(n, nv=NV, stride=601, start_row=230, **camera):
    """Rays of the bench camera (look_at_origin(40 deg), focal 0.8 W; fp64 NumPy ray generation so the fixture does
    not depend on either ray generator) + the bench's source cameras.  camera: azimuth / radius / height of another
    target pose on the test orbit (crop_rays)."""
    W, H = FULL_WH
    rays = take(crop_rays(H, W, **camera), full_strip_index(n, stride, start_row))
    poses, focal, centre = synth.source_views(nv, W, H)
    rays.update(src_poses=poses, src_focal=focal, src_c=centre, src_imgs=torch.zeros(nv, 3, H, W))
    return rays


# More reference chunks of the C3 scene at full size (fixtures g4_neo_full_<tag>, VERDICT r3 task 6: the parity rule for
# ill-conditioned rays was calibrated on ONE chunk from ONE camera): another strip of the bench frame, two other target
# poses (one close to the unit sphere, one low), and five source views at full map size.
FULL_B = {
    "b1": dict(nv=NV, stride=389, start_row=57),
    "b2": dict(nv=NV, stride=601, start_row=230, azimuth=130.0, radius=0.6, height=0.45),
    "b3": dict(nv=NV, stride=463, start_row=311, azimuth=250.0, radius=0.78, height=0.1),
    "b4": dict(nv=5, stride=601, start_row=140),
    # round 5 (VERDICT r4 task 8): beyond random-init weights on N(0, 0.1) features -
    # b5: TRAINED-LIKE sharp densities (density head x 8: opaque surfaces, weights concentrated in a few samples),
    # b6: a second feature seed at std 0.5 (five times the activations' scale: ReLU units far from their kinks, larger cdf steps)
    "b5": dict(nv=NV, stride=521, start_row=400, gain=8.0),
    "b6": dict(nv=NV, stride=433, start_row=19, seed=1, std=0.5),
}
_FULL_SCENE_KEYS = ("seed", "std")


def full_gain(tag):
    """density gain of the MLP weights of fixture g4_neo_full_<tag> (synth.nerf_tp_state(0, density_gain=...))."""
    return float(FULL_B.get(tag, {}).get("gain", 1.0)) if tag else 1.0


def full_case(tag, n=1024):
    """(scene, batch) of fixture g4_neo_full_<tag>; tag None / "" = the round-3 chunk (g4_neo_full)."""
    if not tag:
        return full_scene(), full_batch(n)
    kw = dict(FULL_B[tag])
    nv = kw.pop("nv")
    kw.pop("gain", None)
    scene_kw = {k: kw.pop(k) for k in _FULL_SCENE_KEYS if k in kw}
    return full_scene(nv=nv, **scene_kw), full_batch(n, nv=nv, **kw)


def aabb_cases(seed=3, n=4096):
    """Rays in box frame (float64) vs three boxes; includes exact-zero direction
    components, origins inside the box, and rays parallel to faces."""
    u = synth.uniform01(seed, "aabb_o", n * 3).reshape(n, 3) * 4.0 - 2.0
    d = synth.uniform01(seed, "aabb_d", n * 3).reshape(n, 3) * 2.0 - 1.0
    aim = synth.uniform01(seed, "aabb_aim", n * 3).reshape(n, 3) * 1.6 - 0.8
    d[::2] = (aim - u)[::2]          # half of the rays are aimed at the boxes' neighbourhood
    d[::7, 0] = 0.0
    d[::11, 1] = 0.0
    d[::13, 2] = 0.0
    d[5::97] = np.array([0.0, 0.0, 1.0])
    u[3::17] *= 0.1  # origins inside the unit-ish boxes
    boxes = np.array([
        [[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]],
        [[-1.0, -0.25, 0.0], [0.25, 0.75, 0.5]],
        [[0.2, 0.2, 0.2], [1.5, 0.4, 1.2]],
    ])
    return boxes, u, d


def oriented_box_cases(seed=4, n=3000):
    """World-frame rays (float32, as the dataset holds them) and four ORIENTED boxes in the reference's RTs format
    (datasets/nerds360_ae.py: R (3,3), T (3,), s (2,3) bounds in the box frame): rotated + translated boxes, one
    axis-aligned, exact-zero direction components, origins inside boxes, rays that miss everything."""
    o = (synth.uniform01(seed, "ob_o", n * 3).reshape(n, 3) * 3.0 - 1.5).astype(np.float32)
    aim = (synth.uniform01(seed, "ob_aim", n * 3).reshape(n, 3) * 1.2 - 0.6).astype(np.float32)
    d = aim - o
    d[::9] = (synth.uniform01(seed, "ob_d", n * 3).reshape(n, 3) * 2.0 - 1.0).astype(np.float32)[::9]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[::23, 1] = 0.0
    d[7::31, 0] = 0.0
    o[5::19] *= 0.05                                             # origins inside the central boxes

    def rot(ax, ang):
        c, s = math.cos(ang), math.sin(ang)
        m = {"x": [[1, 0, 0], [0, c, -s], [0, s, c]], "y": [[c, 0, s], [0, 1, 0], [-s, 0, c]], "z": [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[ax]
        return np.array(m, dtype=np.float64)

    Rs = [rot("z", 0.4) @ rot("x", -0.3), np.eye(3), rot("y", 1.1), rot("x", 0.7) @ rot("z", 2.0)]
    Ts = [np.array([0.1, -0.05, 0.0]), np.array([-0.4, 0.3, 0.1]), np.array([0.5, 0.5, -0.2]), np.array([0.0, -0.6, 0.3])]
    ss = [np.array([[-0.3, -0.2, -0.25], [0.3, 0.2, 0.25]]), np.array([[-0.2, -0.2, -0.2], [0.2, 0.2, 0.2]]),
          np.array([[-0.15, -0.4, -0.1], [0.15, 0.4, 0.1]]), np.array([[-0.25, -0.1, -0.3], [0.25, 0.1, 0.3]])]
    return dict(R=Rs, T=Ts, s=ss), o, d


def pdf_cases(seed=5, R=96, nb=64):
    """(bins, weights) pairs for the inverse-CDF sampler: ascending bins with
    positive weights, descending bins (the background branch), all-zero weights,
    one spiky row and tied bins."""
    w = synth.uniform(seed, "pdf_w", (R, nb - 1), 0.0, 1.0)
    w[1] = 0.0
    w[2] = 0.0
    w[2, 17] = 5.0
    w[3, : nb // 2] = 0.0
    edges = torch.cumsum(synth.uniform(seed, "pdf_e", (R, nb + 1), 0.01, 1.0), dim=-1)
    edges = edges / edges[:, -1:]
    mids = 0.5 * (edges[:, 1:] + edges[:, :-1])
    mids[4, 10:14] = mids[4, 10]
    return dict(asc=(mids.contiguous(), w), desc=(torch.flip(mids, dims=[-1]).contiguous(), w))


def composite_case(seed=9, R=64, N=129):
    rgb = synth.uniform(seed, "c_rgb", (R, N, 3), 0.0, 1.0)
    sigma = synth.uniform(seed, "c_sig", (R, N, 1), 0.0, 6.0)
    sigma[0] = 0.0
    sigma[1] = 80.0
    t = torch.cumsum(synth.uniform(seed, "c_t", (R, N), 0.002, 0.03), dim=-1)
    dirs = torch.nn.functional.normalize(synth.uniform(seed, "c_d", (R, 3), -1.0, 1.0), dim=-1) * 1.0
    far = t[:, -1:] + 0.02
    return rgb, sigma, t, dirs, far


def mip_rays(n, radius_px=0.0012):
    """Rays + per-ray cone radii for the Mip-NeRF 360 fixtures (radii as get_rays would give
    for a ~640-wide frame, with a deterministic +-20% spread so the radius really matters)."""
    rays = strided_rays(n)
    spread = synth.uniform(61, "mip_radii", (n, 1), 0.8, 1.2)
    rays["radii"] = spread * radius_px
    return rays
