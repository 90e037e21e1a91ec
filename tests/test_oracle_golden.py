"""CPU: the oracle (oracle/) against the committed fixtures produced by the
reference itself (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import cases
import oracle
from conftest import max_abs
from neo360_amd import synth

EXACT = 0.0


def test_g1_raygen(golden):
    g = golden("g1_raygen")
    for tag, (H, W) in (("s", (32, 32)), ("f", (480, 640))):
        dirs = oracle.rays.pixel_directions(H, W, 0.8 * W)
        for pi, az in enumerate((10.0, 130.0, 250.0)):
            c2w = synth.look_at_origin(az, 0.6 + 0.1 * pi, 0.3 - 0.2 * pi)[:3, :4]
            ro, vd, rd, rad = oracle.rays.camera_rays(dirs, c2w)
            if tag == "f":
                idx = g["idx_f"]
                ro, vd, rd, rad = ro[idx], vd[idx], rd[idx], rad[idx]
            assert max_abs(ro, g["o_%s%d" % (tag, pi)]) == EXACT
            assert max_abs(vd, g["v_%s%d" % (tag, pi)]) == EXACT
            assert max_abs(rd, g["d_%s%d" % (tag, pi)]) == EXACT
            assert max_abs(rad, g["r_%s%d" % (tag, pi)]) == EXACT


def test_g2_aabb_bit_exact(golden):
    g = golden("g2_aabb")
    boxes, o, d = cases.aabb_cases()
    nears, fars = [], []
    for bi, b in enumerate(boxes):
        hit, tmin, tmax = oracle.rays.aabb_slab_batch(b, o, d)
        assert np.array_equal(hit.astype(np.uint8), g["hit%d" % bi].numpy())
        assert np.array_equal(tmin, g["tmin%d" % bi].numpy())
        assert np.array_equal(tmax, g["tmax%d" % bi].numpy())
        nears.append(torch.Tensor(tmin[:, None]))
        fars.append(torch.Tensor(tmax[:, None]))
    assert 0.05 < float(g["hit0"].float().mean()) < 0.95
    _, _, mask = oracle.rays.merge_boxes(nears, fars)
    assert torch.equal(mask.to(torch.uint8), g["merged_mask"])


def test_g2_oriented_boxes_bit_exact(golden):
    """sample_rays_in_bbox / get_object_rays_in_bbox (neo360/helper.py:348-373) called on the reference itself:
    per-box masks, merged near / far (float32) and the merged mask, bit for bit."""
    g = golden("g2_aabb")
    RTs, o, d = cases.oriented_box_cases()
    near, far, mask, hits = oracle.rays.sample_rays_in_bbox(RTs, o, d)
    for bi, h in enumerate(hits):
        assert torch.equal(h.to(torch.uint8), g["ob_hit%d" % bi])
    assert torch.equal(near, g["ob_near"]) and torch.equal(far, g["ob_far"])
    assert torch.equal(mask.to(torch.uint8), g["ob_mask"])
    assert 0.2 < float(g["ob_mask"].float().mean()) < 0.8


def test_g3_encoding_and_samplers(golden):
    g = golden("g3_stages")
    x3 = synth.uniform(11, "pe3", (257, 3), -1.7, 1.7)
    x4 = synth.uniform(11, "pe4", (129, 4), -1.0, 1.0)
    assert max_abs(oracle.encoding.pos_enc(x3, 0, 10), g["pe3"]) == EXACT
    assert max_abs(oracle.encoding.pos_enc(x4, 0, 10), g["pe4"]) == EXACT
    assert max_abs(oracle.encoding.pos_enc(x3, 0, 4), g["pe3v"]) == EXACT
    rays = cases.strided_rays(96)
    far, ok = oracle.rays.sphere_exit_depth(rays["rays_o"], rays["rays_d"])
    assert bool(ok.all()) and max_abs(far, g["far"]) == EXACT
    inv_r = torch.flip(torch.linspace(0, 1, 33), dims=[-1])[None].repeat(96, 1).contiguous()
    assert max_abs(oracle.sampling.inverted_sphere_points(rays["rays_o"], rays["rays_d"], inv_r), g["outside"]) == EXACT
    near = torch.full((96, 1), 1e-4)
    t, p = oracle.sampling.neo_fg_level0(rays["rays_o"], rays["rays_d"], 32, near, far)
    assert max_abs(t, g["fg0_t"]) == EXACT and max_abs(p, g["fg0_p"]) == EXACT
    s, p4, lin = oracle.sampling.neo_bg_level0(rays["rays_o"], rays["rays_d"], 32, far, 3.0)
    assert max_abs(s, g["bg0_s"]) == EXACT and max_abs(p4, g["bg0_p"]) == EXACT and max_abs(lin, g["bg0_lin"]) == EXACT
    tv, pv = oracle.sampling.vanilla_level0(rays["rays_o"], rays["viewdirs"], 64, 0.2, 3.0)
    assert max_abs(tv[:2], g["v0_t"]) == EXACT and max_abs(pv[:2], g["v0_p"]) == EXACT


def test_g3_inverse_cdf(golden):
    g = golden("g3_stages")
    pc = cases.pdf_cases()
    for tag in ("asc", "desc"):
        bins, w = pc[tag]
        assert max_abs(oracle.sampling.piecewise_constant_samples(bins, w, 128), g["pdf_" + tag]) == EXACT
    # the descending (background) case is genuinely non-monotone: a real sort is required downstream
    assert bool((g["pdf_desc"][:, 1:] < g["pdf_desc"][:, :-1]).any())
    assert max_abs(oracle.sampling.piecewise_constant_samples(*pc["asc"], 128), g["pdf_asc_v"]) == EXACT


def test_g3_compositing(golden):
    g = golden("g3_stages")
    rgb, sigma, t, dirs, far = cases.composite_case()
    names = ("rgb", "acc", "w", "lam", "depth")
    for nm, v in zip(names, oracle.compositing.neo_composite(rgb, sigma, t, dirs, True, far)):
        assert max_abs(v, g["cfg_" + nm]) == EXACT, nm
    t_desc = torch.flip(t / t.max(), dims=[-1]).contiguous()
    for nm, v in zip(names, oracle.compositing.neo_composite(rgb, sigma, t_desc, dirs, False)):
        if v is not None:
            assert max_abs(v, g["cbg_" + nm]) == EXACT, nm
    for nm, v in zip(("rgb", "acc", "w", "depth"), oracle.compositing.vanilla_composite(rgb, sigma, t, dirs * 1.3, True)):
        assert max_abs(v, g["cv_" + nm]) == EXACT, nm


def test_g3_feature_lookups(golden):
    g = golden("g3_stages")
    scene = cases.small_scene()
    poses, focal, centre = synth.source_views(cases.NV, *cases.IMG_WH)
    pts = synth.uniform(13, "gpts", (16, 4, 3), -1.6, 1.6)
    assert max_abs(oracle.gather.world_to_camera(pts.reshape(-1, 3), poses), g["cam"]) == EXACT
    tri = oracle.gather.triplane_features(pts, scene["plane_xz"], scene["plane_xy"], scene["plane_yz"], poses)
    assert max_abs(tri, g["triplane"]) < 1e-6
    loc = oracle.gather.pixel_aligned_features(pts, scene["latent"], poses, focal, centre, scene["image_wh"])
    assert max_abs(loc, g["local"]) < 1e-6
    # zero padding really is exercised (points outside the planes)
    assert float((g["triplane"].abs().sum(-1) == 0).float().mean()) > 0.01


def test_bilinear_restatement_matches_torch_grid_sample():
    import torch.nn.functional as F
    maps = synth.normal(3, "gs_maps", (2, 5, 7, 9), 1.0)
    grid = synth.uniform(3, "gs_grid", (2, 300, 2), -1.3, 1.3)
    grid[0, :4] = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [1.0, -1.0], [0.0, 0.0]])
    want = F.grid_sample(maps, grid.unsqueeze(2), align_corners=True, mode="bilinear", padding_mode="zeros")
    want = want[..., 0].permute(0, 2, 1)
    assert max_abs(oracle.gather.bilinear_zero_pad(maps, grid), want) < 1e-6


def test_g3_mlps(golden):
    g = golden("g3_stages")
    for nv in (1, 3):
        P = 50
        sd = synth.nerfpp_mlp_state(21 + nv, "")
        x = synth.uniform(31, "mlp_x%d" % nv, (nv, P, 63), -1, 1)
        cond = synth.uniform(31, "mlp_c%d" % nv, (nv * P, 27), -1, 1)
        world = synth.normal(31, "mlp_w%d" % nv, (nv * P, 128), 0.3)
        local = synth.normal(31, "mlp_l%d" % nv, (nv * P, 512), 0.3)
        r, s = oracle.mlp.nerfpp_mlp(sd, "", x, cond, world, local, nv)
        assert max_abs(r, g["mlp%d_rgb" % nv].reshape(-1, 3)) < 1e-6
        assert max_abs(s, g["mlp%d_sigma" % nv].reshape(-1, 1)) < 1e-6
    sd = synth.vanilla_mlp_state(41, "")
    xe = synth.uniform(43, "vmlp_x", (6, 11, 63), -1, 1)
    ce = synth.uniform(43, "vmlp_c", (6, 27), -1, 1)
    r, s = oracle.mlp.vanilla_mlp(sd, "", xe, ce)
    assert max_abs(r, g["vmlp_rgb"]) < 1e-6 and max_abs(s, g["vmlp_sigma"]) < 1e-6


def test_g4_vanilla_config1(golden):
    """BASELINE config 1: 32x32 crop, 64+128 samples, CPU (plumbing)."""
    g = golden("g4_vanilla")
    for tag, gain in (("", 1.0), ("_sharp", 8.0)):
        res = oracle.vanilla.render(synth.vanilla_state(0, density_gain=gain), cases.crop_rays(32, 32), 0.2, 3.0)
        for lv in (0, 1):
            assert max_abs(res[lv][0], g["rgb%d%s" % (lv, tag)]) < 2e-6
            assert max_abs(res[lv][1], g["acc%d%s" % (lv, tag)]) < 2e-6
            assert max_abs(res[lv][2], g["depth%d%s" % (lv, tag)]) < 5e-6
    res = oracle.vanilla.render(synth.vanilla_state(0), cases.strided_rays(200), 0.2, 3.0, white_bkgd=True)
    assert max_abs(res[1][0], g["rgb1_white"]) < 2e-6 and max_abs(res[1][2], g["depth1_white"]) < 5e-6


def _neo_oracle(n_rays, chunk, n_coarse, n_fine, gain=1.0):
    scene = cases.small_scene()
    batch = cases.neo_batch(cases.strided_rays(n_rays))
    params = synth.nerf_tp_state(0, density_gain=gain)
    rgb0, rgb1, depth1, lam1 = [], [], [], []
    for i in range(0, n_rays, chunk):
        part = {k: (v if k.startswith("src_") else v[i:i + chunk]) for k, v in batch.items()}
        res = oracle.neo360.render(params, part, scene, n_coarse, n_fine)
        rgb0.append(res[0][0]); rgb1.append(res[1][0]); depth1.append(res[1][5]); lam1.append(res[1][4])
    return torch.cat(rgb0), torch.cat(rgb1), torch.cat(depth1), torch.cat(lam1)


def test_g4_neo_small_two_chunks(golden):
    """300 rays in chunks of 256: the view-direction tiling quirk and the short last chunk."""
    g = golden("g4_neo_small")
    rgb0, rgb1, depth1, lam1 = _neo_oracle(300, 256, 32, 64)
    assert max_abs(rgb0, g["rgb0"]) < 5e-6
    assert max_abs(rgb1, g["rgb1"]) < 5e-6
    assert max_abs(depth1, g["depth1"]) < 5e-5
    assert max_abs(lam1, g["lam1"]) < 5e-6


def test_g5_chunk_dependence_is_reproduced(golden):
    a, b = golden("g4_neo_c128"), golden("g4_neo_c64")
    # the reference's own outputs depend on the chunk size (quirk Q1) ...
    assert max_abs(a["rgb1"], b["rgb1"]) > 1e-5
    # ... and the oracle follows each of them
    for g, chunk in ((a, 128), (b, 64)):
        _, rgb1, depth1, _ = _neo_oracle(128, chunk, 32, 64)
        assert max_abs(rgb1, g["rgb1"]) < 5e-6
        assert max_abs(depth1, g["depth1"]) < 5e-5


def test_g6_mip360_stages(golden):
    from oracle import mip360
    g = golden("g6_mip360")
    basis = mip360.icosahedron_basis()
    assert max_abs(basis, g["basis"]) == 0.0
    from neo360_amd import geopoly
    assert max_abs(geopoly.icosahedron_basis(), g["basis"]) == 0.0
    means = synth.uniform(51, "mip_mean", (40, 7, 3), -2.5, 2.5)
    means[0] *= 0.2
    A = synth.uniform(51, "mip_cov", (40, 7, 3, 3), -0.05, 0.05)
    covs = A @ A.transpose(-1, -2)
    cm, cc = mip360.contract(means, covs)
    assert max_abs(cm, g["con_mean"]) < 5e-7          # closed-form Jacobian vs the reference's functorch autograd
    assert max_abs(cc, g["con_cov"]) < 5e-7
    lm, lv = mip360.lift_and_diagonalize(g["con_mean"], g["con_cov"], basis)
    assert max_abs(lm, g["lift_mean"]) == 0.0 and max_abs(lv, g["lift_var"]) == 0.0
    assert max_abs(mip360.integrated_pos_enc(lm, lv, 0, 12), g["ipe"]) == 0.0
    t = torch.sort(synth.uniform(53, "mip_t", (24, 33), 0.0, 1.0), dim=-1).values
    w = synth.uniform(53, "mip_w", (24, 32), 0.0, 1.0)
    w[1] = 0.0
    w[1, 5] = 1.0
    td, wd = mip360.max_dilate_weights(t, w, 0.01, (0.0, 1.0))
    assert max_abs(td, g["dil_t"]) == 0.0 and max_abs(wd, g["dil_w"]) == 0.0
    logits = torch.where(td[..., 2:-1] > td[..., 1:-2], torch.log(wd[..., 1:-1]), torch.full_like(wd[..., 1:-1], -torch.inf))
    assert max_abs(mip360.sample_intervals(td[..., 1:-1], logits, 32, (0.0, 1.0)), g["intervals"]) == 0.0


@pytest.mark.parametrize("tag,tf,gain,counts", [("a", 1.0, 1.0, (64, 32)), ("b", 0.3, 1.0, (64, 32)),
                                                ("sharp", 1.0, 6.0, (64, 32))])
def test_g6_mip360_end_to_end(golden, tag, tf, gain, counts):
    from oracle import mip360
    g = golden("g6_mip360")
    rend, hist = mip360.render(synth.mip360_state(0, density_gain=gain, weight_gain=0.5), cases.mip_rays(160), tf, 0.2, 3.0,
                               num_prop_samples=counts[0], num_nerf_samples=counts[1])
    for lv in range(3):
        assert max_abs(rend[lv]["rgb"], g["rgb%d_%s" % (lv, tag)]) < 5e-6
        assert max_abs(hist[lv]["sdist"], g["sdist%d_%s" % (lv, tag)]) < 5e-5
        assert max_abs(hist[lv]["weights"], g["w%d_%s" % (lv, tag)]) < 5e-5


@pytest.mark.parametrize("tag,n_rays,chunk,gain,white", [("a", 300, 256, 1.0, False), ("sharp", 128, 128, 8.0, False),
                                                         ("white", 96, 96, 1.0, True)])
def test_g7_pixelnerf_end_to_end(golden, tag, n_rays, chunk, gain, white):
    """PixelNeRF baseline decoder (vanilla_nerf/model_pixel.py:133-258) vs the reference's own output:
    two chunks with a short last one (the direction-tiling quirk), a sharp density field, white background."""
    g = golden("g7_pixelnerf")
    scene = cases.small_scene()
    batch = cases.neo_batch(cases.strided_rays(n_rays))
    state = synth.pixelnerf_state(0, density_gain=gain)
    lv = {k: [] for k in ("rgb0", "acc0", "depth0", "rgb1", "acc1", "depth1")}
    for i in range(0, n_rays, chunk):
        part = {k: (v[i:i + chunk] if k in ("rays_o", "rays_d", "viewdirs") else v) for k, v in batch.items()}
        res = oracle.pixelnerf.render(state, part, scene, 0.2, 2.5, white_bkgd=white)
        for l in (0, 1):
            lv["rgb%d" % l].append(res[l][0]); lv["acc%d" % l].append(res[l][1]); lv["depth%d" % l].append(res[l][2])
    for k, v in lv.items():
        tol = 1e-6 if k.endswith("0") else (5e-5 if "depth" in k else 1e-5)   # level 1 passes through the resampler
        assert max_abs(torch.cat(v), g["%s_%s" % (k, tag)]) < tol, (k, tag)


def test_g9_pillar_stage(golden):
    """oracle.pillar.floorplans == the reference's GridEncoder.forward up to the inputs of its floor-plan conv nets
    (world grid, masked directions, pixel-aligned lookup, depth_fc, the three axis scorers and softmaxes)."""
    g = golden("g9_pillar")
    sc = cases.small_scene()
    poses, focal, centre = synth.source_views(cases.NV, *cases.IMG_WH)
    got = oracle.pillar.floorplans(synth.pillar_state(0), sc["latent"], sc["image_wh"], poses, focal, centre, (12, 10, 8))
    for name, fp in zip(("yz", "xz", "xy"), got):
        assert max_abs(fp[..., ::4], g["fp_" + name]) < 1e-6, name
        assert max_abs(fp.double().sum(-1), g["sum_" + name]) < 1e-4
        assert max_abs((fp.double() ** 2).sum(-1), g["sq_" + name]) < 1e-4
