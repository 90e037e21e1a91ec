"""CPU: training-side oracle pieces against reference-generated fixture g8_training (randomized sampling with the
draws injected into the reference through a patched torch.rand, NeRF_TP's out_depth=False tuple), known-answer
vectors of the counter-based generator, and the prefix-sum distortion loss against its O(N^2) definition."""
import numpy as np
import torch

import cases
import oracle
from conftest import max_abs
from neo360_amd import synth
from oracle import training as T

SEED, NR, NC, NF = 1234, 64, 16, 24


def _philox_block(counter, key):
    M0, M1 = 0xD2511F53, 0xCD9E8D57
    x, (k0, k1) = list(counter), key
    for _ in range(10):
        p0, p1 = M0 * x[0], M1 * x[2]
        x = [(p1 >> 32) ^ x[1] ^ k0, p1 & 0xFFFFFFFF, (p0 >> 32) ^ x[3] ^ k1, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return x


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32_10."""
    assert _philox_block([0] * 4, (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert _philox_block([0xFFFFFFFF] * 4, (0xFFFFFFFF,) * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _philox_block([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    # the vectorised oracle generator is that block function, word 0, top 24 bits
    seed = 0x299F31D0A4093822
    u = T.philox_uniform(seed, 5, 7, 9)
    for r, c in ((0, 0), (3, 8), (6, 2)):
        want = np.float32(_philox_block([r, c, 5, 0], (seed & 0xFFFFFFFF, seed >> 32))[0] >> 8) * np.float32(2.0 ** -24)
        assert float(u[r, c]) == float(want)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0


def test_stratified_and_randomized_samplers_vs_reference(golden):
    g = golden("g8_training")
    rays = cases.strided_rays(NR)
    far, _ = oracle.rays.sphere_exit_depth(rays["rays_o"], rays["rays_d"])
    fg, bg = T.neo_level0_randomized(far, NC, T.philox_uniform(SEED, 0, NR, NC + 1), T.philox_uniform(SEED, 1, NR, NC + 1))
    assert max_abs(fg, g["strat_fg"]) == 0.0 and max_abs(bg, g["strat_bg"]) == 0.0
    pc = cases.pdf_cases()
    up = T.philox_uniform(SEED, 7, pc["asc"][0].shape[0], 48)
    for tag in ("asc", "desc"):
        got = oracle.sampling.piecewise_constant_samples(pc[tag][0], pc[tag][1], 48, u=up)
        assert max_abs(got, g["pdf_rand_" + tag]) == 0.0


def test_training_tuple_vs_reference(golden):
    """NeRF_TP.forward(out_depth=False): deterministic, white background, and randomized=True on injected draws."""
    g = golden("g8_training")
    params, scene = synth.nerf_tp_state(0), cases.small_scene()
    batch = cases.neo_batch(cases.strided_rays(NR))
    uni = dict(fg0=T.philox_uniform(SEED, 0, NR, NC + 1), bg0=T.philox_uniform(SEED, 1, NR, NC + 1),
               fg1=T.philox_uniform(SEED, 2, NR, NF), bg1=T.philox_uniform(SEED, 3, NR, NF))
    names = ("rgb", "fg_w", "bg_w", "fg_sd", "bg_sd", "bg_acc")
    for tag, white, u in (("det", False, None), ("white", True, None), ("rand", False, uni)):
        res = oracle.neo360.render(params, batch, scene, NC, NF, white_bkgd=white, out_depth=False, uniforms=u)
        for lv in (0, 1):
            for nm, v in zip(names, res[lv]):
                assert max_abs(v, g["%s_%s%d" % (tag, nm, lv)]) < 5e-6, (tag, nm, lv)


def test_distloss_prefix_sums_equal_the_definition():
    gen = torch.Generator().manual_seed(3)
    w = torch.rand(7, 40, generator=gen, dtype=torch.float64)
    w = w / w.sum(-1, keepdim=True) * 0.8
    m = torch.sort(torch.rand(7, 40, generator=gen, dtype=torch.float64), dim=-1).values
    a, b = T.eff_distloss(w, m, 1.0 / 40), T.distloss_bruteforce(w, m, 1.0 / 40)
    assert abs(float(a) - float(b)) < 1e-14
    # analytic gradient of the published backward == autograd of the forward
    with torch.enable_grad():          # conftest switches autograd off globally
        w.requires_grad_(True)
        (gw,) = torch.autograd.grad(T.eff_distloss(w, m, 1.0 / 40), w)
    wd = w.detach()
    wc, wmc = wd.cumsum(-1), (wd * m).cumsum(-1)
    w_pre, wm_pre = wc - wd, wmc - wd * m
    w_suf, wm_suf = wc[:, -1:] - wc, wmc[:, -1:] - wmc
    manual = (2.0 * (1.0 / 40) * wd / 3.0 + 2.0 * (m * (w_pre - w_suf) + (wm_suf - wm_pre))) / 7
    assert float((gw - manual).abs().max()) < 1e-14


def test_given_samples_hooks_of_the_pixelnerf_and_mip_oracles():
    """The hooks the GPU training tests use to evaluate the oracle at the library's sample positions: feeding an oracle its own
    positions reproduces its render bit for bit; `sigma_noise` (PixelNeRF) and `jitters` (Mip-NeRF 360) change it; the Mip oracle's
    sample positions carry no gradient (stop_level_grad)."""
    from oracle import mip360
    scene, sd = cases.small_scene(), synth.pixelnerf_state(0)
    b = cases.neo_batch(cases.strided_rays(8))
    out, ex = oracle.pixelnerf.render(sd, b, scene, 0.2, 2.5, n_coarse=8, n_fine=8, keep=True)
    again = oracle.pixelnerf.render(sd, b, scene, 0.2, 2.5, n_coarse=8, n_fine=8, samples=(ex[0]["t"], ex[1]["t"]))
    assert torch.equal(out[1][0], again[1][0]) and torch.equal(out[0][2], again[0][2])
    noisy = oracle.pixelnerf.render(sd, b, scene, 0.2, 2.5, n_coarse=8, n_fine=8, samples=(ex[0]["t"], ex[1]["t"]),
                                    sigma_noise=[torch.full((8, 9), 0.5), torch.full((8, 17), 0.5)])
    assert max_abs(noisy[1][0], out[1][0]) > 1e-4
    msd, rays = synth.mip360_state(0, weight_gain=0.25), cases.mip_rays(6)
    r, h = mip360.render(msd, rays, 0.5, 0.2, 3.0, num_prop_samples=8, num_nerf_samples=4)
    r2, _ = mip360.render(msd, rays, 0.5, 0.2, 3.0, num_prop_samples=8, num_nerf_samples=4, sdist_given=[x["sdist"] for x in h])
    assert torch.equal(r[2]["rgb"], r2[2]["rgb"])
    eps = float(torch.finfo(torch.float32).eps)
    mj = lambda n: (1 - (eps + (1 - eps) / n)) / (n - 1) - eps
    r3, h3 = mip360.render(msd, rays, 0.5, 0.2, 3.0, num_prop_samples=8, num_nerf_samples=4,
                           jitters=[torch.full((6, 1), 0.9 * mj(8)), torch.full((6, 1), 0.5 * mj(8)), torch.full((6, 1), 0.3 * mj(4))])
    assert max_abs(h3[0]["sdist"], h[0]["sdist"]) > 1e-4 and all(not x["sdist"].requires_grad for x in h3)


def test_view_branch_on_the_view_means_is_a_reassociation():
    """What the fused training chains rely on (csrc/train_chain.h, round 6): behind relu(L3_v) the reference's NeRFPPMLP is linear up to
    the view means - no activation on the bottleneck, view layer 0 averaged over the views BEFORE its ReLU (neo360/model.py:139-150) -
    so views_linear.0([bottleneck(h3_v) | cond_v]) averaged over v equals views_linear.0([bottleneck(mean_v h3_v) | mean_v cond_v]).
    Checked on the oracle's own forward in float64 (values and parameter gradients): equal to rounding."""
    from oracle import mlp as M
    nv, P = 3, 50
    g = torch.Generator().manual_seed(3)
    sd = {k: v.double() for k, v in synth.nerf_tp_state(0).items() if k.startswith("fg_fine_mlp.")}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x_enc = torch.randn(nv, P, 63, generator=g, dtype=torch.float64)
    cond = torch.randn(nv * P, 27, generator=g, dtype=torch.float64)
    world = torch.randn(nv * P, 128, generator=g, dtype=torch.float64) * 0.3
    local = torch.randn(nv * P, 512, generator=g, dtype=torch.float64) * 0.3
    pre = "fg_fine_mlp."
    with torch.enable_grad():
        rgb_ref, sig_ref = M.nerfpp_mlp(params, pre, x_enc, cond, world, local, nv)

        lin = lambda name, x: torch.nn.functional.linear(x, params[pre + name + ".weight"], params[pre + name + ".bias"])
        x0 = torch.cat([x_enc.reshape(-1, 63), local, world], dim=-1)
        h = x0
        for i in range(4):
            h = torch.relu(lin("pts_linears.%d" % i, h))
            if i == 2:
                h = torch.cat([h, x0], dim=-1)
        hm = h.reshape(nv, P, -1).mean(0)                                   # mean_v relu(L3_v)
        cm = cond.reshape(nv, P, -1).mean(0)
        y = torch.relu(lin("views_linear.0", torch.cat([lin("bottleneck_layer", hm), cm], dim=-1)))
        y = torch.relu(lin("views_linear.1", y))
        rgb, sig = lin("rgb_layer", y), lin("density_layer", hm)
        assert max_abs(rgb, rgb_ref) < 1e-12 and max_abs(sig, sig_ref) < 1e-12
        names = sorted(params)
        up = torch.randn(P, 3, generator=g, dtype=torch.float64)
        g_ref = torch.autograd.grad((rgb_ref * up).sum() + sig_ref.sum(), [params[k] for k in names])
        g_new = torch.autograd.grad((rgb * up).sum() + sig.sum(), [params[k] for k in names])
    for k, a, b in zip(names, g_ref, g_new):
        assert float((a - b).abs().max()) <= 1e-11 * max(1.0, float(a.abs().max())), k
