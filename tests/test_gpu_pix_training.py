"""GPU: PixelNeRF's training call (vanilla_nerf/model_pixel.py:174-258 under LitPixelNeRF.training_step) on the operator chain of
training.pix_render_train - every matrix product on the library's exact-fp32 GEMMs (neo_linear_forward / _input_grad /
_weight_grad), the latent projected per texel and gathered at the decoder's taps (neo_pix_gather_map)."""
import pytest
import torch

import cases
import oracle
from conftest import max_abs, record_parity
from neo360_amd import models, synth, training

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net(n0=16, n1=24, noise_std=0.0):
    scene = cases.small_scene()
    net = models.PixelNeRF(num_src_views=cases.NV, num_coarse_samples=n0, num_fine_samples=n1, noise_std=noise_std).to(DEV)
    net.load_state_dict(synth.pixelnerf_state(0))
    latent = scene["latent"].to(DEV)
    net.set_scene(latent, scene["image_wh"])
    return net, scene, latent


@pytest.mark.parametrize("rows,out_f,in_f,block,relu", [(5000, 128, 128, None, True), (3001, 128, 63, (0, 63), False),
                                                        (2000, 128, 27, (128, 155), False), (1500, 1, 128, None, False),
                                                        (700, 3, 128, None, False), (4097, 256, 512, (63, 575), True)])
def test_linear_matches_fp64_autograd(rows, out_f, in_f, block, relu):
    """training.linear = F.linear (+ ReLU) forward and backward (x, W incl. a column block of a wider matrix, b)."""
    g = torch.Generator(device=DEV).manual_seed(rows + out_f)
    wide = in_f if block is None else 640
    w_full = (torch.randn(out_f, wide, device=DEV, generator=g) * 0.1).requires_grad_(True)
    b = (torch.randn(out_f, device=DEV, generator=g) * 0.1).requires_grad_(True)
    x = torch.randn(rows, in_f, device=DEV, generator=g).requires_grad_(True)
    up = torch.randn(rows, out_f, device=DEV, generator=g)
    with torch.enable_grad():
        w = w_full if block is None else w_full[:, block[0]:block[1]]
        y = training.linear(x, w, b, relu=relu)
        got = torch.autograd.grad((y * up).sum(), [x, w_full, b])
        xd, wd, bd = x.detach().double().requires_grad_(True), w_full.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
        wb = wd if block is None else wd[:, block[0]:block[1]]
        ref = torch.nn.functional.linear(xd, wb, bd)
        ref = torch.relu(ref) if relu else ref
        want = torch.autograd.grad((ref * up.double()).sum(), [xd, wd, bd])
    assert max_abs(y.detach().double(), ref.detach()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    for a, r in zip(got, want):
        assert float((a.double() - r).abs().max()) <= 3e-6 * max(float(r.abs().max()), 1e-6)


def test_training_call_forward_equals_the_fused_kernels():
    """randomized=False through the operator chain = the fused no-grad render (<= 1e-4: the chain blends the latent after its
    projection in exact fp32, the fused split kernels blend the same projected map)."""
    net, _, latent = _net(n0=32, n1=32)
    rays = {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(192)).items()}
    fused = net(rays, False, False, 0.2, 2.5)
    net.differentiable = True
    with torch.enable_grad():
        chain = net(rays, False, False, 0.2, 2.5)
    assert chain[1][0].requires_grad
    for lv in (0, 1):
        for j in range(3):
            assert max_abs(chain[lv][j].detach(), fused[lv][j]) < 1e-4, (lv, j)


def test_training_step_gradients_vs_fp64_autograd():
    """loss = img2mse(coarse) + img2mse(fine) on 160 rays, randomized; gradients of all 36 parameter tensors and of the latent
    against fp64 autograd of oracle.pixelnerf.render at the library's sample positions: the library may miss the fp64 gradients by
    no more than 1.5 x what the reference's own fp32 arithmetic misses them by (+ 2e-5)."""
    R, n0, n1 = 160, 16, 24
    net, scene, latent = _net(n0, n1)
    sd = synth.pixelnerf_state(0)
    batch_c = cases.neo_batch(cases.strided_rays(R))
    rays = {k: v.to(DEV) for k, v in batch_c.items()}
    target = synth.uniform(92, "pix_target", (R, 3), 0.0, 1.0)
    names = sorted(sd)
    with torch.enable_grad():
        for p in net.parameters():
            p.requires_grad_(True)
        latent.requires_grad_(True)
        out, ts = training.pix_render_train(net, rays, True, False, 0.2, 2.5, latent, seed=7, return_samples=True)
        loss_g = ((out[0][0] - target.to(DEV)) ** 2).mean() + ((out[1][0] - target.to(DEV)) ** 2).mean()
        params = dict(net.named_parameters())
        g_g = torch.autograd.grad(loss_g, [params[k] for k in names] + [latent])

        def oracle_grads(dtype):
            cv = lambda v: v.to(dtype) if torch.is_floating_point(v) else v
            pp = {k: cv(v).clone().requires_grad_(True) for k, v in sd.items()}
            lat = cv(scene["latent"]).clone().requires_grad_(True)
            sc = dict(scene, latent=lat)
            want = oracle.pixelnerf.render(pp, {k: cv(v) for k, v in batch_c.items()}, sc, 0.2, 2.5, n_coarse=n0, n_fine=n1,
                                           samples=(cv(ts[0].cpu()), cv(ts[1].cpu())))
            loss = ((want[0][0] - cv(target)) ** 2).mean() + ((want[1][0] - cv(target)) ** 2).mean()
            return float(loss), torch.autograd.grad(loss, [pp[k] for k in names] + [lat])

        loss_c, g_c = oracle_grads(torch.float64)
        _, g_r = oracle_grads(torch.float32)
    assert abs(float(loss_g) - loss_c) < 1e-5 * max(1.0, abs(loss_c))
    rel = lambda x, ref: (float(x.abs().max()) / (float(ref.abs().max()) + 1e-15), float(x.norm()) / (float(ref.norm()) + 1e-30))
    worst = 0.0
    for nm, a, b, r in zip(names + ["latent"], g_g, g_c, g_r):
        a = a.detach().cpu().double()
        lib, ref = rel(a - b, b), rel(r.double() - b, b)
        worst = max(worst, lib[1])
        assert lib[0] <= 1.5 * ref[0] + 2e-5 and lib[1] <= 1.5 * ref[1] + 2e-5, (nm, lib, ref)
    record_parity("train_pixelnerf_module_call", max_rel_l2_grad_err_vs_fp64=worst, rays=R, loss_abs_err=abs(float(loss_g) - loss_c))


def test_randomized_module_call_and_noise_std():
    """PixelNeRF.forward(randomized=True) runs (it raised NotImplementedError until round 5), repeats for a seed, and `noise_std`
    adds the call's uniform stream 4 / 6 to the raw density (model_pixel.py:235-236) - the oracle fed the same tables agrees."""
    R, n0, n1, std, seed = 64, 16, 24, 0.75, 11
    net, scene, latent = _net(n0, n1, noise_std=std)
    batch_c = cases.neo_batch(cases.strided_rays(R))
    rays = {k: v.to(DEV) for k, v in batch_c.items()}
    a = net(rays, True, False, 0.2, 2.5, seed=seed)
    b = net(rays, True, False, 0.2, 2.5, seed=seed)
    assert torch.equal(a[1][0], b[1][0])
    out, ts = training.pix_render_train(net, rays, True, False, 0.2, 2.5, latent, seed=seed, return_samples=True)
    u = [training.rand_uniform(seed, 4 + 2 * lv, R, n).cpu() * std for lv, n in ((0, n0 + 1), (1, n0 + 1 + n1))]
    want = oracle.pixelnerf.render(synth.pixelnerf_state(0), batch_c, scene, 0.2, 2.5, n_coarse=n0, n_fine=n1,
                                   samples=(ts[0].cpu(), ts[1].cpu()), sigma_noise=u)
    for lv in (0, 1):
        assert max_abs(out[lv][0], want[lv][0]) < 1e-4 and max_abs(out[lv][1], want[lv][1]) < 1e-4
    quiet, _, _ = _net(n0, n1, noise_std=0.0)
    assert max_abs(quiet(rays, True, False, 0.2, 2.5, seed=seed)[0][0], a[0][0]) > 1e-5


def test_fused_chain_equals_the_per_layer_operators():
    """neo_pix_mlp_train_forward_pre / _backward_pre (one native chain each way, round 6) against pixel_mlp_projected (one
    operator per layer + torch glue): same exact-fp32 GEMMs in the same order, so outputs and all 19 gradients agree to rounding
    of the accumulation order (view means / split-K slices)."""
    NV, P = 3, 1700
    net, _, _ = _net()
    mlp = net.fine_mlp
    g = torch.Generator(device=DEV).manual_seed(5)
    x_enc = torch.randn(NV, P, 63, device=DEV, generator=g)
    cond = torch.randn(NV * P, 27, device=DEV, generator=g)
    pre0 = torch.randn(NV * P, 128, device=DEV, generator=g) * 0.5
    up_rgb, up_sigma = torch.randn(P, 3, device=DEV, generator=g), torch.randn(P, 1, device=DEV, generator=g)
    layers = mlp.ordered_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    res = []
    with torch.enable_grad():
        for p in params:
            p.requires_grad_(True)
        for fn in (training.pixel_mlp_fused, training.pixel_mlp_projected):
            pre = pre0.clone().requires_grad_(True)
            rgb, sigma = fn(mlp, x_enc, cond, pre, NV)
            grads = torch.autograd.grad((rgb * up_rgb).sum() + (sigma * up_sigma).sum(), [pre] + params)
            res.append((rgb.detach(), sigma.detach(), grads))
    (rgb_a, sig_a, g_a), (rgb_b, sig_b, g_b) = res
    assert max_abs(rgb_a, rgb_b) <= 2e-6 * max(1.0, float(rgb_b.abs().max()))
    assert max_abs(sig_a, sig_b) <= 2e-6 * max(1.0, float(sig_b.abs().max()))
    # (the chain runs the bottleneck and view layer 0 on the view means - a reassociation; a unit that rounds to +0 on one side and to a
    # tiny positive number on the other flips its ReLU derivative for ONE row: rows are compared one by one, see test_gpu_host_r6.py)
    flipped = int(((g_a[0] - g_b[0]).abs().amax(dim=1) > 1e-5 * max(float(g_b[0].abs().max()), 1e-6)).sum())
    assert flipped <= 3, flipped
    for i, (a, b) in enumerate(zip(g_a, g_b)):
        if i == 0:
            continue
        if i == 1:                                                  # pts_linears.0: the latent columns belong to the texel-space GEMM
            assert float(a[:, 63:].abs().max()) == 0.0
            a, b = a[:, :63], b[:, :63]
        if flipped == 0:
            assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-6), i
        else:
            assert float((a - b).norm()) <= 5e-3 * float(b.norm()) + 1e-12, (i, flipped)
