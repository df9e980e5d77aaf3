"""GPU tests (-m gpu) of the training path: HIP backward kernels behind torch.autograd.Function,
against PyTorch autograd of the CPU oracle on the same inputs and cotangents.

Tolerance: gradients are sums over up to ~10^4 samples of f32 products in a different order than
ATen's; per tensor we require  max|got - want| <= 2e-4 * max|want| + 1e-7  (observed ~1e-6).
"""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available()
    from dm_nerf_amd import _lib, autograd
    from dm_nerf_amd.networks import dm_nerf as M, helpers as H, render as R
    _lib.load()
    return types.SimpleNamespace(M=M, H=H, R=R, G=autograd)


def tclose(got, want, what, rel=2e-4):
    got, want = got.detach().cpu().double(), want.detach().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = float((got - want).abs().max())
    scale = float(want.abs().max())
    assert err <= rel * scale + 1e-7, (what, err, scale)
    return err / (scale + 1e-30)


def model_from(A, sd, ins_num):
    m = A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
    m.load_state_dict(sd)
    return m.cuda().train()


def test_composite_backward_vs_autograd(A):
    for N, S, C, seed in ((7, 64, 14, 1), (5, 192, 60, 2), (3, 70, 3, 3)):
        g = torch.Generator().manual_seed(seed)
        raw = torch.randn(N, S, 4 + C, generator=g)
        raw[..., 3] = raw[..., 3] * 2 + 0.3
        raw[0, S // 2, 3] = 50.0                                   # near-opaque sample: T ~ 1e-10 afterwards
        z = torch.sort(torch.rand(N, S, generator=g) * 11 + 4, -1)[0]
        d = torch.randn(N, 3, generator=g)
        ct = [torch.randn(N, 3, generator=g), torch.randn(N, S, generator=g), torch.randn(N, generator=g),
              torch.randn(N, C - 1, generator=g)]
        r0 = raw.clone().requires_grad_(True)
        outs = O.render_train(r0, z, d)
        loss = sum((o * c).sum() for o, c in zip(outs, ct))
        want, = torch.autograd.grad(loss, r0)
        r1 = raw.cuda().requires_grad_(True)
        outs = A.R.render_train(r1, z.cuda(), d.cuda())
        loss = sum((o * c.cuda()).sum() for o, c in zip(outs, ct))
        got, = torch.autograd.grad(loss, r1)
        tclose(got, want, f"d_raw N={N} S={S} C={C}")
        # only rgb + ins cotangents (what the reference's losses produce): weights/depth grads absent
        r1 = raw.cuda().requires_grad_(True)
        o = A.R.render_train(r1, z.cuda(), d.cuda())
        got, = torch.autograd.grad((o[0] * ct[0].cuda()).sum() + (o[3] * ct[3].cuda()).sum(), r1)
        r0 = raw.clone().requires_grad_(True)
        o = O.render_train(r0, z, d)
        want, = torch.autograd.grad((o[0] * ct[0]).sum() + (o[3] * ct[3]).sum(), r0)
        tclose(got, want, "d_raw rgb+ins only")
        # the ins path is detached from sigma (render.py:22-23)
        r1 = raw.cuda().requires_grad_(True)
        o = A.R.render_train(r1, z.cuda(), d.cuda())
        g_ins_only, = torch.autograd.grad((o[3] * ct[3].cuda()).sum(), r1)
        assert float(g_ins_only[..., :4].abs().max()) == 0.0


def _oracle_mlp_grads(sd, rays_o, rays_d, z, cot):
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
    vd = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    x = torch.cat([O.embed(pts.reshape(-1, 3), 10), O.embed(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
    raw = O.mlp_forward(sdg, x).reshape(z.shape[0], z.shape[1], -1)
    (raw * cot).sum().backward()
    return raw.detach(), {k: v.grad for k, v in sdg.items()}


def test_mlp_backward_vs_autograd(A):
    # ins_num 127 / 1: the widest (C = 128 = DMNERF_MAX_LOGITS, four logit blocks) and the narrowest object-code head
    for N, S, ins_num, seed in ((8, 64, 13, 5), (3, 21, 13, 6), (4, 32, 59, 7), (5, 40, 93, 8), (3, 20, 127, 9), (2, 17, 1, 10)):
        sd = O.make_weights(seed, ins_num, gain=1.7)
        g = torch.Generator().manual_seed(seed)
        rays_o = torch.randn(N, 3, generator=g)
        rays_d = torch.randn(N, 3, generator=g)
        z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0]
        cot = torch.randn(N, S, 4 + ins_num + 1, generator=g)
        raw_want, want = _oracle_mlp_grads(sd, rays_o, rays_d, z, cot)
        m = model_from(A, sd, ins_num)
        raw = A.G.run_network_train(m, rays_o.cuda(), rays_d.cuda(), z.cuda())
        tclose(raw, raw_want, "raw (training forward)", rel=1e-5)
        (raw * cot.cuda()).sum().backward()
        worst = {}
        for k, p in m.named_parameters():
            assert p.grad is not None, k
            worst[k] = tclose(p.grad, want[k], f"grad {k} (N={N},S={S},ins={ins_num})")
        # inference kernel and training forward agree bit for bit
        with torch.no_grad():
            assert torch.equal(A.R.run_network(m, rays_o.cuda(), rays_d.cuda(), z.cuda()), raw.detach())


def test_ins_branch_gradient_barrier(A):
    """h.detach() (dm_nerf.py:95): an ins-only loss reaches only the three ins layers."""
    sd = O.make_weights(9, 13, gain=1.7)
    m = model_from(A, sd, 13)
    g = torch.Generator().manual_seed(9)
    ro, rd = torch.randn(4, 3, generator=g).cuda(), torch.randn(4, 3, generator=g).cuda()
    z = torch.sort(torch.rand(4, 32, generator=g) * 5 + 1, -1)[0].cuda()
    raw = A.G.run_network_train(m, ro, rd, z)
    raw[..., 4:].square().sum().backward()
    ins_layers = ("ins_feature_linear", "ins_feature_linears.0", "ins_linear")
    for k, p in m.named_parameters():
        nz = float(p.grad.abs().max()) > 0
        assert nz == k.startswith(ins_layers), (k, nz)


def _loss_from(out, cts):
    return ((out['rgb_fine'] * cts[0]).sum() + (out['rgb_coarse'] * cts[1]).sum() + (out['ins_fine'] * cts[2]).sum()
            + (out['ins_coarse'] * cts[3]).sum() + (out['raw_fine'][..., 4:] * cts[4]).sum()
            + (out['raw_coarse'][..., 4:] * cts[5]).sum())


def test_dm_nerf_training_grads_vs_oracle(A, capsys):
    ins_num, N = 13, 48
    sd_c = O.make_weights(31, ins_num, gain=1.7, sigma_bias=0.3)
    sd_f = O.make_weights(32, ins_num, gain=1.7, sigma_bias=0.3)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(25.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(3).choice(480 * 640, N, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous()
    g = torch.Generator().manual_seed(33)
    t_rand, u = torch.rand(N, 64, generator=g), torch.rand(N, 128, generator=g)
    C = ins_num + 1
    cts = [torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, ins_num, generator=g),
           torch.randn(N, ins_num, generator=g), 0.01 * torch.randn(N, 192, C, generator=g), 0.01 * torch.randn(N, 64, C, generator=g)]
    mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None)
    out = A.R.dm_nerf(rays.cuda(), None, None, mc, mf, z.cuda(), args, t_rand=t_rand.cuda(), u=u.cuda())
    assert out['rgb_fine'].requires_grad and out['raw_fine'].requires_grad
    _loss_from(out, [c.cuda() for c in cts]).backward()
    # oracle on the SAME fine depths (the inverse-CDF step is ill-conditioned; it is tested on its own)
    sdc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    sdf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    want = O.dm_nerf(rays, sdc, sdf, z, perturb=1.0, t_rand=t_rand, u=u, z_fine_override=out['z_vals_fine'].detach().cpu())
    _loss_from(want, cts).backward()
    for k in ('rgb_fine', 'rgb_coarse', 'ins_fine', 'ins_coarse', 'depth_fine', 'raw_coarse'):
        tclose(out[k], want[k], k, rel=2e-5)
    # The f64 REFEREE.  48 rays x (64 + 192) samples push ~10^7 ReLU units through the backward; a unit
    # within float noise of zero takes a different side in f32 than in exact arithmetic, so ANY f32 implementation -- the
    # reference's included -- sits ~1e-2 of a tensor's scale away from the f64 gradient.  The claim that can be tested is
    # therefore relative: the HIP gradient is as close to the f64 gradient as the f32 oracle (= the reference's arithmetic) is,
    #     err(HIP, f64) <= 1.5 x err(oracle f32, f64) + 3e-7 x scale        per parameter tensor, max norm and l2 norm
    # (the floor is f32 rounding of the tensors whose error is at that level).  Observed: the two error columns agree to three
    # digits on 55 of 60 tensors (the f32 forward is the same fmaf chain, so both take the same side of every ReLU).
    sdc64 = {k: v.double().clone().requires_grad_(True) for k, v in sd_c.items()}
    sdf64 = {k: v.double().clone().requires_grad_(True) for k, v in sd_f.items()}
    want64 = O.dm_nerf(rays.double(), sdc64, sdf64, z.double(), perturb=1.0, t_rand=t_rand.double(), u=u.double(),
                       z_fine_override=out['z_vals_fine'].detach().cpu().double())
    _loss_from(want64, [c.double() for c in cts]).backward()
    table = []
    for m_, sd32, sd64, tag in ((mc, sdc, sdc64, "coarse"), (mf, sdf, sdf64, "fine")):
        for k, p in m_.named_parameters():
            truth = sd64[k].grad
            scale, nrm = float(truth.abs().max()), float(truth.norm())
            d_hip, d_o32 = p.grad.cpu().double() - truth, sd32[k].grad.double() - truth
            e = (float(d_hip.abs().max()), float(d_o32.abs().max()), float(d_hip.norm()), float(d_o32.norm()))
            table.append((tag, k, scale, e[0] / scale, e[1] / scale, e[2] / nrm, e[3] / nrm))
            assert e[0] <= 1.5 * e[1] + 3e-7 * scale, (tag, k, e, scale)
            assert e[2] <= 1.5 * e[3] + 3e-7 * nrm, (tag, k, e, nrm)
    with capsys.disabled():
        print("\n[f64 referee, 48 rays] per tensor: |HIP - f64| and |oracle f32 - f64|, max norm / scale  (l2 / l2)")
        for tag, k, scale, a_, b_, c_, d_ in table:
            print(f"  {tag:6s} {k:30s} scale {scale:.2e}   HIP {a_:.2e}  oracle-f32 {b_:.2e}   ({c_:.2e}  {d_:.2e})")
        worst = max(table, key=lambda t: t[3])
        print(f"  worst HIP column entry {worst[3]:.2e} at {worst[0]}.{worst[1]} (oracle f32 there: {worst[4]:.2e}); "
              f"sum of the HIP column {sum(t[3] for t in table):.4e}, of the oracle-f32 column {sum(t[4] for t in table):.4e}")
    # the fine-level loss gives the coarse model nothing beyond its own terms: zero the coarse cotangents
    mc.zero_grad(); mf.zero_grad()
    out = A.R.dm_nerf(rays.cuda(), None, None, mc, mf, z.cuda(), args, t_rand=t_rand.cuda(), u=u.cuda())
    (out['rgb_fine'] * cts[0].cuda()).sum().backward()
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in mc.parameters())   # z_samples.detach() (render.py:68)


def test_large_training_batch_equals_sum_of_chunks(A):
    """Maximum sizes on the training side: one dm_nerf + backward over 12 001 rays (2.3 M fine samples: beyond the
    1 048 576 samples one training launch addresses, so the fine network runs as three launches behind the same call;
    N is not a multiple of the 32-sample tile) gives the gradients of the same rays taken 4096 at a time -- the loss is
    a sum over rays, so the chunk gradients add up (f32 summation order aside)."""
    ins_num, N = 13, 12001
    sd_c = O.make_weights(91, ins_num, gain=1.7, sigma_bias=0.3)
    sd_f = O.make_weights(92, ins_num, gain=1.7, sigma_bias=0.3)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(55.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(13).choice(480 * 640, N, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0).cuda()
    z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous().cuda()
    g = torch.Generator().manual_seed(93)
    c_rgb, c_ins = torch.randn(N, 3, generator=g).cuda(), torch.randn(N, ins_num, generator=g).cuda()
    args = types.SimpleNamespace(perturb=0.0, N_importance=128, is_train=True, N_ins=None)

    def grads(chunks):
        mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
        for s0, n in chunks:
            sl = slice(s0, s0 + n)
            out = A.R.dm_nerf(rays[:, sl].contiguous(), None, None, mc, mf, z[sl].contiguous(), args)
            ((out['rgb_fine'] * c_rgb[sl]).sum() + (out['rgb_coarse'] * c_rgb[sl]).sum()
             + (out['ins_fine'] * c_ins[sl]).sum() + (out['ins_coarse'] * c_ins[sl]).sum()).backward()
            del out
        return torch.cat([p.grad.reshape(-1) for m in (mc, mf) for p in m.parameters()]).double().cpu()

    whole = grads([(0, N)])
    parts = grads([(0, 4096), (4096, 4096), (8192, N - 8192)])
    assert bool(torch.isfinite(whole).all()) and float(whole.abs().max()) > 0
    rel = float((whole - parts).norm() / parts.norm())
    assert rel <= 1e-5, rel
    assert float((whole - parts).abs().max()) <= 1e-4 * float(parts.abs().max())


def test_adam_steps_follow_the_oracle_trajectory(A):
    """Three optimiser steps of the reference recipe (Adam lr 5e-4, img2mse on both levels + an ins term)."""
    ins_num, N = 13, 32
    sd_c = O.make_weights(41, ins_num, gain=1.7, sigma_bias=0.3)
    sd_f = O.make_weights(42, ins_num, gain=1.7, sigma_bias=0.3)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(25.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(4).choice(480 * 640, N, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous()
    g = torch.Generator().manual_seed(43)
    target = torch.rand(N, 3, generator=g)
    tgt_ins = torch.rand(N, ins_num, generator=g)

    def loss_fn(out, tc, ti):
        return ((out['rgb_fine'] - tc) ** 2).mean() + ((out['rgb_coarse'] - tc) ** 2).mean() \
            + ((out['ins_fine'] - ti) ** 2).mean() + ((out['ins_coarse'] - ti) ** 2).mean()

    mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    args = types.SimpleNamespace(perturb=0.0, N_importance=128, is_train=True, N_ins=None)
    sdc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    sdf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    opt_o = torch.optim.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4, betas=(0.9, 0.999))
    got, want = [], []
    for it in range(3):
        out = A.R.dm_nerf(rays.cuda(), None, None, mc, mf, z.cuda(), args)
        loss = loss_fn(out, target.cuda(), tgt_ins.cuda())
        opt.zero_grad(); loss.backward(); opt.step()
        got.append(float(loss.detach()))
        o = O.dm_nerf(rays, sdc, sdf, z, perturb=0.)
        lo = loss_fn(o, target, tgt_ins)
        opt_o.zero_grad(); lo.backward(); opt_o.step()
        want.append(float(lo.detach()))
    assert np.allclose(got, want, rtol=2e-3), (got, want)
    assert got[2] < got[0]


def test_full_training_recipe_follows_the_oracle(A):
    """The complete loss of train_dmsr.py:33-64 -- img2mse + Hungarian-matched ins_criterion + ins_penalizer on both
    levels -- for three Adam steps: device kernels vs the oracle (scipy assignment, PyTorch autograd)."""
    from dm_nerf_amd.networks import evaluator as E
    from dm_nerf_amd.networks import penalizer as P
    ins_num, N = 13, 48
    sd_c = O.make_weights(51, ins_num, gain=1.7, sigma_bias=0.3)
    sd_f = O.make_weights(52, ins_num, gain=1.7, sigma_bias=0.3)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(40.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(5).choice(480 * 640, N, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous()
    g = torch.Generator().manual_seed(53)
    target = torch.rand(N, 3, generator=g)
    labels = torch.randint(0, 6, (N,), generator=g)
    tol, dw = 0.05, 0.05                                            # configs/dmsr/train/study.txt

    def loss_dev(out, rays_d):
        a = types.SimpleNamespace(tolerance=tol, deta_w=dw)
        return E.img2mse(out['rgb_fine'], target.cuda()) + E.img2mse(out['rgb_coarse'], target.cuda()) \
            + E.ins_criterion(out['ins_fine'], labels.cuda(), ins_num)[0] + E.ins_criterion(out['ins_coarse'], labels.cuda(), ins_num)[0] \
            + P.ins_penalizer(out['raw_fine'], out['z_vals_fine'], out['depth_fine'], rays_d, a).sum() \
            + P.ins_penalizer(out['raw_coarse'], out['z_vals_coarse'], out['depth_coarse'], rays_d, a).sum()

    def loss_ora(o):
        return ((o['rgb_fine'] - target) ** 2).mean() + ((o['rgb_coarse'] - target) ** 2).mean() \
            + O.ins_criterion(o['ins_fine'], labels, ins_num)[0].sum() + O.ins_criterion(o['ins_coarse'], labels, ins_num)[0].sum() \
            + O.ins_penalizer(o['raw_fine'], o['z_vals_fine'], o['depth_fine'], rays[1], tol, dw).sum() \
            + O.ins_penalizer(o['raw_coarse'], o['z_vals_coarse'], o['depth_coarse'], rays[1], tol, dw).sum()

    mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    args = types.SimpleNamespace(perturb=0.0, N_importance=128, is_train=True, N_ins=None)
    sdc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    sdf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    opt_o = torch.optim.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4, betas=(0.9, 0.999))
    got, want = [], []
    for it in range(3):
        out = A.R.dm_nerf(rays.cuda(), None, None, mc, mf, z.cuda(), args)
        loss = loss_dev(out, rays[1].cuda())
        opt.zero_grad(); loss.backward(); opt.step()
        got.append(float(loss.detach()))
        lo = loss_ora(O.dm_nerf(rays, sdc, sdf, z, perturb=0.))
        opt_o.zero_grad(); lo.backward(); opt_o.step()
        want.append(float(lo.detach()))
    assert np.allclose(got, want, rtol=2e-3), (got, want)


def test_penalizer_golden_value_and_gradient(A, golden):
    """fused emptiness penalizer vs the vectors produced by the reference's own ins_penalizer."""
    from dm_nerf_amd.networks import penalizer as P
    g = golden("penalizer")
    for k in ("S64_C14", "S192_C14", "S192_C60"):
        a = types.SimpleNamespace(tolerance=float(g[f"{k}_tol"]), deta_w=0.05)
        raw = g[f"{k}_raw"].cuda().requires_grad_(True)
        loss = P.ins_penalizer(raw, g[f"{k}_z"].cuda(), g[f"{k}_depth"].cuda(), g[f"{k}_d"].cuda(), a)
        assert loss.shape == g[f"{k}_loss"].shape
        assert abs(float(loss.detach()) - float(g[f"{k}_loss"])) <= 2e-6 * abs(float(g[f"{k}_loss"])) + 1e-7
        grad, = torch.autograd.grad(loss.sum() * 1.5, raw)
        assert float(grad[..., :4].abs().max()) == 0.0
        tclose(grad[..., 4:] / 1.5, g[f"{k}_grad"], f"penalizer grad {k}", rel=2e-5)


@pytest.mark.parametrize("mode", ["fuse_heads", "mfma_split", "mfma_split=f16x2"])
def test_opt_in_fused_heads_training(A, mode):
    """``args.fuse_heads`` / ``args.mfma_split`` (True = "bf16x3", or "f16x2") in training mode (the split forwards are the fused-heads
    function on the 16-bit MFMA, f32-class; they write the same f32 workspace as the f32 forward).  For ``fuse_heads``: the forward runs on the fused-heads blob (the two activation-free feature linears
    folded into the hidden layers), the backward is the same re-associated one.  Forward inside the MLP contract against the
    oracle (NOT bit-equal to the default path: that is why it is opt-in); per-tensor gradients against PyTorch autograd of the
    oracle; a dm_nerf step through the flag moves the loss like the default step."""
    key, _, val = mode.partition("=")
    split = {"mfma_split": "bf16x3", "mfma_split=f16x2": "f16x2"}.get(mode)
    for ins_num, seed, N, S in ((13, 15, 8, 64), (59, 16, 5, 33)):
        sd = O.make_weights(seed, ins_num, gain=1.7)
        g = torch.Generator().manual_seed(seed)
        rays_o, rays_d = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
        z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0]
        cot = torch.randn(N, S, 4 + ins_num + 1, generator=g)
        raw_want, want = _oracle_mlp_grads(sd, rays_o, rays_d, z, cot)
        m = model_from(A, sd, ins_num)
        raw = A.G.run_network_train(m, rays_o.cuda(), rays_d.cuda(), z.cuda(), fused=mode == "fuse_heads", split=split)
        tclose(raw, raw_want, "raw (fused training forward)", rel=1e-5)
        (raw * cot.cuda()).sum().backward()
        for k, p in m.named_parameters():
            tclose(p.grad, want[k], f"grad {k} (fused forward, ins={ins_num})")
        with torch.no_grad():
            layerwise = A.R.run_network(m, rays_o.cuda(), rays_d.cuda(), z.cuda())
        assert float((raw.detach() - layerwise).abs().max()) <= 1e-5 * (1 + float(layerwise.abs().max()))
    # through dm_nerf
    ins_num, N = 13, 64
    sd_c, sd_f = O.make_weights(61, ins_num, gain=1.7, sigma_bias=0.3), O.make_weights(62, ins_num, gain=1.7, sigma_bias=0.3)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(25.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(6).choice(480 * 640, N, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0).cuda()
    z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous().cuda()
    target = torch.rand(N, 3, generator=torch.Generator().manual_seed(7)).cuda()
    losses = {}
    for fused in (False, True):
        mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
        opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
        args = types.SimpleNamespace(perturb=0.0, N_importance=128, is_train=True, N_ins=None, **{key: ((val or True) if fused else False)})
        tr = []
        for _ in range(3):
            out = A.R.dm_nerf(rays, None, None, mc, mf, z, args)
            loss = ((out['rgb_fine'] - target) ** 2).mean() + ((out['rgb_coarse'] - target) ** 2).mean() + out['ins_fine'].square().mean()
            opt.zero_grad(); loss.backward(); opt.step()
            tr.append(float(loss.detach()))
        losses[fused] = tr
    # (f16x2: twice the f32 kernels' rounding error -> a few more ReLU-boundary mask flips in 16 k samples x 3 Adam steps, see
    # test_split_backward_kernels_vs_oracle_autograd; Adam turns each into an update of size lr)
    assert np.allclose(losses[True], losses[False], rtol=(1e-3 if mode.endswith("f16x2") else 1e-4)) and losses[True][2] < losses[True][0], losses


@pytest.mark.parametrize("flavour", ["bf16x3", "f16x2"])
def test_split_dgrad_kernel_vs_f32_dgrad_kernel(A, flavour):
    """The opt-in split data-gradient kernels (csrc/mlp_bwd_split.hip: bf16x3; csrc/mlp_bwd_f16.hip: f16x2) against the default f32 one on the SAME saved
    forward and the same dL/draw: every dy tensor of the workspace (31 weight quarters' worth of products, masks applied) and
    the transposed d raw.  f32-class: per tensor  max|diff| <= 1e-5 * max|f32 result|  (six bf16 products per f32 product,
    f32 accumulation; the d raw copy is bit-equal).  Ragged M (tail block) and every logit-block count (OBI 1..3)."""
    from dm_nerf_amd import _lib
    lib = _lib.load()
    for ins_num, N, S, seed in ((13, 9, 64, 31), (59, 4, 33, 32), (93, 3, 70, 33), (1, 2, 17, 34), (120, 2, 40, 35)):
        sd = O.make_weights(seed, ins_num, gain=1.7)
        m = model_from(A, sd, ins_num)
        g = torch.Generator().manual_seed(seed)
        rays_o, rays_d = torch.randn(N, 3, generator=g).cuda(), torch.randn(N, 3, generator=g).cuda()
        z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0].cuda()
        M_, C = N * S, ins_num + 1
        graw = torch.randn(M_, 4 + C, generator=g).cuda()
        raw = torch.empty(N, S, 4 + C, device="cuda")
        save = torch.empty(lib.dmnerf_train_save_floats(M_), device="cuda")
        _lib.check(lib.dmnerf_mlp_fwd_rays_train(_lib.ptr(m.blob()), ins_num, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z), N, S,
                                                 _lib.ptr(raw), _lib.ptr(save), _lib.stream()), "fwd")
        Mp = A.G._row_len(M_)
        outs = []
        for split in (False, True):
            dsave = torch.full_like(save, float("nan"))
            gt = torch.full((Mp // 32, 4 + C, 32), float("nan"), device="cuda")
            if split:
                fn, blob_t = (lib.dmnerf_mlp_bwd_data_split, m.blob_t_split()) if flavour == "bf16x3" else (lib.dmnerf_mlp_bwd_data_f16, m.blob_t_f16())
                extra = (None,) if flavour == "f16x2" else ()               # (no gradient scaling: the kernels are compared on the same dL/draw)
                _lib.check(fn(_lib.ptr(blob_t), ins_num, _lib.ptr(save), _lib.ptr(graw), M_,
                              _lib.ptr(dsave), _lib.ptr(gt), *extra, _lib.stream()), "bwd split")
            else:
                _lib.check(lib.dmnerf_mlp_bwd_data(_lib.ptr(m.blob()), _lib.ptr(m.blob_t()), ins_num, _lib.ptr(save), _lib.ptr(graw), M_,
                                                   _lib.ptr(dsave), _lib.ptr(gt), _lib.stream()), "bwd")
            torch.cuda.synchronize()
            outs.append((dsave.cpu(), gt.cpu()))
        (d0, g0), (d1, g1) = outs
        assert torch.equal(torch.isnan(d0), torch.isnan(d1)), "the two kernels write different parts of the workspace"
        assert torch.equal(torch.nan_to_num(g0), torch.nan_to_num(g1)), "d raw (transposed copy) differs"
        a, b = torch.nan_to_num(d0), torch.nan_to_num(d1)
        assert float(a.abs().max()) > 0
        # per written region of 256 Mp (or less) floats: relative to that region's own scale
        written = (~torch.isnan(d0)).nonzero().flatten()
        lo, hi = int(written.min()), int(written.max()) + 1
        step = 128 * Mp
        for s in range(lo, hi, step):
            x, y = a[s:s + step], b[s:s + step]
            scale = float(x.abs().max())
            assert float((x - y).abs().max()) <= 1e-5 * scale + 1e-12, (flavour, ins_num, s // step, float((x - y).abs().max()), scale)


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_split_backward_kernels_vs_oracle_autograd(A, mode, capsys):
    """The opt-in split modes' BACKWARD against the oracle directly (not only against their f32 twins): forward + dgrad + wgrad of
    the mode on the MLP alone (run_network_train), per-parameter gradients vs PyTorch autograd of the oracle's mlp_forward on the
    same inputs and the same cotangent, ins_num 13 and 93, ragged sample counts.  Tolerance: the default path's
    (max|diff| <= 2e-4 max|want| per tensor); the observed worst ratio is printed.  The third case scales the cotangent by 1e-7
    -- the magnitude of dL/draw in a real step (a mean over thousands of rays) --: the f16x2 backward runs on 2^s dL/draw
    (dmnerf_grad_scale) and must be as accurate there as at O(1).

    ReLU's derivative is discontinuous: a pre-activation within rounding distance of zero can get a different mask bit from two
    f32-class forwards, which moves single gradient entries by O(dy) -- for ANY two implementations that are not bitwise equal
    (measured for f16x2 against the f32 kernels: about one such bit per 2 M activations, i.e. in every other batch of 384 samples;
    1 in 5.5 M at 2368 samples for one seed).  To keep the comparison about arithmetic, a case draws its inputs from the first of
    a few seeds for which the mode's forward saves exactly the ReLU masks the default f32 forward saves (checked here)."""
    from dm_nerf_amd import _lib
    lib = _lib.load()
    worst, used = {}, {}
    for ins_num, seed0, N, S, cs in ((13, 71, 6, 64, 1.0), (93, 172, 5, 50, 1.0), (13, 273, 7, 33, 1e-7)):
        for seed in range(seed0, seed0 + 8):
            sd = O.make_weights(seed, ins_num, gain=1.7)
            g = torch.Generator().manual_seed(seed)
            rays_o, rays_d = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
            z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0]
            cot = torch.randn(N, S, 4 + ins_num + 1, generator=g) * cs
            m = model_from(A, sd, ins_num)
            M_ = N * S
            Mp = A.G._row_len(M_)
            ro_d, rd_d, z_d = rays_o.cuda(), rays_d.cuda(), z.cuda()          # kept alive: the C ABI takes raw pointers
            bits = []
            for fn, blob in ((lib.dmnerf_mlp_fwd_rays_train, m.blob()),
                             (lib.dmnerf_mlp_fwd_rays_train_split, m.blob_split()) if mode == "bf16x3" else (lib.dmnerf_mlp_fwd_rays_train_f16, m.blob_f16())):
                raw_ = torch.empty(N, S, 4 + ins_num + 1, device="cuda")
                save_ = torch.empty(lib.dmnerf_train_save_floats(M_), device="cuda")
                _lib.check(fn(_lib.ptr(blob), ins_num, _lib.ptr(ro_d), _lib.ptr(rd_d), _lib.ptr(z_d), N, S,
                              _lib.ptr(raw_), _lib.ptr(save_), _lib.stream()), "train forward")
                bits.append(save_[(63 + 27 + 8 * 256 + 128 + 128) * Mp:].view(torch.int32).cpu())
            if not torch.equal(bits[0], bits[1]):
                continue
            # ... and for which the ORACLE's masks are those of the f32 kernels too: the default path meets the tolerance
            raw_want, want = _oracle_mlp_grads(sd, rays_o, rays_d, z, cot)
            md = model_from(A, sd, ins_num)
            (A.G.run_network_train(md, rays_o.cuda(), rays_d.cuda(), z.cuda()) * cot.cuda()).sum().backward()
            if all(float((p.grad.cpu() - want[k]).abs().max()) <= 2e-4 * float(want[k].abs().max()) + 1e-7 * cs for k, p in md.named_parameters()):
                break
        else:
            raise AssertionError(f"{mode}: no seed in {seed0}..{seed0 + 7} without a ReLU-boundary mask difference")
        used[(ins_num, cs)] = seed
        raw = A.G.run_network_train(m, rays_o.cuda(), rays_d.cuda(), z.cuda(), split=mode)
        tclose(raw, raw_want, f"raw ({mode} training forward)", rel=1e-5)
        (raw * cot.cuda()).sum().backward()
        for k, p in m.named_parameters():
            scale = float(want[k].abs().max())
            err = float((p.grad.cpu() - want[k]).abs().max())
            assert err <= 2e-4 * scale + 1e-7 * cs, (mode, ins_num, cs, k, err, scale)
            if scale > 0:
                worst[(ins_num, cs)] = max(worst.get((ins_num, cs), 0.0), err / scale)
    with capsys.disabled():
        print(f"\n[{mode} backward vs oracle autograd] worst per-tensor max|diff| / max|want|: {worst}  (seeds {used})")


def test_split_wgrad_kernel_vs_f32_wgrad_kernel(A):
    """The opt-in split weight-gradient kernels (csrc/wgrad_split.hip: bf16x3; csrc/wgrad_f16.hip: f16x2) against the default f32 one on the SAME saved
    forward and the same dy (f32 dgrad): every parameter gradient, per tensor  max|diff| <= 2e-5 * max|f32 result|  (both sum
    ~10^3 .. 10^4 products per entry in f32; six bf16 products per f32 product).  Both plans (f32- and split-balanced) are valid
    for both kernels; ragged M and every logit-block count."""
    from dm_nerf_amd import _lib
    lib = _lib.load()
    for ins_num, N, S, seed in ((13, 33, 64, 41), (59, 9, 33, 42), (93, 5, 70, 43), (1, 2, 17, 44), (120, 3, 40, 45)):      # logit blocks 1, 2, 3, 1, 4
        sd = O.make_weights(seed, ins_num, gain=1.7)
        m = model_from(A, sd, ins_num)
        g = torch.Generator().manual_seed(seed)
        rays_o, rays_d = torch.randn(N, 3, generator=g).cuda(), torch.randn(N, 3, generator=g).cuda()
        z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0].cuda()
        M_, C = N * S, ins_num + 1
        graw = torch.randn(M_, 4 + C, generator=g).cuda()
        raw = torch.empty(N, S, 4 + C, device="cuda")
        save = torch.empty(lib.dmnerf_train_save_floats(M_), device="cuda")
        _lib.check(lib.dmnerf_mlp_fwd_rays_train(_lib.ptr(m.blob()), ins_num, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z), N, S,
                                                 _lib.ptr(raw), _lib.ptr(save), _lib.stream()), "fwd")
        Mp = A.G._row_len(M_)
        dsave = torch.empty_like(save)
        gt = torch.empty(Mp // 32, 4 + C, 32, device="cuda")
        _lib.check(lib.dmnerf_mlp_bwd_data(_lib.ptr(m.blob()), _lib.ptr(m.blob_t()), ins_num, _lib.ptr(save), _lib.ptr(graw), M_,
                                           _lib.ptr(dsave), _lib.ptr(gt), _lib.stream()), "bwd")
        flat = m.flat()
        grads = {}
        for kern in ("f32", "split", "f16"):
            for plan in ("f32", "split"):
                jobs, n_jobs, outs, n_outs, pf = A.G.wgrad_plan(ins_num, M_, raw.device, split=plan == "split")
                part = torch.full((pf,), float("nan"), device="cuda")
                out = torch.full((lib.dmnerf_param_count(ins_num),), float("nan"), device="cuda")
                fn = {"f32": lib.dmnerf_mlp_bwd_weights, "split": lib.dmnerf_mlp_bwd_weights_split, "f16": lib.dmnerf_mlp_bwd_weights_f16}[kern]
                extra = (None,) if kern == "f16" else ()                  # (no gradient scale: same operands for every kernel)
                _lib.check(fn(_lib.ptr(save), _lib.ptr(dsave), _lib.ptr(gt), M_, _lib.ptr(jobs), n_jobs, _lib.ptr(outs), n_outs,
                              _lib.ptr(flat), ins_num, _lib.ptr(part), _lib.ptr(out), *extra, _lib.stream()), kern)
                torch.cuda.synchronize()
                assert not bool(torch.isnan(out).any()), (kern, plan)
                grads[kern, plan] = A.G.split_flat_grads(m, out)
        names = [k for k, _ in m.named_parameters()]
        for kern in ("split", "f16"):
            for plan in ("f32", "split"):
                for k, a, b in zip(names, grads["f32", "f32"], grads[kern, plan]):
                    scale = float(a.abs().max())
                    err = float((a - b).abs().max())
                    assert err <= 2e-5 * scale + 1e-9, (ins_num, kern, plan, k, err, scale)
        for k, a, b in zip(names, grads["f32", "f32"], grads["f32", "split"]):      # the f32 kernel on the other plan: slice order only
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-9, (ins_num, k)
