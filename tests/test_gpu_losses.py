"""The training step's loss tail as one autograd node (dm_nerf_amd/losses.py, csrc/losses.hip; extension) against the drop-in
functions the reference's loop calls one by one (train_dmsr.py:33-61: img2mse, ins_criterion, ins_penalizer on both levels):
the same total to f32 rounding (the tail adds the terms in the association order of train_dmsr.py:47-58), the same six terms, and the SAME gradients for everything the tail touches -- rgb, ins and raw of both levels --
bit for bit (same criterion / penalizer kernels, the squared-error gradient formed with autograd's own products)."""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
INS = 13


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from dm_nerf_amd import _lib, distributed as D, losses as L
    from dm_nerf_amd.networks import dm_nerf as M, evaluator as E, helpers as H, penalizer as P, render as R
    _lib.load()
    return types.SimpleNamespace(M=M, H=H, R=R, D=D, E=E, P=P, L=L)


def _scene(A, N, ins_num=INS):
    models = []
    for seed in (91, 92):
        m = A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
        m.load_state_dict(O.make_weights(seed, ins_num, gain=1.7, sigma_bias=0.3))
        models.append(m.cuda().train())
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(40.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(4).choice(480 * 640, N, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]]).cuda()
    z = A.H.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    g = torch.Generator().manual_seed(6)
    return models, rays, z, torch.rand(N, 3, generator=g).cuda(), g


LEAVES = ('rgb_fine', 'rgb_coarse', 'ins_fine', 'ins_coarse', 'raw_fine', 'raw_coarse')


@pytest.mark.parametrize("penalize,n_ins,ins_num,ref_pen", [(True, None, 13, "separate"), (True, None, 13, "drop_in"), (False, None, 13, "separate"),
                                                             (True, 70, 13, "separate"), (True, None, 59, "separate")])
def test_fused_tail_equals_the_drop_in_losses(A, penalize, n_ins, ins_num, ref_pen):
    """``ref_pen``: how the REFERENCE side evaluates the emptiness term -- "separate": ``emptiness_penalizer`` on plain tensors, i.e.
    the stand-alone forward / backward kernels (dmnerf_penalizer_fwd / _bwd) and autograd's add of the two d raw; "drop_in":
    ``ins_penalizer`` on the dict's own tensors, which finds the per-ray sums the fused compositing pass left behind and lets that
    pass's backward kernel add the gradient.  The fused tail must equal both, bit for bit."""
    N = 200
    (mc, mf), rays, z, target, g = _scene(A, N, ins_num)
    labels = torch.randint(0, 7, (n_ins or N,), generator=g).cuda()
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=penalize,
                                 tolerance=0.05 if penalize else None, deta_w=0.05 if penalize else None)
    torch.cuda.manual_seed(3)
    out = A.R.dm_nerf(rays, None, None, mc, mf, z, args)
    leaves = [out[k] for k in LEAVES]
    cut = (lambda t: t[-n_ins:]) if n_ins is not None else (lambda t: t)

    # the reference's sequence on the drop-in functions
    terms_ref = []
    loss_ref = 0.
    for lvl in ("fine", "coarse"):
        t = [A.E.img2mse(out['rgb_' + lvl], target), A.E.ins_criterion(cut(out['ins_' + lvl]), labels, ins_num)[0]]
        loss_ref = loss_ref + t[0] + t[1]
        if penalize and ref_pen == "drop_in":
            t.append(A.P.ins_penalizer(out['raw_' + lvl], out['z_vals_' + lvl], out['depth_' + lvl], rays[1], args).sum())
            loss_ref = loss_ref + t[2]
        elif penalize:
            t.append(A.P.emptiness_penalizer(out['raw_' + lvl], out['z_vals_' + lvl], out['depth_' + lvl][..., None].detach(), rays[1],
                                             args.tolerance, args.deta_w).sum())
            loss_ref = loss_ref + t[2]
        else:
            t.append(torch.zeros((), device="cuda"))
        terms_ref += t
    grads_ref = torch.autograd.grad(loss_ref, leaves if penalize else leaves[:4], retain_graph=True)

    total, terms = A.L.train_losses(out, rays[1], target, labels, ins_num, args,
                                    rgb_ins=(out['rgb_fine'], out['rgb_coarse'], cut(out['ins_fine']), cut(out['ins_coarse'])))
    grads = torch.autograd.grad(total, leaves if penalize else leaves[:4], retain_graph=True)
    assert terms.shape == (6,) and not terms.requires_grad and total.requires_grad
    if penalize:                                                 # the step penalises: the compositing pass carried the per-ray sums
        from dm_nerf_amd import autograd as G
        assert G.pen_partials(out['depth_fine'], out['raw_fine'], out['z_vals_fine'], rays[1], G.pen_consts(args)) is not None
    for i, (a, b) in enumerate(zip(terms.tolist(), [float(t.detach()) for t in terms_ref])):
        if i % 3 == 0:                                           # squared error: summed in double here, f32 tree in ATen
            assert abs(a - b) <= 2e-7 * abs(b), (i, a, b)
        else:                                                    # same kernels, same arithmetic
            assert a == b, (i, a, b)
    assert abs(float(total) - float(loss_ref)) <= 1e-6 * abs(float(loss_ref))
    for k, a, b in zip(LEAVES, grads, grads_ref):
        assert torch.equal(a, b), (k, float((a - b).abs().max()))
    assert float(grads[0].abs().max()) > 0 and float(grads[2].abs().max()) > 0
    # ... and a scaled upstream gradient reaches every term
    g3 = torch.autograd.grad(total * 3.0, leaves[:4], retain_graph=True)
    g3_ref = torch.autograd.grad(loss_ref * 3.0, leaves[:4])
    for a, b in zip(g3, g3_ref):
        assert torch.equal(a, b)


def test_fused_tail_reports_label_conditions_like_the_criterion(A):
    N = 96
    (mc, mf), rays, z, target, g = _scene(A, N)
    args = types.SimpleNamespace(perturb=0.0, N_importance=128, is_train=True, N_ins=None, penalize=False)
    out = A.R.dm_nerf(rays, None, None, mc, mf, z, args)
    bad = torch.randint(0, 5, (N,), generator=g)
    bad[7] = INS + 3                                            # outside [0, ins_num]: the reference's one_hot would raise
    with pytest.raises(ValueError, match="outside"):
        A.L.train_losses(out, rays[1], target, bad.cuda(), INS, args, check=True)
    total, _ = A.L.train_losses(out, rays[1], target, bad.cuda(), INS, args)       # default: no sync, no raise
    assert torch.isfinite(total)
    many = torch.arange(N) % (INS + 1)                          # 14 distinct labels for 13 channels
    with pytest.raises(ValueError, match="distinct labels"):
        A.L.train_losses(out, rays[1], target, many.cuda(), INS, args, check=True)


def test_step_with_the_fused_tail_equals_the_step_without(A, monkeypatch):
    """sharded_train_step uses the fused tail by default; DMNERF_FUSED_TAIL=0 runs the reference's sequence of drop-in functions.
    Two Adam steps each way from the same start and jitter: the same parameters bit for bit (same gradients, see above)."""
    N = 160
    res = []
    for fused in ("1", "0"):
        monkeypatch.setenv("DMNERF_FUSED_TAIL", fused)
        (mc, mf), rays, z, target, g = _scene(A, N)
        labels = torch.randint(0, 7, (N,), generator=g).cuda()
        args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05)
        opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
        torch.cuda.manual_seed(8)
        losses = [float(A.D.sharded_train_step(rays, z, target, labels, (mc, mf), args, opt, INS)[0]) for _ in range(2)]
        res.append((losses, [p.detach().clone() for m in (mc, mf) for p in m.parameters()]))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-6)
    assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
