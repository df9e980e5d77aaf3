"""GPU tests (-m gpu): does the recipe LEARN on this path, colours and object codes, and does it learn what the oracle learns?

The scene is analytic (oracle/analytic_scene.py: four spheres on a ground disc, exact images and labels from ray / primitive
intersection), so there is a right answer without a dataset:

* ``test_object_branch_learns_on_the_analytic_scene`` -- the reference's loop (train_dmsr.py:24-64: one random view per step,
  N_train = 3072 random pixels, img2mse + Hungarian-matched ins_criterion + emptiness penalizer on both levels, Adam with the
  reference's decay, perturb = 1) for 3000 steps: held-out PSNR, permutation-invariant label purity (untrained: 0.39), the
  number of object channels in use, and every loss term falling.
* ``test_training_trajectory_follows_the_oracle`` -- the first 300 steps at 512 rays on exactly the batches and jitter the CPU
  oracle was run on in the build container (tests/golden/make_train_traj.py -> train_traj.npz): per-term losses step by step,
  and the held-out PSNR / purity after them, for the default kernels and both opt-in split modes.
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import analytic_scene as S

pytestmark = pytest.mark.gpu
INS_NUM = 13


def _setup(H, W, views, dev):
    from dm_nerf_amd.networks import helpers as Hh
    thetas = list(np.linspace(0.0, 360.0, views, endpoint=False)) + [17.0]           # the last view is held out
    poses, ims, labs = S.make_views(H, W, thetas, INS_NUM)
    K = S.dmsr_intrinsics(H, W)
    ro, rd = Hh.get_rays_k(H, W, K, poses[-1, :3, :4])
    test_rays = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)])
    return poses, ims, labs, K, test_rays


def _evaluate(mc, mf, test_rays, ze, im, lab, mode):
    from dm_nerf_amd.networks import render as R
    eargs = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None, mfma_split=mode or False)
    mc.eval(); mf.eval()
    with torch.no_grad():
        out = R.dm_nerf(test_rays, None, None, mc, mf, ze, eargs)
    mc.train(); mf.train()
    pred = out['ins_fine'].cpu().argmax(-1)
    return S.psnr(out['rgb_fine'].cpu(), im.reshape(-1, 3)), S.purity(pred, lab.reshape(-1)), int(len(torch.unique(pred)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", [None, "f16x2"])
def test_object_branch_learns_on_the_analytic_scene(mode, capsys):
    """3000 steps x 3072 rays of the shipped loop on the analytic scene (12 training views of 120 x 160 + one held out).
    Measured (profiles/r03/convergence_*.json): PSNR 7.0 -> 29.6 .. 30.0 dB, purity 0.39 -> 0.99, six channels in use (five
    objects + "nothing"), for the f32 kernels and the f16x2 mode alike; the asserted bounds leave the run-to-run room of an
    Adam trajectory at lr 5e-4 (+- 0.5 dB between checkpoints)."""
    from dm_nerf_amd import config as Cfg, distributed as D
    from dm_nerf_amd.networks import evaluator as E, helpers as Hh, penalizer as P, render as R
    dev = torch.device("cuda:0")
    H, W, views, steps, batch = 120, 160, 12, 3000, 3072
    poses, ims, labs, K, test_rays = _setup(H, W, views, dev)
    d_ims, d_labs = ims.to(dev), labs.to(dev)
    torch.manual_seed(0)
    cargs = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=8, netwidth=256, ins_num=INS_NUM, device=dev)
    _, _, mc, mf, _ = Cfg.create_nerf(cargs)
    mc.train(); mf.train()
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05,
                                 mfma_split=mode or False)
    z = Hh.z_val_sample(batch, S.NEAR, S.FAR, 64, device=dev)
    ze = Hh.z_val_sample(H * W, S.NEAR, S.FAR, 64, device=dev)
    psnr0, pur0, _ = _evaluate(mc, mf, test_rays, ze, ims[-1], labs[-1], mode)
    np.random.seed(0)
    torch.cuda.manual_seed(0)
    terms = []                                                       # per step: rgb fine, rgb coarse, ins fine, ins coarse (device tensors)
    for it in range(1, steps + 1):
        v = np.random.choice(views)
        tc, ti, rays = Hh.get_select_full(d_ims[v], poses[v, :3, :4], K, d_labs[v], batch)
        out = R.dm_nerf(rays, None, None, mc, mf, z, args)
        t = [E.img2mse(out['rgb_fine'], tc), E.img2mse(out['rgb_coarse'], tc),
             E.ins_criterion(out['ins_fine'], ti, INS_NUM)[0], E.ins_criterion(out['ins_coarse'], ti, INS_NUM)[0]]
        loss = sum(t) + P.ins_penalizer(out['raw_fine'], out['z_vals_fine'], out['depth_fine'], rays[1], args).sum() \
            + P.ins_penalizer(out['raw_coarse'], out['z_vals_coarse'], out['depth_coarse'], rays[1], args).sum()
        opt.zero_grad(); loss.backward(); opt.step()
        for g in opt.param_groups:
            g['lr'] = 5e-4 * (0.1 ** (it / 500000.0))                  # train_dmsr.py:68-72
        terms.append(torch.stack([x.detach() for x in t]))
    tr = torch.stack(terms).cpu().numpy()
    psnr1, pur1, used = _evaluate(mc, mf, test_rays, ze, ims[-1], labs[-1], mode)
    first, last = tr[:100].mean(0), tr[-100:].mean(0)
    with capsys.disabled():
        print(f"\n[analytic scene, {mode or 'default'}] PSNR {psnr0:.2f} -> {psnr1:.2f} dB, purity {pur0:.3f} -> {pur1:.3f}, {used} channels; "
              f"loss terms (rgb f/c, ins f/c) first 100 steps {np.round(first, 4).tolist()} -> last 100 {np.round(last, 4).tolist()}")
    assert np.isfinite(tr).all()
    assert psnr0 < 10.0 and pur0 < 0.5                                # the untrained networks know nothing
    assert psnr1 >= 28.0, psnr1
    assert pur1 >= 0.97 and used == S.N_OBJECTS + 1, (pur1, used)
    assert (last < 0.25 * first).all(), (first, last)                 # every term fell by more than 4x


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", [None, "bf16x3", "f16x2"])
def test_training_trajectory_follows_the_oracle(mode, golden, capsys):
    """300 steps x 512 rays on the batches and jitter of the oracle's run (same seeds -> same numpy / CPU-generator draws).
    A training trajectory amplifies rounding differences (ReLU boundaries, Adam's sign-like first steps), so the comparison is
    step by step where it is tight and in windows afterwards.  Measured (r03, printed below): total loss within 1.7e-4 of the
    oracle's over the first 20 steps for all three modes, 4e-4 .. 5e-3 over the first 100 (the largest single step of a window is
    spiky: the same mode moved from 1.7e-3 to 5.0e-3 when only the summation order of its weight-gradient plan changed), 1.2 .. 1.6 %
    on 20-step windows over all 300; after them the held-out PSNR is 20.87 / 21.16 / 21.13 dB (default / bf16x3 / f16x2) against the
    oracle's 21.21 while still climbing ~0.03 dB per step -- the three GPU modes differ from the oracle no more than from each
    other.  The bounds are ~2.5-3x those figures."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_train_traj as T
    from dm_nerf_amd.networks import dm_nerf as M, evaluator as E, helpers as Hh, penalizer as P, render as R
    g = golden("train_traj")
    assert [int(v) for v in g["config"]] == [T.INS_NUM, T.H, T.W, T.VIEWS, T.STEPS, T.BATCH]
    dev = torch.device("cuda:0")
    poses, ims, labs, K, test_rays = _setup(T.H, T.W, T.VIEWS, dev)
    rays_v = []
    for p in poses:
        ro, rd = Hh.get_rays_k(T.H, T.W, K, p[:3, :4])
        rays_v.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)]))
    d_ims, d_labs = ims.to(dev), labs.to(dev)
    sd_c, sd_f = T.start_weights()
    mc, mf = M.DM_NeRF(8, 256, 63, 27, [4], T.INS_NUM), M.DM_NeRF(8, 256, 63, 27, [4], T.INS_NUM)
    mc.load_state_dict(sd_c); mf.load_state_dict(sd_f)
    mc, mf = mc.to(dev).train(), mf.to(dev).train()
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, tolerance=T.TOL, deta_w=T.DW, mfma_split=mode or False)
    z = Hh.z_val_sample(T.BATCH, S.NEAR, S.FAR, 64, device=dev)
    ze = Hh.z_val_sample(T.H * T.W, S.NEAR, S.FAR, 64, device=dev)
    psnr0, pur0, _ = _evaluate(mc, mf, test_rays, ze, ims[-1], labs[-1], mode)
    rows = []
    for it, (v, idx, t_rand, u) in enumerate(T.draws(), 1):
        idx = idx.to(dev)
        rays = rays_v[v][:, idx]
        tc, ti = d_ims[v].reshape(-1, 3)[idx], d_labs[v].reshape(-1)[idx]
        out = R.dm_nerf(rays.contiguous(), None, None, mc, mf, z, args, t_rand=t_rand.to(dev), u=u.to(dev))
        t = [E.img2mse(out['rgb_fine'], tc), E.img2mse(out['rgb_coarse'], tc),
             E.ins_criterion(out['ins_fine'], ti, T.INS_NUM)[0], E.ins_criterion(out['ins_coarse'], ti, T.INS_NUM)[0],
             P.ins_penalizer(out['raw_fine'], out['z_vals_fine'], out['depth_fine'], rays[1], args).sum(),
             P.ins_penalizer(out['raw_coarse'], out['z_vals_coarse'], out['depth_coarse'], rays[1], args).sum()]
        loss = sum(t)
        opt.zero_grad(); loss.backward(); opt.step()
        for grp in opt.param_groups:
            grp['lr'] = 5e-4 * (0.1 ** (it / 500000.0))
        rows.append(torch.stack([loss.detach()] + [x.detach() for x in t]))
    got = torch.stack(rows).double().cpu().numpy()
    want = g["losses"].numpy() if torch.is_tensor(g["losses"]) else np.asarray(g["losses"])
    psnr1, pur1, _ = _evaluate(mc, mf, test_rays, ze, ims[-1], labs[-1], mode)
    rel = np.abs(got[:, 0] - want[:, 0]) / np.abs(want[:, 0])
    win = lambda a: a[:, 0].reshape(-1, 20).mean(1)
    relw = np.abs(win(got) - win(want)) / np.abs(win(want))
    gp, gpu_ = [float(x) for x in g["psnr"]], [float(x) for x in g["purity"]]
    with capsys.disabled():
        print(f"\n[trajectory vs oracle, {mode or 'default'}] total loss: max rel gap first 20 steps {rel[:20].max():.2e}, first 100 {rel[:100].max():.2e}, "
              f"all 300 {rel.max():.2e}; 20-step windows {relw.max():.2e};  PSNR {psnr1:.3f} dB (oracle {gp[1]:.3f}, start {gp[0]:.3f}), "
              f"purity {pur1:.4f} (oracle {gpu_[1]:.4f})")
    assert abs(psnr0 - gp[0]) <= 0.01 and abs(pur0 - gpu_[0]) <= 0.005        # same start
    assert rel[:20].max() <= 5e-4, rel[:20].max()                       # step by step while rounding has not been amplified yet
    assert rel[:100].max() <= 1.5e-2, rel[:100].max()
    assert relw.max() <= 0.04, relw.max()                               # 20-step windows over the whole run
    assert abs(psnr1 - gp[1]) <= 0.8 and abs(pur1 - gpu_[1]) <= 0.02, (psnr1, gp, pur1, gpu_)
    assert psnr1 >= gp[0] + 8.0                                         # ... and it learned: + 9.5 dB in 300 steps
