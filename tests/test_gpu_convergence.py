"""GPU tests (-m gpu): does the recipe LEARN on this path, colours and object codes, and does it learn what the oracle learns?

The scene is analytic (oracle/analytic_scene.py: four spheres on a ground disc, exact images and labels from ray / primitive
intersection), so there is a right answer without a dataset:

* ``test_object_branch_learns_on_the_analytic_scene`` -- the reference's loop (train_dmsr.py:24-64: one random view per step,
  N_train = 3072 random pixels, img2mse + Hungarian-matched ins_criterion + emptiness penalizer on both levels, Adam with the
  reference's decay, perturb = 1) for 3000 steps: held-out PSNR, permutation-invariant label purity (untrained: 0.39), the
  number of object channels in use, and every loss term falling.
  Then the TRAINED networks are rendered by the CPU oracle too: on real learned surfaces the two renderers give the same image
  (PSNR vs ground truth within 0.05 dB -- north_star's bound --, <= 1e-3 of the pixels change label, PSNR(HIP, oracle) >= 80 dB).
* ``test_training_trajectory_follows_the_oracle`` -- the first 300 steps at 512 rays on exactly the batches and jitter the CPU
  oracle was run on in the build container (tests/golden/make_train_traj.py -> train_traj.npz): per-term losses step by step,
  and the held-out PSNR / purity after them, for the default kernels and both opt-in split modes.  A training trajectory is
  chaotic, so the PSNR bound is not a guess: the same training is repeated on the HIP path with only the SUMMATION ORDER of the
  weight-gradient split changed (autograd.PLAN_MAX_WGS), and the allowed gap to the oracle is 1.5 x the spread of those runs
  (never below 0.05 dB).
* ``test_trained_psnr_matches_the_oracle_at_the_plateau`` -- the same run continued to 2000 steps.  From ~1000 steps on the
  held-out PSNR no longer climbs steadily (oracle: 24.79 / 26.79 / 25.14 dB at 1000 / 1500 / 2000 -- a 512-ray Adam run at lr 5e-4
  moves a dB between checkpoints), so the comparison is the MEAN over those three checkpoints, bounded the same way.
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
    _evaluate.last = out
    return S.psnr(out['rgb_fine'].cpu(), im.reshape(-1, 3)), S.purity(pred, lab.reshape(-1)), int(len(torch.unique(pred)))


def _oracle_render(mc, mf, test_rays, ze, chunk=4800):
    """The held-out view through oracle/ref_cpu.dm_nerf on the networks' CURRENT weights (host cores; 19 200 rays ~ 1 min)."""
    from oracle import ref_cpu as O
    sd_c = {k: v.detach().cpu() for k, v in mc.state_dict().items()}
    sd_f = {k: v.detach().cpu() for k, v in mf.state_dict().items()}
    rays, z = test_rays.cpu(), ze.cpu()
    rgb, ins = [], []
    with torch.no_grad():
        for s in range(0, rays.shape[1], chunk):
            o = O.dm_nerf(rays[:, s:s + chunk].contiguous(), sd_c, sd_f, z[s:s + chunk].contiguous(), perturb=0.)
            rgb.append(o['rgb_fine']); ins.append(o['ins_fine'])
    return torch.cat(rgb), torch.cat(ins)


@pytest.mark.timeout(900)
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
    # The trained regime, rendered by BOTH paths (tester.py:86-88 is where the reference measures PSNR): real learned surfaces --
    # peaked weights, saturated sigmoids, confident labels -- instead of scaled random weights.
    hip_rgb, hip_ins = _evaluate.last['rgb_fine'].cpu(), _evaluate.last['ins_fine'].cpu()
    gt = ims[-1].reshape(-1, 3)
    # (the oracle renders a subset of the image rows -- every third, 6400 of the 19 200 pixels: ~15 s of host time instead of 50; the
    # full view was compared in rounds 3-5: 0 / 19 200 flips, profiles/r05.  The opt-in mode's trained networks are not re-rendered
    # by the oracle any more: its inference parity has its own tests, test_gpu_parity.py / test_gpu_configs.py.)
    if mode is not None:
        return
    rows = torch.arange(0, H, 3)[:, None] * W + torch.arange(W)[None, :]
    sel = rows.reshape(-1)
    or_rgb, or_ins = _oracle_render(mc, mf, test_rays[:, sel.to(test_rays.device)], ze[:sel.numel()])
    hip_rgb, hip_ins, gt = hip_rgb[sel], hip_ins[sel], gt[sel]
    psnr1 = S.psnr(hip_rgb, gt)                                       # (PSNR of the same pixels on both sides)
    psnr_or = S.psnr(or_rgb, gt)
    flips = float((hip_ins.argmax(-1) != or_ins.argmax(-1)).float().mean())
    agree = S.psnr(hip_rgb, or_rgb)
    with capsys.disabled():
        print(f"[trained regime, {mode or 'default'}] held-out PSNR vs ground truth: HIP {psnr1:.4f} dB, oracle {psnr_or:.4f} dB "
              f"(|d| = {abs(psnr1 - psnr_or):.5f}); label flips {flips:.2e} of {hip_ins.shape[0]} pixels; PSNR(HIP, oracle) {agree:.1f} dB; "
              f"max |d rgb| {float((hip_rgb - or_rgb).abs().max()):.2e}")
    assert abs(psnr1 - psnr_or) <= 0.05, (psnr1, psnr_or)             # north_star: PSNR within 0.05 dB of the reference
    assert flips <= 1e-3, flips
    assert agree >= 80.0, agree


def _run_trajectory(mode, steps, eval_at, max_wgs=None):
    """``steps`` steps x 512 rays on the batches and jitter of the oracle's run (same seeds -> same numpy / CPU-generator draws),
    held-out PSNR / purity at step 0 and at ``eval_at``.  ``max_wgs``: workgroup budget of the weight-gradient split-K plan
    (None = the default, one per CU) -- another budget sums the same products in another order, nothing else changes."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_train_traj as T
    from dm_nerf_amd import autograd as G
    from dm_nerf_amd.networks import dm_nerf as M, evaluator as E, helpers as Hh, penalizer as P, render as R
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
    evals = {0: _evaluate(mc, mf, test_rays, ze, ims[-1], labs[-1], mode)[:2]}
    rows = []
    G.PLAN_MAX_WGS = max_wgs
    try:
        for it, (v, idx, t_rand, u) in enumerate(T.draws(steps), 1):
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
            if it in eval_at:
                evals[it] = _evaluate(mc, mf, test_rays, ze, ims[-1], labs[-1], mode)[:2]
    finally:
        G.PLAN_MAX_WGS = None
    return torch.stack(rows).double().cpu().numpy(), evals


_long_runs = {}            # weight-gradient plan budget -> (per-step loss rows, {step: (PSNR, purity)}) of the 2000-step default-mode run


def _long_run(budget):
    """The 2000-step default-mode run with this plan budget, made ONCE per session: its first 300 steps ARE the 300-step run (same
    draws in the same order; the evaluation at step 300 draws nothing), so ``test_training_trajectory_follows_the_oracle[None]`` and
    ``test_trained_psnr_matches_the_oracle_at_the_plateau`` share the five runs instead of repeating their first 300 steps."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_train_traj as T
    if budget not in _long_runs:
        _long_runs[budget] = _run_trajectory(None, T.LONG_STEPS, T.EVAL_AT, max_wgs=budget)
    return _long_runs[budget]


def plan_budgets(n=5):
    """Weight-gradient plans: the default (one workgroup per CU) and ``n - 1`` other splits of the sample axis -- 7/8, 3/4, 5/8 and 1/2
    of the CU count (224, 192, 160, 128 workgroups on the 256 CUs of an MI355X); n = 8 adds 15/16, 13/16 and 11/16 (240, 208, 176);
    n = 12 adds 31/32, 27/32, 23/32 and 9/16 (248, 216, 184, 144)."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    fracs = (0.875, 0.75, 0.625, 0.5, 0.9375, 0.8125, 0.6875, 0.96875, 0.84375, 0.71875, 0.5625)[:n - 1]
    return (None,) + tuple(int(cus * f) for f in fracs)


N_PLANS_DEFAULT_MODE = 12          # HIP runs of the default mode (shared by the 300-step and the plateau test: _long_run)


def _oracle_runs(golden, steps_needed):
    """The ORACLE's own samples: the committed 2000-step run (train_traj.npz) and its summation-order variants
    (train_traj_v{k}.npz: the identical training with the rays of every batch visited in another order, so that every f32 sum --
    loss means, cost-matrix sums, the weight-gradient contractions -- is taken in another order; make_train_traj.py) ->
    {step: [held-out PSNR of each run that has all of ``steps_needed``]}, number of runs.  (A variant file may hold fewer
    checkpoints than the base run: the 300-step figures were committed while the 2000-step runs were still going.)"""
    import glob
    base = golden("train_traj")
    steps = [int(x) for x in base["eval_steps"]]
    runs = {s_: [float(p)] for s_, p in zip(steps, base["eval_psnr"]) if s_ in steps_needed}
    n = 1
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_traj_v*.npz"))):
        with np.load(path) as z:
            assert [int(x) for x in z["config"]] == [int(x) for x in base["config"]]
            have = {int(s_): float(p) for s_, p in zip(z["eval_steps"], z["eval_psnr"])}
        if all(s_ in have for s_ in steps_needed):
            n += 1
            for s_ in steps_needed:
                runs[s_].append(have[s_])
    return runs, n


def _two_sample(hip, oracle, cap):
    """Two-sample comparison of held-out PSNRs: HIP runs (weight-gradient sums in different orders) against oracle runs (batch rows
    in different orders).  Both sets are draws from 'a correct f32 implementation of this training, up to summation order':
        |mean_HIP - mean_oracle| <= 2 sqrt(s2_HIP / n_HIP + s2_oracle / n_oracle)      (Welch's standard error)
    AND <= ``cap``, an absolute ceiling that does not grow with either spread -- a path with a systematic deficit of ``cap`` or more
    fails however noisy it is.  (Deterministic on a given GPU model: the HIP runs are, the oracle's are fixtures.)"""
    hip, oracle = np.asarray(hip, dtype=np.float64), np.asarray(oracle, dtype=np.float64)
    se = float(np.sqrt(hip.var(ddof=1) / len(hip) + oracle.var(ddof=1) / len(oracle)))
    gap = abs(float(hip.mean() - oracle.mean()))
    return gap, 2.0 * se, gap <= 2.0 * se and gap <= cap


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", [None, "bf16x3", "f16x2"])
def test_training_trajectory_follows_the_oracle(mode, golden, capsys):
    """300 steps x 512 rays on the batches and jitter of the oracle's run.  A training trajectory amplifies rounding differences
    (ReLU boundaries, Adam's sign-like first steps), so the losses are compared step by step where that is tight and in windows
    afterwards, and the PSNR after the 300 steps -- still climbing ~0.03 dB per step there -- as a TWO-SAMPLE comparison (VERDICT
    r04 item 3): twelve HIP runs (five in the opt-in modes) that differ only in the order of the weight-gradient partial sums
    (plan_budgets()) against the oracle's runs -- eight: train_traj.npz and train_traj_v1 .. v7 -- that differ only in the order the
    batch rows are visited (tests/golden/train_traj*.npz).  The means must
    agree within twice the standard error of their difference AND within 0.5 dB whatever the spreads.  Loss bounds (r03
    measurements: 1.7e-4 over the first 20 steps, 4e-4 .. 5e-3 over the first 100, 1.2 .. 1.6 % on 20-step windows) are ~2.5-3x
    those figures."""
    g = golden("train_traj")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_train_traj as T
    assert [int(v) for v in g["config"]] == [T.INS_NUM, T.H, T.W, T.VIEWS, T.STEPS, T.BATCH]
    oracle_runs, n_or = _oracle_runs(golden, (T.STEPS,))
    assert n_or >= 4, "the oracle's summation-order variants (tests/golden/train_traj_v*.npz) are missing"
    if mode is None:                                                    # (shared with the plateau test: _long_run)
        runs = {b: (_long_run(b)[0][:T.STEPS], _long_run(b)[1]) for b in plan_budgets(N_PLANS_DEFAULT_MODE)}
    else:
        runs = {b: _run_trajectory(mode, T.STEPS, (T.STEPS,), max_wgs=b) for b in plan_budgets()}
    got, evals = runs[None]
    want = (g["losses"].numpy() if torch.is_tensor(g["losses"]) else np.asarray(g["losses"]))[:T.STEPS]
    (psnr0, pur0), (psnr1, pur1) = evals[0], evals[T.STEPS]
    rel = np.abs(got[:, 0] - want[:, 0]) / np.abs(want[:, 0])
    win = lambda a: a[:, 0].reshape(-1, 20).mean(1)
    relw = np.abs(win(got) - win(want)) / np.abs(win(want))
    gp, gpu_ = [float(x) for x in g["psnr"]], [float(x) for x in g["purity"]]
    psnrs = {b: r[1][T.STEPS][0] for b, r in runs.items()}
    gap, bound, ok = _two_sample(list(psnrs.values()), oracle_runs[T.STEPS], cap=0.5)
    with capsys.disabled():
        print(f"\n[trajectory vs oracle, {mode or 'default'}] total loss: max rel gap first 20 steps {rel[:20].max():.2e}, first 100 {rel[:100].max():.2e}, "
              f"all 300 {rel.max():.2e}; 20-step windows {relw.max():.2e};  purity {pur1:.4f} (oracle {gpu_[1]:.4f})")
        print(f"  two samples after {T.STEPS} steps | oracle runs (row orders): " + ", ".join(f"{v:.3f}" for v in oracle_runs[T.STEPS])
              + f" -> mean {np.mean(oracle_runs[T.STEPS]):.3f}, sd {np.std(oracle_runs[T.STEPS], ddof=1):.3f} | HIP runs (wgrad plan budgets): "
              + ", ".join(f"{b or 'default'}: {v:.3f}" for b, v in psnrs.items())
              + f" -> mean {np.mean(list(psnrs.values())):.3f}, sd {np.std(list(psnrs.values()), ddof=1):.3f} | "
              f"|difference of means| {gap:.3f} dB <= 2 SE = {bound:.3f} dB and <= 0.5 dB")
    assert abs(psnr0 - gp[0]) <= 0.01 and abs(pur0 - gpu_[0]) <= 0.005        # same start
    assert rel[:20].max() <= 5e-4, rel[:20].max()                       # step by step while rounding has not been amplified yet
    assert rel[:100].max() <= 1.5e-2, rel[:100].max()
    assert relw.max() <= 0.04, relw.max()                               # 20-step windows over the whole run
    assert ok, (gap, bound, psnrs, oracle_runs[T.STEPS])                # the two samples are one population; a 0.5 dB deficit fails regardless
    assert abs(pur1 - gpu_[1]) <= 0.02, (pur1, gpu_)
    assert psnr1 >= gp[0] + 8.0                                         # ... and it learned: + 9.5 dB in 300 steps


@pytest.mark.timeout(1500)
def test_trained_psnr_matches_the_oracle_at_the_plateau(golden, capsys):
    """The same run continued to 2000 steps (the oracle's: ~1-3 h of CPU each in the build container, make_train_traj.py): from
    ~1000 steps on the held-out PSNR has no steady slope left and wanders by a dB between checkpoints, so single checkpoints say
    little.  Statistic per run: the MEAN PSNR over steps 1000 / 1500 / 2000.  Two samples: twelve HIP runs (weight-gradient sums in
    twelve orders) against the oracle's six to eight (batch rows in as many orders): |difference of the sample means| <= 2 standard errors AND
    <= 0.75 dB whatever the spreads."""
    g = golden("train_traj")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_train_traj as T
    oracle_runs, n_or = _oracle_runs(golden, tuple(T.EVAL_AT))
    if n_or < 4:
        pytest.skip("the oracle's 2000-step summation-order variants (tests/golden/train_traj_v*.npz with checkpoints at "
                    f"{list(T.EVAL_AT)}) are not all committed: {n_or} of 4 runs")
    assert sorted(oracle_runs) == list(T.EVAL_AT) and T.EVAL_AT[-1] == T.LONG_STEPS
    late = [s for s in T.EVAL_AT if s >= 1000]
    oracle_means = [float(np.mean([oracle_runs[s][k] for s in late])) for k in range(n_or)]
    runs = {b: _long_run(b)[1] for b in plan_budgets(N_PLANS_DEFAULT_MODE)}
    means = {b: float(np.mean([ev[s][0] for s in late])) for b, ev in runs.items()}
    gap, bound, ok = _two_sample(list(means.values()), oracle_means, cap=0.75)
    with capsys.disabled():
        print(f"\n[plateau] held-out PSNR (dB) at steps {list(T.EVAL_AT)}")
        for k in range(n_or):
            print(f"  oracle run {k}: " + ", ".join(f"{oracle_runs[s][k]:.3f}" for s in T.EVAL_AT) + f" | mean of {late}: {oracle_means[k]:.3f}")
        for b, ev in runs.items():
            print(f"  HIP, wgrad plan budget {b or 'default'}: " + ", ".join(f"{ev[s][0]:.3f}" for s in T.EVAL_AT) + f" | mean of {late}: {means[b]:.3f}")
        print(f"  two samples | oracle mean {np.mean(oracle_means):.3f} sd {np.std(oracle_means, ddof=1):.3f} | HIP mean {np.mean(list(means.values())):.3f} "
              f"sd {np.std(list(means.values()), ddof=1):.3f} | |difference of means| {gap:.3f} dB <= 2 SE = {bound:.3f} dB and <= 0.75 dB; "
              f"purity at {T.LONG_STEPS}: HIP {runs[None][T.LONG_STEPS][1]:.4f}, oracle {float(g['eval_purity'][-1]):.4f}")
    assert ok, (gap, bound, means, oracle_means)
    assert abs(runs[None][T.LONG_STEPS][1] - float(g["eval_purity"][-1])) <= 0.02
