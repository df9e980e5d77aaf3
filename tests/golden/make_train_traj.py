"""Build-container script (CPU, ~1 h alone on 8 cores; ~2 h next to other work): the ORACLE's training trajectory on the analytic scene, for
tests/test_gpu_convergence.py::test_training_trajectory_follows_the_oracle (the first STEPS = 300 steps, loss by loss) and
::test_trained_psnr_matches_the_oracle_at_the_plateau (all LONG_STEPS = 2000 steps, held-out PSNR at EVAL_AT).

    python tests/golden/make_train_traj.py            -> tests/golden/train_traj.npz

The full recipe of train_dmsr.py:24-64 on oracle/ref_cpu.py (PyTorch autograd, scipy assignment, torch Adam with the
reference's lr decay): STEPS steps x BATCH rays of the analytic scene (oracle/analytic_scene.py), one random view per step.
Everything random is drawn from seeds that the GPU test re-creates bit for bit (numpy RandomState for the view / pixel
selection, a CPU torch Generator for the jitter), so the fixture holds only results: the seven loss terms of every step, and the
held-out view's PSNR / label purity at steps 0 and EVAL_AT.  Draws are sequential: the first STEPS steps of the long run ARE the
300-step run of earlier rounds (its losses / PSNR regenerate bit for bit).

Round 5 -- the oracle's OWN chaos floor (tests/test_gpu_convergence.py compares two samples, HIP runs against oracle runs):

    TRAJ_VARIANT=k TRAJ_THREADS=t python tests/golden/make_train_traj.py   -> tests/golden/train_traj_v{k}.npz

repeats the identical training with the rays of every batch visited in another order (a permutation of the batch rows drawn
from RandomState(1000 + k); each ray keeps its pixel and its jitter).  Every loss is a mean / sum over the rays of the batch, so
the run is the same mathematics with the f32 sums -- the loss means, the cost-matrix sums, the K = 98 304-row weight-gradient
contractions -- taken in another order, which is all that separates two correct f32 implementations.  Only the held-out
PSNR / purity at EVAL_AT are kept for the variants."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import analytic_scene as S, ref_cpu as O  # noqa: E402

INS_NUM, H, W, VIEWS, STEPS, BATCH = 13, 60, 80, 12, 300, 512
LONG_STEPS, EVAL_AT = 2000, (300, 1000, 1500, 2000)
TOL, DW = 0.05, 0.05
THETAS = list(np.linspace(0.0, 360.0, VIEWS, endpoint=False)) + [17.0]         # the last view is held out


def draws(steps=STEPS, variant=0):
    """The batch selection and jitter of every step, generated lazily: (view, pixel index [BATCH], t_rand [BATCH,64], u [BATCH,128]).
    variant k > 0: the same rays with the same jitter, rows permuted (summation-order variant, see the module docstring)."""
    rs = np.random.RandomState(0)
    gen = torch.Generator().manual_seed(0)
    prs = np.random.RandomState(1000 + variant) if variant else None
    for _ in range(steps):
        v = int(rs.choice(VIEWS))
        idx = torch.from_numpy(rs.choice(H * W, BATCH, replace=False))
        t_rand, u = torch.rand(BATCH, 64, generator=gen), torch.rand(BATCH, 128, generator=gen)
        if prs is not None:
            perm = torch.from_numpy(prs.permutation(BATCH))
            idx, t_rand, u = idx[perm], t_rand[perm].contiguous(), u[perm].contiguous()
        yield v, idx, t_rand, u


def start_weights():
    return O.make_weights(903, INS_NUM), O.make_weights(904, INS_NUM)


def main():
    variant = int(os.environ.get("TRAJ_VARIANT", "0"))
    torch.set_num_threads(max(1, min(16, int(os.environ.get("TRAJ_THREADS", os.cpu_count() or 1)))))
    poses, ims, labs = S.make_views(H, W, THETAS, INS_NUM)
    K = S.dmsr_intrinsics(H, W)
    rays_v = []
    for p in poses:
        ro, rd = O.get_rays_k(H, W, K, p)
        rays_v.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0).contiguous())
    sd_c, sd_f = start_weights()
    sdc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    sdf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    opt = torch.optim.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4, betas=(0.9, 0.999))
    z = O.z_val_sample(BATCH, S.NEAR, S.FAR, 64).contiguous()
    ze = O.z_val_sample(H * W, S.NEAR, S.FAR, 64).contiguous()

    def evaluate():
        with torch.no_grad():
            o = O.dm_nerf(rays_v[-1], {k: v.detach() for k, v in sdc.items()}, {k: v.detach() for k, v in sdf.items()}, ze, perturb=0.)
        return S.psnr(o['rgb_fine'], ims[-1].reshape(-1, 3)), S.purity(o['ins_fine'].argmax(-1), labs[-1].reshape(-1))

    psnr0, pur0 = evaluate()
    print(f"step 0: PSNR {psnr0:.3f} dB, purity {pur0:.4f}", flush=True)
    losses = np.zeros((LONG_STEPS, 7), dtype=np.float64)
    evals = [(0, psnr0, pur0)]
    t0 = time.time()
    for it, (v, idx, t_rand, u) in enumerate(draws(LONG_STEPS, variant), 1):
        rays = rays_v[v][:, idx]
        tc, ti = ims[v].reshape(-1, 3)[idx], labs[v].reshape(-1)[idx]
        o = O.dm_nerf(rays, sdc, sdf, z, perturb=1.0, t_rand=t_rand, u=u)
        terms = [((o['rgb_fine'] - tc) ** 2).mean(), ((o['rgb_coarse'] - tc) ** 2).mean(),
                 O.ins_criterion(o['ins_fine'], ti, INS_NUM)[0].sum(), O.ins_criterion(o['ins_coarse'], ti, INS_NUM)[0].sum(),
                 O.ins_penalizer(o['raw_fine'], o['z_vals_fine'], o['depth_fine'], rays[1], TOL, DW).sum(),
                 O.ins_penalizer(o['raw_coarse'], o['z_vals_coarse'], o['depth_coarse'], rays[1], TOL, DW).sum()]
        loss = sum(terms)
        opt.zero_grad(); loss.backward(); opt.step()
        for g in opt.param_groups:
            g['lr'] = 5e-4 * (0.1 ** (it / 500000.0))                           # train_dmsr.py:68-72
        losses[it - 1] = [float(loss.detach())] + [float(t.detach()) for t in terms]
        if it % 20 == 0:
            print(f"step {it}: loss {losses[it - 1, 0]:.4f}  ({time.time() - t0:.0f} s)", flush=True)
        if it in EVAL_AT:
            evals.append((it,) + evaluate())
            print(f"step {it}: PSNR {evals[-1][1]:.3f} dB, purity {evals[-1][2]:.4f}", flush=True)
    at = {e[0]: e for e in evals}
    here = os.path.dirname(os.path.abspath(__file__))
    if variant:
        np.savez(os.path.join(here, f"train_traj_v{variant}.npz"), eval_steps=np.array([e[0] for e in evals], dtype=np.int64),
                 eval_psnr=np.array([e[1] for e in evals]), eval_purity=np.array([e[2] for e in evals]),
                 loss_windows=losses[:, 0].reshape(-1, 20).mean(1), threads=np.int64(torch.get_num_threads()),
                 config=np.array([INS_NUM, H, W, VIEWS, STEPS, BATCH], dtype=np.int64))
        return
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_traj.npz"), losses=losses,
             psnr=np.array([psnr0, at[STEPS][1]]), purity=np.array([pur0, at[STEPS][2]]),
             eval_steps=np.array([e[0] for e in evals], dtype=np.int64), eval_psnr=np.array([e[1] for e in evals]),
             eval_purity=np.array([e[2] for e in evals]),
             config=np.array([INS_NUM, H, W, VIEWS, STEPS, BATCH], dtype=np.int64))


if __name__ == "__main__":
    main()
