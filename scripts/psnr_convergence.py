"""PSNR that means something without a dataset (north_star: "PSNR within 0.05 dB of reference").

An analytic synthetic scene: a fixed TEACHER network pair ('trained-like' weights, oracle.PEAKY: opaque surfaces in empty
space, a varied label map) defines ground-truth colours and object labels for a few low-resolution views -- rendered by
the CPU oracle, so the targets do not depend on the code under test.  A STUDENT pair (default-init class) is then trained
for K steps with the reference's full recipe (train_dmsr.py:24-64: img2mse + Hungarian-matched ins_criterion + emptiness
penalizer on both levels, Adam lr 5e-4, perturb = 1) twice from the same start, on the same batches and the same jitter:

  * on the MI355X path (dm_nerf_amd: fused HIP forward / backward kernels, device-side criterion), and
  * on the CPU oracle (the restated reference: PyTorch autograd, scipy assignment),

and both students are evaluated on a HELD-OUT view against the teacher: PSNR, label accuracy, and the agreement of the
two students with each other.  Prints one JSON object (and writes it to --out).

    python scripts/psnr_convergence.py --steps 100 --batch 64 --out gpurun_out/psnr_convergence.json
    python scripts/psnr_convergence.py --mode mfma_split ...      (the opt-in split-bf16 training kernels)
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_cpu as O  # noqa: E402  (checker: this script is test infrastructure, like tests/)

INS_NUM, NEAR, FAR, TOL, DW = 13, 4.0, 15.0, 0.05, 0.05
torch.set_num_threads(min(16, torch.get_num_threads()))     # 256-ray batches: more threads only add synchronisation (128 threads: 3.6 s per oracle step, 16: ~0.4 s)


def psnr(a, b):
    return float(-10.0 * torch.log10(((a.double() - b.double()) ** 2).mean()))


def purity(pred, gt):
    """Permutation-invariant label quality: the Hungarian-matched loss (evaluator.py:19-74) leaves the channel <-> object
    assignment free, so predicted channel ids are compared with the teacher's labels through the best many-to-one map:
    sum over predicted channels of their largest overlap with one teacher label, / pixels."""
    tot = 0
    for c in torch.unique(pred):
        tot += int(torch.bincount(gt[pred == c]).max())
    return tot / pred.numel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--H", type=int, default=30)
    ap.add_argument("--W", type=int, default=40)
    ap.add_argument("--mode", default="default", choices=["default", "fuse_heads", "mfma_split"],
                    help="opt-in training mode of the MI355X student (args.fuse_heads / args.mfma_split)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X"
    from dm_nerf_amd.networks import dm_nerf as M, evaluator as E, helpers as Hh, penalizer as P, render as R
    H, W = a.H, a.W
    K = O.dmsr_intrinsics(H, W)
    poses = [O.pose_spherical(th, -65.0, 7.0) for th in (100.0, 110.0, 120.0, 105.0)]      # 3 training views + 1 held out
    teacher_c, teacher_f = O.make_weights(803, INS_NUM, **O.PEAKY), O.make_weights(804, INS_NUM, **O.PEAKY)
    t0 = time.time()
    rays_v, rgb_v, lab_v = [], [], []
    with torch.no_grad():
        for c2w in poses:
            ro, rd = O.get_rays_k(H, W, K, c2w)
            rays = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0).contiguous()
            out = O.dm_nerf(rays, teacher_c, teacher_f, O.z_val_sample(H * W, NEAR, FAR, 64).contiguous(), perturb=0.)
            rays_v.append(rays); rgb_v.append(out['rgb_fine']); lab_v.append(out['ins_fine'].argmax(-1))
    t_teacher = time.time() - t0
    train_rays = torch.cat(rays_v[:3], 1); train_rgb = torch.cat(rgb_v[:3], 0); train_lab = torch.cat(lab_v[:3], 0)
    test_rays, test_rgb, test_lab = rays_v[3], rgb_v[3], lab_v[3]
    n_pool = train_rays.shape[1]
    # the batches and the jitter of every step, drawn once
    rs = np.random.RandomState(0)
    gen = torch.Generator().manual_seed(0)
    steps = []
    for _ in range(a.steps):
        idx = torch.from_numpy(rs.choice(n_pool, a.batch, replace=False))
        steps.append((idx, torch.rand(a.batch, 64, generator=gen), torch.rand(a.batch, 128, generator=gen)))
    sd0_c, sd0_f = O.make_weights(903, INS_NUM), O.make_weights(904, INS_NUM)
    z = O.z_val_sample(a.batch, NEAR, FAR, 64).contiguous()
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, tolerance=TOL, deta_w=DW,
                                 fuse_heads=a.mode == "fuse_heads", mfma_split=a.mode == "mfma_split")

    # ---- student 1: the MI355X path
    mc, mf = M.DM_NeRF(8, 256, 63, 27, [4], INS_NUM), M.DM_NeRF(8, 256, 63, 27, [4], INS_NUM)
    mc.load_state_dict(sd0_c); mf.load_state_dict(sd0_f)
    mc, mf = mc.cuda().train(), mf.cuda().train()
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    zc = z.cuda()
    loss_hip = []
    torch.cuda.synchronize(); t0 = time.time()
    for idx, t_rand, u in steps:
        rays = train_rays[:, idx].cuda(); tc = train_rgb[idx].cuda(); ti = train_lab[idx].cuda()
        out = R.dm_nerf(rays, None, None, mc, mf, zc, args, t_rand=t_rand.cuda(), u=u.cuda())
        loss = E.img2mse(out['rgb_fine'], tc) + E.img2mse(out['rgb_coarse'], tc) \
            + E.ins_criterion(out['ins_fine'], ti, INS_NUM)[0] + E.ins_criterion(out['ins_coarse'], ti, INS_NUM)[0] \
            + P.ins_penalizer(out['raw_fine'], out['z_vals_fine'], out['depth_fine'], rays[1], args).sum() \
            + P.ins_penalizer(out['raw_coarse'], out['z_vals_coarse'], out['depth_coarse'], rays[1], args).sum()
        opt.zero_grad(); loss.backward(); opt.step()
        loss_hip.append(float(loss.detach()))
    torch.cuda.synchronize(); t_hip = time.time() - t0
    mc.eval(); mf.eval()
    eargs = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    with torch.no_grad():
        ev = R.dm_nerf(test_rays.cuda(), None, None, mc, mf, Hh.z_val_sample(H * W, NEAR, FAR, 64), eargs)
    hip_rgb, hip_lab = ev['rgb_fine'].cpu(), ev['ins_fine'].cpu().argmax(-1)

    # ---- student 2: the CPU oracle
    sdc = {k: v.clone().requires_grad_(True) for k, v in sd0_c.items()}
    sdf = {k: v.clone().requires_grad_(True) for k, v in sd0_f.items()}
    opt_o = torch.optim.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4, betas=(0.9, 0.999))
    loss_ora = []
    t0 = time.time()
    for idx, t_rand, u in steps:
        rays = train_rays[:, idx]; tc = train_rgb[idx]; ti = train_lab[idx]
        o = O.dm_nerf(rays, sdc, sdf, z, perturb=1.0, t_rand=t_rand, u=u)
        lo = ((o['rgb_fine'] - tc) ** 2).mean() + ((o['rgb_coarse'] - tc) ** 2).mean() \
            + O.ins_criterion(o['ins_fine'], ti, INS_NUM)[0].sum() + O.ins_criterion(o['ins_coarse'], ti, INS_NUM)[0].sum() \
            + O.ins_penalizer(o['raw_fine'], o['z_vals_fine'], o['depth_fine'], rays[1], TOL, DW).sum() \
            + O.ins_penalizer(o['raw_coarse'], o['z_vals_coarse'], o['depth_coarse'], rays[1], TOL, DW).sum()
        opt_o.zero_grad(); lo.backward(); opt_o.step()
        loss_ora.append(float(lo.detach()))
    t_ora = time.time() - t0
    with torch.no_grad():
        eo = O.dm_nerf(test_rays, {k: v.detach() for k, v in sdc.items()}, {k: v.detach() for k, v in sdf.items()},
                       O.z_val_sample(H * W, NEAR, FAR, 64).contiguous(), perturb=0.)
    ora_rgb, ora_lab = eo['rgb_fine'], eo['ins_fine'].argmax(-1)

    p_hip, p_ora = psnr(hip_rgb, test_rgb), psnr(ora_rgb, test_rgb)
    dparam = max(float((p.detach().cpu() - sdc[k].detach()).abs().max()) for k, p in mc.named_parameters())
    res = {
        "scene": f"teacher = oracle.PEAKY weights (seeds 803/804), ins_num {INS_NUM}; 3 training views + 1 held-out view of {H}x{W}, 64+128 samples; "
                 f"teacher label map: {int(len(torch.unique(test_lab)))} labels in the held-out view",
        "mode": a.mode,
        "recipe": f"{a.steps} steps x {a.batch} rays, img2mse + ins_criterion + ins_penalizer on both levels, Adam 5e-4, perturb=1, identical batches and jitter",
        "psnr_heldout_hip_db": p_hip, "psnr_heldout_oracle_db": p_ora, "abs_delta_psnr_db": abs(p_hip - p_ora),
        "psnr_untrained_db": None,
        "psnr_hip_vs_oracle_student_db": psnr(hip_rgb, ora_rgb),
        "label_purity_hip": purity(hip_lab, test_lab), "label_purity_oracle": purity(ora_lab, test_lab),
        "label_purity_untrained": None, "channels_used_hip": int(len(torch.unique(hip_lab))), "channels_used_oracle": int(len(torch.unique(ora_lab))),
        "label_agreement_hip_vs_oracle": float((hip_lab == ora_lab).float().mean()),
        "loss_first": [loss_hip[0], loss_ora[0]], "loss_last": [loss_hip[-1], loss_ora[-1]],
        "max_rel_loss_gap": float(max(abs(x - y) / abs(y) for x, y in zip(loss_hip, loss_ora))),
        "max_abs_param_gap_after_training": dparam,
        "seconds": {"teacher_render_oracle": t_teacher, "training_hip": t_hip, "training_oracle": t_ora},
    }
    with torch.no_grad():
        e0 = O.dm_nerf(test_rays, sd0_c, sd0_f, O.z_val_sample(H * W, NEAR, FAR, 64).contiguous(), perturb=0.)
    res["psnr_untrained_db"] = psnr(e0['rgb_fine'], test_rgb)
    res["label_purity_untrained"] = purity(e0['ins_fine'].argmax(-1), test_lab)
    print(json.dumps(res))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
