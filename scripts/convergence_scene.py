"""GPU box: does the whole recipe LEARN on this path -- colours AND object codes?

Trains a default-init DM-NeRF pair on the analytic scene of oracle/analytic_scene.py (four spheres on a ground disc, exact
images and labels) with the reference's loop (train_dmsr.py:24-64: one random view per step, N_train random pixels of it,
img2mse + Hungarian-matched ins_criterion + emptiness penalizer on both levels, Adam 5e-4 with the reference's decay,
perturb = 1) and reports, at checkpoints, the held-out view's PSNR and permutation-invariant label purity, and the loss terms.

    python scripts/convergence_scene.py --steps 5000 --batch 3072 [--mode f16x2] [--out gpurun_out/conv.json]
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
from oracle import analytic_scene as S  # noqa: E402  (scripts/ are diagnostics: the scene and its metrics are test infrastructure)

INS_NUM = 13


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=3072)
    ap.add_argument("--H", type=int, default=60)
    ap.add_argument("--W", type=int, default=80)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--mode", default="", help="'' (default f32 kernels) | bf16x3 | f16x2")
    ap.add_argument("--every", type=int, default=1000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from dm_nerf_amd import config as Cfg, distributed as D
    from dm_nerf_amd.networks import helpers as Hh, render as R
    dev = torch.device("cuda:0")
    H, W = a.H, a.W
    thetas = list(np.linspace(0.0, 360.0, a.views, endpoint=False)) + [17.0]           # the last view is held out
    poses, ims, labs = S.make_views(H, W, thetas, INS_NUM)
    K = S.dmsr_intrinsics(H, W)
    d_ims, d_labs, d_poses = ims.to(dev), labs.to(dev), poses.to(dev)
    torch.manual_seed(0)
    cargs = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=8, netwidth=256, ins_num=INS_NUM, device=dev)
    _, _, mc, mf, _ = Cfg.create_nerf(cargs)
    mc.train(); mf.train()
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05,
                                 mfma_split=a.mode or False)
    eargs = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None, mfma_split=a.mode or False)
    z = Hh.z_val_sample(a.batch, S.NEAR, S.FAR, 64, device=dev)
    ze = Hh.z_val_sample(H * W, S.NEAR, S.FAR, 64, device=dev)
    ro_t, rd_t = Hh.get_rays_k(H, W, K, d_poses[-1, :3, :4])
    test_rays = torch.stack([ro_t.reshape(-1, 3), rd_t.reshape(-1, 3)])

    def evaluate():
        mc.eval(); mf.eval()
        with torch.no_grad():
            out = R.dm_nerf(test_rays, None, None, mc, mf, ze, eargs)
        mc.train(); mf.train()
        rgb, lab = out['rgb_fine'].cpu(), out['ins_fine'].cpu().argmax(-1)
        conf = out['ins_fine'].cpu().max(-1).values
        # (the reference labels a pixel "empty" when no object channel is confident; evaluator.py ins_eval works on argmax only)
        return {"psnr_db": S.psnr(rgb, ims[-1].reshape(-1, 3)), "purity": S.purity(lab, labs[-1].reshape(-1)),
                "channels_used": int(len(torch.unique(lab))), "mean_conf": float(conf.mean())}

    np.random.seed(0)
    torch.manual_seed(0); torch.cuda.manual_seed(0)
    hist = [dict(step=0, **evaluate())]
    print(json.dumps(hist[-1]), flush=True)
    t0 = time.time()
    win = []
    for it in range(1, a.steps + 1):
        v = np.random.choice(a.views)
        tc, ti, rays = Hh.get_select_full(d_ims[v], poses[v, :3, :4], K, d_labs[v], a.batch)      # (pose on the host: no sync)
        loss, _ = D.sharded_train_step(rays, z, tc, ti, (mc, mf), args, opt, INS_NUM)
        for g in opt.param_groups:
            g['lr'] = 5e-4 * (0.1 ** (it / 500000.0))                                    # train_dmsr.py:68-72
        win.append(loss)
        if it % a.every == 0 or it == a.steps:
            torch.cuda.synchronize()
            hist.append(dict(step=it, loss=float(torch.stack(win).mean()), seconds=time.time() - t0, **evaluate()))
            win = []
            print(json.dumps(hist[-1]), flush=True)
    res = {"scene": f"oracle/analytic_scene.py: {S.N_OBJECTS} objects, {a.views} training views + 1 held-out of {H}x{W}", "mode": a.mode or "default",
           "recipe": f"{a.steps} steps x {a.batch} rays, train_dmsr.py loop", "history": hist}
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
