"""GPU experiment (NOT product code; docs/EXPERIMENTS.md "Round 5"): the coarse level's losses and network backward do not depend
on the fine forward (z_samples.detach(), render.py:68) -- run them on the side stream UNDER the fine forward, whose last round at
small batches leaves most CUs idle (384 rays: 576 workgroups = 2.25 rounds of 256 CUs).  Same kernels, same sums: bit-equal
gradients.  Both orders use the drop-in per-level loss functions (the fused two-level tail cannot be split), so the comparison is
like for like; the product's fused-tail step is printed beside them.
    python scripts/exp_early_coarse.py [rays ...]"""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [384, 512, 1024]
    dev = torch.device("cuda", 0)
    from dm_nerf_amd import autograd as G, distributed as D
    from dm_nerf_amd.networks import evaluator as E, helpers as H, penalizer as P
    from dm_nerf_amd.optim import FlatAdam
    from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical
    K = dmsr_intrinsics(B.H_IMG, B.W_IMG)
    ro, rd = H.get_rays_k(B.H_IMG, B.W_IMG, K, pose_spherical(30.0, -65.0, 7.0).to(dev), row0=0, nrows=8)
    args = types.SimpleNamespace(perturb=1.0, N_importance=B.N_IMP, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05)
    out = {}
    for n in sizes:
        rays = torch.stack([ro.reshape(-1, 3)[:n], rd.reshape(-1, 3)[:n]])
        rays_o, rays_d = rays[0].contiguous(), rays[1].contiguous()
        z = H.z_val_sample(n, B.NEAR, B.FAR, B.S_COARSE, device=dev)
        g = torch.Generator(device=dev).manual_seed(0)
        target = torch.rand(n, 3, device=dev, generator=g)
        labels = torch.randint(0, 9, (n,), device=dev, generator=g)
        consts = G.pen_consts(args)

        def composite(raw, zz):
            rgb, w, depth, ins, part = G.CompositePenFunction.apply(raw, zz, rays_d, consts)
            depth._dmn_pen = (part, raw.data_ptr(), zz.data_ptr(), rays_d.data_ptr(), consts)
            return rgb, w, depth, ins

        def level_loss(raw, zz, rgb, depth, ins):
            return E.img2mse(rgb, target) + E.ins_criterion(ins[..., :], labels, B.INS_NUM)[0] + P.ins_penalizer(raw, zz, depth, rays_d, args).sum()

        def step(mc, mf, opt, early):
            opt.zero_grad()
            t_rand = torch.rand(z.shape, device=dev)
            u = torch.rand([n, B.N_IMP], device=dev)
            z_c = H.stratify(z, t_rand)
            raw_c = G.run_network_train(mc, rays_o, rays_d, z_c)
            rgb_c, w_c, depth_c, ins_c = composite(raw_c, z_c)
            if early:
                loss_c = level_loss(raw_c, z_c, rgb_c, depth_c, ins_c)
                with G.overlapped_backward(True, join_on_exit=False):
                    loss_c.backward()                                   # coarse dgrad + wgrad: side stream, left in flight
            with torch.no_grad():
                z_f = H.importance_resample(z_c, w_c.detach(), B.N_IMP, det=False, u=u)
            raw_f = G.run_network_train(mf, rays_o, rays_d, z_f)
            rgb_f, _, depth_f, ins_f = composite(raw_f, z_f)
            loss_f = level_loss(raw_f, z_f, rgb_f, depth_f, ins_f)
            if early:
                with G.overlapped_backward(True):
                    loss_f.backward()                                   # fine: main stream beside the side stream, then join
                loss = loss_f.detach() + loss_c.detach()
            else:
                loss = loss_f + level_loss(raw_c, z_c, rgb_c, depth_c, ins_c)
                with G.overlapped_backward(D.overlap_enabled(n, B.S_COARSE, B.S_COARSE + B.N_IMP, dev)):
                    loss.backward()
            opt.step()
            return loss.detach()

        res = {}
        grads = {}
        for name, early in (("drop_in_losses_standard_order", False), ("drop_in_losses_early_coarse_backward", True)):
            pe, ve, mc, mf = B.build_models(dev)
            mc.train(); mf.train()
            opt = FlatAdam((mc, mf), lr=5e-4)
            torch.manual_seed(0); torch.cuda.manual_seed(0)
            step(mc, mf, opt, early)
            torch.cuda.synchronize()
            grads[name] = opt.arena.flat.clone()
            one = lambda: step(mc, mf, opt, early)
            B.warm_up(one)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                one()
            torch.cuda.synchronize()
            res[name] = round((time.perf_counter() - t0) / 40 * 1e3, 3)
        res["first_step_gradients_bit_equal"] = bool(torch.equal(*grads.values()))
        # the product's step (fused two-level tail, one backward pass) for scale
        pe, ve, mc, mf = B.build_models(dev)
        mc.train(); mf.train()
        opt = FlatAdam((mc, mf), lr=5e-4)
        one = lambda: D.sharded_train_step(rays, z, target, labels, (mc, mf), args, opt, B.INS_NUM)
        B.warm_up(one)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            one()
        torch.cuda.synchronize()
        res["product_fused_tail"] = round((time.perf_counter() - t0) / 40 * 1e3, 3)
        out[f"n{n}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
