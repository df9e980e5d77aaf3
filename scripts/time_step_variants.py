"""GPU: the 384 / 512-ray optimisation step under every combination this round added -- optimizer (torch.optim.Adam plain /
capturable / fused=True, dm_nerf_amd.optim.FlatAdam), eager or one HIP graph, one or two backward streams (DMNERF_OVERLAP_BWD).
    python scripts/time_step_variants.py [steps]      -> one JSON line"""
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.graphed import GraphedTrainStep
    from dm_nerf_amd.networks import helpers as H
    from dm_nerf_amd.optim import FlatAdam
    from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical
    K = dmsr_intrinsics(B.H_IMG, B.W_IMG)
    ro, rd = H.get_rays_k(B.H_IMG, B.W_IMG, K, pose_spherical(30.0, -65.0, 7.0).to(dev), row0=0, nrows=8)
    args = types.SimpleNamespace(perturb=1.0, N_importance=B.N_IMP, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05)
    out = {}
    for n in (384, 512):
        rays = torch.stack([ro.reshape(-1, 3)[:n], rd.reshape(-1, 3)[:n]])
        z = H.z_val_sample(n, B.NEAR, B.FAR, B.S_COARSE, device=dev)
        g = torch.Generator(device=dev).manual_seed(0)
        target = torch.rand(n, 3, device=dev, generator=g)
        labels = torch.randint(0, 9, (n,), device=dev, generator=g)
        for overlap in ("1", "0"):
            os.environ["DMNERF_OVERLAP_BWD"] = overlap
            for name in ("adam", "adam_fused", "flat"):
                for graph in (False, True):
                    pe, ve, mc, mf = B.build_models(dev)
                    mc.train(); mf.train()
                    params = list(mc.parameters()) + list(mf.parameters())
                    if name == "flat":
                        opt = FlatAdam((mc, mf), lr=5e-4, capturable=graph)
                    elif name == "adam_fused":
                        opt = torch.optim.Adam(params, lr=torch.tensor(5e-4, device=dev) if graph else 5e-4, fused=True, capturable=graph)
                    else:
                        opt = torch.optim.Adam(params, lr=torch.tensor(5e-4, device=dev) if graph else 5e-4, capturable=graph)
                    if graph:
                        gs = GraphedTrainStep((mc, mf), opt, args, B.INS_NUM, rays, z, target, labels)
                        one = gs.step
                    else:
                        one = lambda: D.sharded_train_step(rays, z, target, labels, (mc, mf), args, opt, B.INS_NUM)
                    B.warm_up(one)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        one()
                    torch.cuda.synchronize()
                    out[f"n{n}_overlap{overlap}_{name}_{'graph' if graph else 'eager'}"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
    os.environ.pop("DMNERF_OVERLAP_BWD", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
