"""GPU: N optimisation steps at a given batch (default: the 384-ray shard of an 8-way split), for a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --stats -d out -- python scripts/shard_step.py 384 40
Prints the wall ms per step; ``DMNERF_OVERLAP_BWD`` = 0 / 1 selects one or two backward streams, ``DMNERF_FLAT_ADAM=1`` the extension
optimizer instead of torch.optim.Adam."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda", 0)
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.networks import helpers as H
    from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical
    pe, ve, mc, mf = B.build_models(dev)
    K = dmsr_intrinsics(B.H_IMG, B.W_IMG)
    ro, rd = H.get_rays_k(B.H_IMG, B.W_IMG, K, pose_spherical(30.0, -65.0, 7.0).to(dev), row0=0, nrows=8)
    rays = torch.stack([ro.reshape(-1, 3)[:n], rd.reshape(-1, 3)[:n]])
    z = H.z_val_sample(n, B.NEAR, B.FAR, B.S_COARSE, device=dev)
    mc.train(); mf.train()
    if os.environ.get("DMNERF_FLAT_ADAM") == "1":             # the extension optimizer (dm_nerf_amd.optim.FlatAdam): update + re-pack = 2 launches
        from dm_nerf_amd.optim import FlatAdam
        opt = FlatAdam((mc, mf), lr=5e-4)
    else:
        opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
    args = types.SimpleNamespace(perturb=1.0, N_importance=B.N_IMP, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05)
    g = torch.Generator(device=dev).manual_seed(0)
    target = torch.rand(n, 3, device=dev, generator=g)
    labels = torch.randint(0, 9, (n,), device=dev, generator=g)
    one = lambda: D.sharded_train_step(rays, z, target, labels, (mc, mf), args, opt, B.INS_NUM)
    B.warm_up(one)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    print(f"n={n} overlap={os.environ.get('DMNERF_OVERLAP_BWD', '1')}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")


if __name__ == "__main__":
    main()
