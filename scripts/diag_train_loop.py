"""GPU box: where does the training LOOP (bench.py train_loop: prefetched batches + step) lose time against the step on a resident
batch?  Times, in one process: resident batch; prefetcher loop; prefetcher loop with the worker's numpy draw replaced by a cheap
one (GIL contention of the 307 200-pixel permutation); each with the penalizer's scalar tail as device kernels (shipped) or as the
chain of scalar torch operations it replaced (DIAG_OLD_TAIL=1)."""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                                   # noqa: E402
from dm_nerf_amd import _lib, distributed as D                      # noqa: E402
from dm_nerf_amd.networks import helpers as H, penalizer as P       # noqa: E402
from dm_nerf_amd.prefetch import TrainBatchPrefetcher               # noqa: E402


def old_tail():
    F = P._Penalizer

    def forward(ctx, raw, z, depth, rays_d, tolerance, deta_w, sharded=False):
        lib = _lib.load()
        N, S, ch = raw.shape
        C = ch - 4
        k2w, kh = P._consts(deta_w)
        part = torch.empty(N, 4, dtype=torch.float64, device=raw.device)
        _lib.check(lib.dmnerf_penalizer_fwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(depth), _lib.ptr(rays_d), N, S, C,
                                            float(tolerance), k2w, kh, _lib.ptr(part), _lib.stream()), "fwd")
        s = part.sum(0)
        nb = torch.clamp(s[1], min=1e-8)
        nm = torch.clamp(s[3], min=1e-8)
        loss = (s[0] / (C * nb) + s[2] / nm).to(torch.float32)
        inv = torch.stack([(1.0 / (C * nb)).to(torch.float32), (1.0 / nm).to(torch.float32)])
        ctx.save_for_backward(raw, z, depth, rays_d, inv)
        ctx.consts = (float(tolerance), k2w, kh, C)
        return loss.reshape(1)
    F.forward = staticmethod(forward)


def main():
    dev = torch.device("cuda:0")
    if os.environ.get("DIAG_OLD_TAIL") == "1":
        old_tail()
    mode = os.environ.get("DIAG_MODE") or False
    _, _, mc, mf = B.build_models(dev)
    n_img, N, steps = 4, B.N_TRAIN_SHIPPED, 40
    g = torch.Generator().manual_seed(1)
    images = torch.rand(n_img, B.H_IMG, B.W_IMG, 3, generator=g)
    labels = torch.randint(0, 9, (n_img, B.H_IMG, B.W_IMG), generator=g).to(torch.int16)
    from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical
    poses = torch.stack([pose_spherical(30.0 + 40.0 * k, -65.0, 7.0) for k in range(n_img)])
    K = dmsr_intrinsics(B.H_IMG, B.W_IMG)
    mc.train(); mf.train()
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
    args = types.SimpleNamespace(perturb=1.0, N_importance=B.N_IMP, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05, mfma_split=mode)
    z = H.z_val_sample(N, B.NEAR, B.FAR, B.S_COARSE, device=dev)
    step = lambda b: D.sharded_train_step(b.rays, z, b.target_c, b.target_i, (mc, mf), args, opt, B.INS_NUM)[0]

    def loop(cheap_draw):
        pf = TrainBatchPrefetcher(images, labels, poses, K, np.arange(n_img), N, dev, seed=0, depth=3, max_steps=steps + 3)
        if cheap_draw:
            rng = pf.stream.rng
            pf.stream.rng = types.SimpleNamespace(choice=lambda a, size=None, replace=True: (rng.randint(0, a, size=size) if size is not None else rng.choice(a)))
        it = iter(pf)
        first = next(it)
        step(first); step(next(it)); step(next(it))
        torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for b in it:
            h0 = time.perf_counter()
            step(b)
            host.append(time.perf_counter() - h0)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        pf.close()
        return dt, t_enq / steps * 1e3, float(np.median(host)) * 1e3, float(np.max(host)) * 1e3, first

    for name, cheap in (("prefetcher loop", False), ("prefetcher loop, cheap draw", True), ("prefetcher loop", False)):
        dt, enq, hmed, hmax, first = loop(cheap)
        print(f"{name:32s}: {dt:7.3f} ms/step   host: loop body incl. waiting for a batch {enq:6.3f} ms/step, step() call median {hmed:6.3f} max {hmax:6.3f} ms")
    host = []
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        step(first)
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    print(f"{'resident batch':32s}: {(time.perf_counter() - t0) / steps * 1e3:7.3f} ms/step   host: step() call median {np.median(host) * 1e3:6.3f} max {np.max(host) * 1e3:6.3f} ms")


if __name__ == "__main__":
    main()
