"""Helper process of tests/test_gpu_rccl.py (also run by __graft_entry__.smoke() with ``--quick``): backend "nccl" (= RCCL on
ROCm) in a world of ONE rank, initialised the way bench.py initialises it (``device_id=``), and the PRODUCT's collectives forced on
(dm_nerf_amd.distributed.force_collectives) so that they do not early-return at world 1:

  1. two ``sharded_train_step``s (packed all-gather of the per-ray outputs, 64-B all-reduce of the penalizer sums, in-place
     all-reduce of the 5.57 MB gradient arena) -> parameters ``torch.equal`` to the same two steps with no process group;
  2. a 38 400-ray band (60 rows x 640: what one of 8 ranks owns of a 640 x 480 frame; nine 4096-ray chunks + the ragged 1536-ray
     one) through ``FrameRenderer`` + its ONE all-gather, full and ``labels_only`` -> ``torch.equal`` to the un-gathered band;
  3. the arena all-reduce captured in a HIP graph once and replayed;
  4. (``--graph-step``) the WHOLE sharded step, collectives included, captured by ``GraphedTrainStep`` and replayed.

Prints one JSON line ending the run with "rccl ok".  A separate process: an RCCL failure must not take pytest down, and the
pytest process itself never owns a process group."""
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as O  # noqa: E402  (test infrastructure: seeded weights and the synthetic camera only)

INS = 13


def models_(seeds=(71, 72)):
    from dm_nerf_amd.networks import dm_nerf as M
    out = []
    for seed in seeds:
        m = M.DM_NeRF(8, 256, 63, 27, [4], INS)
        m.load_state_dict(O.make_weights(seed, INS, gain=1.7, sigma_bias=0.3))
        out.append(m.cuda().train())
    return out


def two_steps(n):
    from dm_nerf_amd import autograd as G, distributed as D
    models = models_()
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(35.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(9).choice(480 * 640, n, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]]).cuda()
    z = O.z_val_sample(n, 4.0, 15.0, 64).contiguous().cuda()
    g = torch.Generator().manual_seed(73)
    target = torch.rand(n, 3, generator=g).cuda()
    labels = torch.randint(0, 7, (n,), generator=g).cuda()
    opt = torch.optim.SGD([p for m in models for p in m.parameters()], lr=2e-2)
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05)
    torch.manual_seed(7)
    torch.cuda.manual_seed(7)
    losses, nbytes = [], 0
    for _ in range(2):
        loss, nbytes = D.sharded_train_step(rays, z, target, labels, models, args, opt, INS)
        losses.append(float(loss))
    arena = G.arena_slot(models[0])
    resident = None if arena is None else bool(arena[0].resident())
    flat = torch.cat([p.detach().reshape(-1) for m in models for p in m.parameters()])
    return losses, flat, nbytes, resident, (None if arena is None else arena[0]), models


def band(labels_only):
    from dm_nerf_amd import distributed as D
    mc, mf = [m.eval() for m in models_((1, 2))]
    K = O.dmsr_intrinsics(480, 640)
    c2w = O.pose_spherical(30.0, -65.0, 7.0).cuda()
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    # rows 0..59 of the 480-row frame = rank 0's band at world 8 (H = 60 here so that a world of one owns exactly that band)
    fr = D.FrameRenderer(60, 640, K, c2w, (mc, mf), 4.0, 15.0, args, chunk=4096, n_samples=64, labels_only=labels_only)
    assert fr.n_local == 38400 and fr.n_chunks == 10
    with torch.no_grad():
        for i in range(fr.n_chunks):
            fr.step(i)
        return fr, fr.gather()


def main():
    from dm_nerf_amd import distributed as D
    quick = "--quick" in sys.argv
    res = {}
    n = 96 if quick else 384
    # ---- the reference run: no process group, nothing forced (every collective is skipped)
    assert not dist.is_initialized() and not D.force_collectives()
    want_losses, want, nb0, res0, _, _ = two_steps(n)
    assert nb0 == 0 and res0 is None
    if not quick:
        frames0 = {lo: band(lo)[1] for lo in (False, True)}
    # ---- RCCL, world 1, exactly bench.py's initialisation
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist.all_reduce(torch.zeros(1, device=dev))
    torch.cuda.synchronize()
    res["init_s"] = time.perf_counter() - t0
    res["backend"] = dist.get_backend()
    D.force_collectives(True)
    got_losses, got, nbytes, resident, arena, models = two_steps(n)
    res["train_step"] = {"rays": n, "arena_bytes": nbytes, "arena_resident": resident, "losses": got_losses,
                         "params_equal": bool(torch.equal(got, want)), "max_abs_diff": float((got - want).abs().max())}
    assert resident is True and nbytes == 4 * want.numel(), (resident, nbytes)
    assert got_losses == want_losses and torch.equal(got, want), res["train_step"]
    # ---- the arena all-reduce, captured once and replayed (what GraphedTrainStep records at N > 1)
    before = arena.flat.clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        D.allreduce_grads(models, arena=arena)                          # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        nb = D.allreduce_grads(models, arena=arena)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    res["arena_allreduce_graph"] = {"bytes": nb, "replay_us": e0.elapsed_time(e1) / 20 * 1e3, "unchanged": bool(torch.equal(arena.flat, before))}
    assert nb == nbytes and torch.equal(arena.flat, before)
    # ---- one band of the frame + its gather
    if not quick:
        for lo in (False, True):
            fr, frame = band(lo)
            same = all(torch.equal(a, b) for a, b in zip(frame, frames0[lo]))
            res["band_labels_only" if lo else "band"] = {"rays": fr.n_local, "chunks": fr.n_chunks, "equal": bool(same),
                                                         "gathered_is_new_buffer": frame[0].data_ptr() != fr.band.data_ptr()}
            assert same and frame[0].data_ptr() != fr.band.data_ptr(), res
    # ---- the whole sharded step as one HIP graph, collectives inside
    if "--graph-step" in sys.argv:
        from dm_nerf_amd.graphed import GraphedTrainStep
        models = models_()
        K = O.dmsr_intrinsics(480, 640)
        ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(35.0, -65.0, 7.0))
        sel = torch.from_numpy(np.random.RandomState(9).choice(480 * 640, n, replace=False))
        rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]]).cuda()
        z = O.z_val_sample(n, 4.0, 15.0, 64).contiguous().cuda()
        g = torch.Generator().manual_seed(73)
        target, labels = torch.rand(n, 3, generator=g).cuda(), torch.randint(0, 7, (n,), generator=g).cuda()
        args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05)
        opt = torch.optim.Adam([p for m in models for p in m.parameters()], lr=torch.tensor(5e-4, device=dev), capturable=True)
        gs = GraphedTrainStep(models, opt, args, INS, rays, z, target, labels)
        ls = [float(gs.step(rays, z, target, labels)) for _ in range(3)]
        res["graph_step"] = {"losses": ls}
        assert all(np.isfinite(ls)) and ls[2] < ls[0], ls
    D.force_collectives(False)
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    res["rccl ok"] = True
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
