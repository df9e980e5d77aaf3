"""bench.py -- rays/s of the DM-NeRF render hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the hot path (``dm_nerf``: coarse MLP -> composite -> resample -> fine MLP ->
composite) over one 4096-ray chunk of a synthetic 640x480 DM-SR 'study' frame, 64 + 128 samples,
deterministic sampling -- the chunk loop of the reference's render_test (networks/tester.py:63-72).
Inputs (rays, depth grid, packed weights) are resident in HBM before the timed region.  With N GPUs
every rank renders its own chunks (weak scaling) and the rendered tiles are all-gathered (RCCL).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INS_NUM = 13                 # DM-SR 'study' (data/color_dict.json: 13 labels)
N_RAYS = 4096                # N_test of every shipped config (configs/dmsr/train/study.txt)
S_COARSE, N_IMP = 64, 128
H_IMG, W_IMG = 480, 640
NEAR, FAR = 4.0, 15.0
MAC_PER_SAMPLE = 691712 + 128 * (INS_NUM + 1)          # 693 504
F32_MFMA_PEAK_TFLOPS = 157.3                           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    p.add_argument("--train-steps", type=int, default=5)
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the bounded baseline sample")
    p.add_argument("--ins-num", type=int, default=INS_NUM,
                   help="object-code width: 13 = DM-SR 'study' (the headline config); 59 / 93 = Replica office_0 / room_0 (BASELINE config 2)")
    return p.parse_args()


def flush_c_stdio():
    """RCCL prints its version banner with printf (NCCL_DEBUG=VERSION on this pool); on a pipe that sits in the C
    buffer until exit and would land BEHIND the JSON line.  Push it out early instead."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                           # noqa: BLE001
        pass


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; profiles/pmc_traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except Exception:
        return None


def build_models(device):
    from dm_nerf_amd import config as Cfg
    torch.manual_seed(0)
    args = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=8, netwidth=256,
                                 ins_num=INS_NUM, device=device)
    pe, ve, mc, mf, _ = Cfg.create_nerf(args)
    with torch.no_grad():                       # "trained-like": give the density head surfaces
        mc.density_linear.bias.add_(0.3)
        mf.density_linear.bias.add_(0.3)
    return pe, ve, mc.eval(), mf.eval()


def train_leg(mc, mf, ro, rd, z, steps, dev, world=1):
    """Secondary measurement: rays/s of one full optimisation step on a 4096-ray batch per GPU (64+128 samples,
    perturb=1), the sequence of train_dmsr.py:32-64: dm_nerf forward with saved activations, img2mse on both
    levels, the emptiness penalizer on both levels (fused HIP kernels, tolerance / deta_w of
    configs/dmsr/train/study.txt), the Hungarian-matched object-code loss ins_criterion on both levels (device
    kernels: the reference solves the assignment with scipy on the host), backward (composite_bwd,
    dgrad, wgrad kernels), Adam(lr 5e-4).  Labels: a synthetic 9-object segmentation of the batch.
    world > 1 (weak scaling): ONE batch of world x 4096 rays, sharded over the ranks by
    dm_nerf_amd.distributed.sharded_train_step -- batch-global losses on all-gathered rgb / ins, penalizer sums and
    the 5.57 MB gradient bucket all-reduced over RCCL; ``ro`` / ``rd`` must then hold the same rays on every rank."""
    from dm_nerf_amd import distributed as D
    mc.train(); mf.train()
    params = list(mc.parameters()) + list(mf.parameters())
    opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.999))
    args = types.SimpleNamespace(perturb=1.0, N_importance=N_IMP, is_train=True, N_ins=None, tolerance=0.05, deta_w=0.05)
    n = N_RAYS * world
    g = torch.Generator(device=dev).manual_seed(0)
    target = torch.rand(n, 3, device=dev, generator=g)
    labels = torch.randint(0, 9, (n,), device=dev, generator=g)
    rays = torch.stack([ro[:n], rd[:n]])
    assert rays.shape[1] == n and z.shape[0] == n
    torch.manual_seed(0)                                # identical jitter streams on every rank
    torch.cuda.manual_seed(0)

    def one():
        return D.sharded_train_step(rays, z, target, labels, (mc, mf), args, opt, INS_NUM)[0]

    def fence():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
    one(); one()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    fence()
    dt = (time.perf_counter() - t0) / steps
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    mc.eval(); mf.eval()
    flop = 2.0 * (2 * MAC_PER_SAMPLE + (MAC_PER_SAMPLE - 101248)) * (2 * S_COARSE + N_IMP) * n
    return {"rays_per_s": n / dt, "ms_per_step": dt * 1e3, "tflops": flop / dt / 1e12,
            "frac_of_f32_mfma_peak": flop / dt / 1e12 / (F32_MFMA_PEAK_TFLOPS * world), "final_loss": float(loss.detach()),
            "batch_rays": n, "note": "fwd + img2mse + Hungarian-matched object-code loss (device) + fused emptiness penalizer + bwd + Adam, perturb=1"
                                     + (f"; one batch sharded over {world} ranks (sharded_train_step), weak scaling" if world > 1 else "")}


def cpu_train_baseline(mc, mf, rays_cpu, z_cpu, seconds):
    """The same optimisation step on the oracle (CPU port: PyTorch autograd, scipy assignment, torch Adam) on a
    bounded number of rays of the same chunk; a warm-up step and a calibration step on 128 rays choose the sample size."""
    from oracle import ref_cpu as O
    sdc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mc.state_dict().items()}
    sdf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mf.state_dict().items()}
    opt = torch.optim.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(0)
    target = torch.rand(N_RAYS, 3, generator=g)
    labels = torch.randint(0, 9, (N_RAYS,), generator=g)

    def one(n):
        rays = rays_cpu[:, :n]
        o = O.dm_nerf(rays, sdc, sdf, z_cpu[:n], perturb=1.)
        loss = ((o['rgb_fine'] - target[:n]) ** 2).mean() + ((o['rgb_coarse'] - target[:n]) ** 2).mean() \
            + O.ins_criterion(o['ins_fine'], labels[:n], INS_NUM)[0].sum() + O.ins_criterion(o['ins_coarse'], labels[:n], INS_NUM)[0].sum() \
            + O.ins_penalizer(o['raw_fine'], o['z_vals_fine'], o['depth_fine'], rays[1], 0.05, 0.05).sum() \
            + O.ins_penalizer(o['raw_coarse'], o['z_vals_coarse'], o['depth_coarse'], rays[1], 0.05, 0.05).sum()
        opt.zero_grad(); loss.backward(); opt.step()

    n0 = 128
    one(n0)                                              # warm-up (thread pools, autograd graph caches)
    t0 = time.perf_counter(); one(n0); t1 = time.perf_counter() - t0
    # (the step's cost grows faster than linearly with the batch -- autograd's saved activations fall out of cache)
    n = int(min(512, max(128, 0.4 * (seconds / max(t1, 1e-3)) * n0 // 128 * 128)))
    t0 = time.perf_counter(); one(n); dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"one optimisation step on {n} rays of the same chunk (64+128 samples, perturb=1), oracle/ref_cpu + torch autograd + Adam, {dt:.1f} s"}


def fused_leg(pe, ve, mc, mf, ro, rd, z, steps, rgb_ref, split=False):
    """Not the headline: the same render step with an opt-in inference mode.  split=False: the activation-free
    rgb_feature_linear / ins_feature_linear folded into the hidden layers (562 432 instead of 693 504
    MAC per sample, results equal up to f32 re-association).  split=True: additionally the GEMMs on the bf16 MFMA with
    every f32 operand split into three bf16 planes and six products per term (f32-class accuracy, csrc/mlp_split.hip)."""
    from dm_nerf_amd.networks import render as R
    args = types.SimpleNamespace(perturb=False, N_importance=N_IMP, is_train=False, N_ins=None, fuse_heads=True, mfma_split=split)
    n_chunks = ro.shape[0] // N_RAYS
    with torch.no_grad():
        for i in range(2):
            out = R.dm_nerf(torch.stack([ro[:N_RAYS], rd[:N_RAYS]]), pe, ve, mc, mf, z, args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            c = i % n_chunks
            out = R.dm_nerf(torch.stack([ro[c * N_RAYS:(c + 1) * N_RAYS], rd[c * N_RAYS:(c + 1) * N_RAYS]]), pe, ve, mc, mf, z, args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    d = float((out['rgb_fine'] - rgb_ref).abs().max())          # same last chunk as the headline loop
    return {"rays_per_s": N_RAYS / dt, "ms_per_step": dt * 1e3, "mac_per_sample": 562432,
            "max_abs_rgb_diff_vs_layerwise": d,
            "note": ("opt-in (args.mfma_split): fused heads + split-bf16 MFMA, six bf16 products per f32 product" if split
                     else "opt-in (args.fuse_heads)") + ", not the headline metric"}


def cpu_baseline(mc, mf, rays_cpu, z_cpu, got_rgb, seconds):
    """The oracle (CPU port of the reference path) timed on this host's cores, bounded sample."""
    from oracle import ref_cpu as O
    sd_c = {k: v.detach().cpu() for k, v in mc.state_dict().items()}
    sd_f = {k: v.detach().cpu() for k, v in mf.state_dict().items()}
    cores = torch.get_num_threads()
    with torch.no_grad():
        n0 = 256
        t0 = time.perf_counter()
        O.dm_nerf(rays_cpu[:, :n0], sd_c, sd_f, z_cpu[:n0], perturb=0.)
        t1 = time.perf_counter() - t0                                   # calibration (also the warm-up)
        n = int(min(N_RAYS, max(256, (seconds / max(t1, 1e-3)) * n0 // 256 * 256)))
        t0 = time.perf_counter()
        want = O.dm_nerf(rays_cpu[:, :n], sd_c, sd_f, z_cpu[:n], perturb=0.)
        dt = time.perf_counter() - t0
    mse = float(((got_rgb[:n] - want['rgb_fine']) ** 2).mean())
    psnr = float(-10 * np.log10(max(mse, 1e-20)))
    return {"value": n / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{n} rays of the same 4096-ray chunk (64+128 samples, det), oracle/ref_cpu.dm_nerf, {dt:.1f} s"}, psnr, n


def main():
    global INS_NUM, MAC_PER_SAMPLE
    a = parse()
    INS_NUM = a.ins_num
    MAC_PER_SAMPLE = 691712 + 128 * (INS_NUM + 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # (DMNERF_BENCH_ONE_DEVICE=1 + DMNERF_BENCH_BACKEND=gloo: exercise the multi-rank code path on a 1-GPU box)
    if os.environ.get("DMNERF_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DMNERF_BENCH_BACKEND", "nccl")                      # "nccl" is RCCL on ROCm
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from dm_nerf_amd import _lib, distributed as D
    from dm_nerf_amd.networks import helpers as H, render as R
    from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical

    pe, ve, mc, mf = build_models(dev)
    K = dmsr_intrinsics(H_IMG, W_IMG)
    c2w = pose_spherical(30.0, -65.0, 7.0)
    # rank r owns a contiguous band of rows of the frame and generates its own rays (no scatter)
    rows = H_IMG // world
    ro, rd = H.get_rays_k(H_IMG, W_IMG, K, c2w.to(dev), row0=rank * rows, nrows=rows)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    n_chunks = ro.shape[0] // N_RAYS
    z = H.z_val_sample(N_RAYS, NEAR, FAR, S_COARSE, device=dev)
    args = types.SimpleNamespace(perturb=False, N_importance=N_IMP, is_train=False, N_ins=None)
    mc.blob(); mf.blob()                                        # packed weights resident
    # setup, not a step: load the code objects (a 32-ray render) and create the RCCL communicator (one scalar all-reduce),
    # so that --warmup 0 still times K steady-state steps
    with torch.no_grad():
        R.dm_nerf(torch.stack([ro[:32], rd[:32]]), pe, ve, mc, mf, z[:32].contiguous(), args)
    if world > 1:
        dist.all_reduce(torch.zeros(1, device=dev))
    torch.cuda.synchronize()
    flush_c_stdio()
    tile = torch.empty(N_RAYS, 3 + INS_NUM + 1, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def step(i, events=None):
        c = i % n_chunks
        rays = torch.stack([ro[c * N_RAYS:(c + 1) * N_RAYS], rd[c * N_RAYS:(c + 1) * N_RAYS]])
        out = R.dm_nerf(rays, pe, ve, mc, mf, z, args, _events=events)
        if world > 1:                                           # all-gather of the rendered tile (rgb | ins | depth)
            tile[:, :3] = out['rgb_fine']; tile[:, 3:3 + INS_NUM] = out['ins_fine']; tile[:, -1] = out['depth_fine']
            D.all_gather_cat(tile)
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for i in range(a.warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(a.steps):
            out = step(i, ev[i])
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    train_multi = None
    if world > 1 and not a.no_train:
        # every rank takes part; a failure here must not cost the headline line
        try:
            rows_t = -(-N_RAYS * world // W_IMG)
            tro, trd = H.get_rays_k(H_IMG, W_IMG, K, c2w.to(dev), row0=0, nrows=rows_t)
            zt = H.z_val_sample(N_RAYS * world, NEAR, FAR, S_COARSE, device=dev)
            train_multi = train_leg(mc, mf, tro.reshape(-1, 3), trd.reshape(-1, 3), zt, a.train_steps, dev, world)
        except Exception as e:                                  # noqa: BLE001
            train_multi = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        # dominant kernel = the fine-network fused PE+MLP launch (192 samples/ray): HIP events on its stream
        k_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
        flop_per_launch = 2.0 * MAC_PER_SAMPLE * (S_COARSE + N_IMP) * N_RAYS
        achieved = flop_per_launch / (k_ms * 1e-3) / 1e12
        rays_per_s = world * N_RAYS * a.steps / dt
        res = {
            "metric": "rays/sec (render) at 640x480, 64+128 samples", "value": rays_per_s, "unit": "rays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("DM-SR 'study'" if INS_NUM == 13 else "Replica-width object head,") + " 640x480 synthetic camera, dm_nerf render, 64 coarse + 128 fine samples, "
                                   f"4096-ray chunk per step per GPU, det sampling, ins_num={INS_NUM}, random-init weights",
                       "rays_per_step_per_gpu": N_RAYS, "parallelism": f"ray-sharded x{world}" + (" + RCCL all-gather of tiles" if world > 1 else "")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic() if INS_NUM == 13 else None,
                         "kernel": f"mlp_fwd_kernel<{(INS_NUM + 32) // 32},false,false,false> (fine network, 4096x192 samples)", "kernel_ms": k_ms,
                         "flop_per_launch": flop_per_launch},
            "path_tflops": rays_per_s * 2.0 * MAC_PER_SAMPLE * (2 * S_COARSE + N_IMP) / 1e12,
        }
        if world == 1 and not a.no_cpu_baseline:
            c = 0 if a.steps == 0 else (a.steps - 1) % n_chunks
            rays_cpu = torch.stack([ro[c * N_RAYS:(c + 1) * N_RAYS], rd[c * N_RAYS:(c + 1) * N_RAYS]]).cpu()
            base, psnr, n = cpu_baseline(mc, mf, rays_cpu, z.cpu(), out['rgb_fine'].cpu(), a.cpu_seconds)
            res["cpu_baseline"] = base
            res["psnr_vs_oracle_db"] = psnr
            res["speedup_vs_cpu"] = rays_per_s / base["value"]
        if world == 1:
            res["render_fused_heads"] = fused_leg(pe, ve, mc, mf, ro, rd, z, a.steps, out['rgb_fine'])
            res["render_split_bf16"] = fused_leg(pe, ve, mc, mf, ro, rd, z, a.steps, out['rgb_fine'], split=True)
        if world == 1 and not a.no_train:
            tb = None
            if not a.no_cpu_baseline:                   # (before the GPU leg: it updates the weights in place)
                tb = cpu_train_baseline(mc, mf, torch.stack([ro[:N_RAYS], rd[:N_RAYS]]).cpu(), z.cpu(), a.cpu_seconds / 2)
            res["train"] = train_leg(mc, mf, ro, rd, z, a.train_steps, dev)
            if tb is not None:
                res["train"]["cpu_baseline"] = tb
                res["train"]["speedup_vs_cpu"] = res["train"]["rays_per_s"] / tb["value"]
        if train_multi is not None:
            res["train"] = train_multi
    if world > 1:
        dist.barrier()                                          # nobody is still printing
    flush_c_stdio()
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
