"""bench_train.py -- the training legs of bench.py (never `value`): one optimisation step with its per-kernel roofline and CPU
baseline (`train`), the same step replayed from a HIP graph, the shipped N_train = 3072 loop with prefetched batch selection
(`train_loop`), and the step at the per-rank shard of an 8-way split (`train_shard_proxy`)."""
import time
import types

import numpy as np
import torch

import bench_common as C


def train_flop_per_ray(ins_num=None):
    m = C.mac_counts(C.INS_NUM if ins_num is None else ins_num)
    return 2.0 * (m["reference_fwd"] + m["reference_wgrad"] + m["reference_dgrad"]) * (2 * C.S_COARSE + C.N_IMP)


def train_leg(mc, mf, ro, rd, z, steps, dev, world=1, n=None, fuse_heads=False, mfma_split=False, ins_num=None, flat_adam=False):
    """Secondary measurement: rays/s of one full optimisation step on ONE batch of ``n`` rays (default 4096 per GPU;
    64+128 samples, perturb=1), the sequence of train_dmsr.py:32-64: dm_nerf forward with saved activations, img2mse on both
    levels, the emptiness penalizer on both levels (fused HIP kernels, tolerance / deta_w of
    configs/dmsr/train/study.txt), the Hungarian-matched object-code loss ins_criterion on both levels (device
    kernels: the reference solves the assignment with scipy on the host), backward (composite_bwd,
    dgrad, wgrad kernels), Adam(lr 5e-4).  Labels: a synthetic 9-object segmentation of the batch.
    world > 1: the batch is sharded over the ranks by dm_nerf_amd.distributed.sharded_train_step -- batch-global losses on
    all-gathered rgb / ins, penalizer sums and the 5.57 MB gradient arena all-reduced in place over RCCL; ``ro`` / ``rd``
    must then hold the same rays on every rank.  Also returns the per-kernel roofline of the three MFMA kernels of the
    fine-network pass, timed with HIP events on their stream (autograd.KERNEL_EVENTS): ``frac`` divides the MACs the kernel
    EXECUTES (C.mac_counts) by the peak of the MFMA type it runs on; ``algorithmic_tflops`` is the reference's FLOP count of
    the stage over the same time -- larger than ``achieved`` where the head re-association removed work."""
    C.quiesce()
    from dm_nerf_amd import autograd as G, distributed as D
    ins_num = C.INS_NUM if ins_num is None else ins_num
    mc.train(); mf.train()
    params = list(mc.parameters()) + list(mf.parameters())
    if flat_adam:                                       # extension (dm_nerf_amd.optim.FlatAdam): update + re-pack as two launches
        from dm_nerf_amd.optim import FlatAdam
        opt = FlatAdam((mc, mf), lr=5e-4, betas=(0.9, 0.999))
    else:                                               # the reference's optimizer (train_dmsr.py:124-125)
        opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.999))
    args = types.SimpleNamespace(perturb=1.0, N_importance=C.N_IMP, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05,
                                 fuse_heads=fuse_heads, mfma_split=mfma_split)
    n = C.N_RAYS * world if n is None else n
    g = torch.Generator(device=dev).manual_seed(0)
    target = torch.rand(n, 3, device=dev, generator=g)
    labels = torch.randint(0, 9, (n,), device=dev, generator=g)
    rays = torch.stack([ro[:n], rd[:n]])
    z = z[:n].contiguous()
    assert rays.shape[1] == n and z.shape[0] == n
    torch.manual_seed(0)                                # identical jitter streams on every rank
    torch.cuda.manual_seed(0)

    nbytes_seen = [0]

    def one():
        loss, nbytes_seen[0] = D.sharded_train_step(rays, z, target, labels, (mc, mf), args, opt, ins_num)
        return loss

    def fence():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
    C.warm_up(one, seconds=0.4 if world == 1 else 0.0)           # (N > 1: a fixed count -- every step contains collectives)
    fence()
    G.KERNEL_EVENTS = []
    D.collective_tally(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    tally = D.collective_tally()                        # (before the fence: its barrier is not part of the step)
    fence()
    dt = (time.perf_counter() - t0) / steps
    events, G.KERNEL_EVENTS = G.KERNEL_EVENTS, None
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    mc.eval(); mf.eval()
    mac = C.mac_counts(ins_num)
    flop_ref = train_flop_per_ray(ins_num) * n
    # per-kernel roofline of the fine-network launches (the dominant ones: 192 of the 256 samples per ray)
    n_local = D.ray_slice(n, D.world_info()[0], world)[1]
    m_fine = n_local * (C.S_COARSE + C.N_IMP)
    obi = (ins_num + 32) // 32
    fwd_exec = mac["fwd_fused"] if (mfma_split or fuse_heads) else mac["fwd"]
    products = C.split_products(mfma_split)               # 16-bit MFMA products per f32 product (1 on the f32 MFMA)
    peak = C.B16_MFMA_PEAK_TFLOPS if mfma_split else C.F32_MFMA_PEAK_TFLOPS
    exec_mac = {"mlp_fwd_train": fwd_exec, "mlp_bwd_data": mac["dgrad"], "mlp_bwd_weights": mac["wgrad"]}
    ref_mac = {"mlp_fwd_train": mac["reference_fwd"], "mlp_bwd_data": mac["reference_dgrad"], "mlp_bwd_weights": mac["reference_wgrad"]}
    names = {"mlp_fwd_train": f"mlp_fwd_kernel<{obi},false,true,false>", "mlp_bwd_data": f"mlp_bwd_kernel<{obi}>",
             "mlp_bwd_weights": "wgrad_kernel + wgrad_reduce_kernel + head_unfuse_kernel"}
    if mfma_split:
        names.update(C.split_kernel_names(mfma_split, obi))
    elif fuse_heads:
        names.update(mlp_fwd_train=f"mlp_fwd_kernel<{obi},true,true,false>")
    kernels = []
    for tag in ("mlp_fwd_train", "mlp_bwd_data", "mlp_bwd_weights"):
        ms = [b.elapsed_time(e) for t, M, b, e in events if t == tag and M == m_fine]
        if ms:
            k_ms = float(np.mean(ms))
            tf = 2.0 * exec_mac[tag] * products * m_fine / (k_ms * 1e-3) / 1e12
            tf_ref = 2.0 * ref_mac[tag] * m_fine / (k_ms * 1e-3) / 1e12
            entry = {"kernel": names[tag], "launches": len(ms), "kernel_ms": k_ms, "mac_per_sample_executed": exec_mac[tag],
                     "mfma_products_per_mac": products, "bound": "mfma", "unit": "TFLOP/s", "achieved": tf, "peak": peak, "frac": tf / peak,
                     "algorithmic_tflops": tf_ref}
            entry["frac_best_roof"], entry["bound_best_roof"] = entry["frac"], "mfma"
            hbm_bytes = C.train_hbm_bytes_per_sample() if ins_num == 13 else None
            if hbm_bytes is not None:
                # which roof is this kernel closer to?  (the opt-in f16x2 weight-gradient kernel reads the same f32 rows as the f32
                # one in less than half the time: it sits at 0.74 of the HBM spec and 0.38 of the 16-bit MFMA roof -- HBM-bound.)
                # `frac` stays the MFMA fraction (the definition of rounds 1-4, comparable across rounds); the HBM view rides beside it
                # and `frac_best_roof` = max of the two
                gbs = hbm_bytes[tag] * m_fine / (k_ms * 1e-3) / 1e9
                entry["hbm"] = {"achieved_GBs": gbs, "frac_of_spec_8TBs": gbs / C.HBM_PEAK_GBS, "frac_of_guide_measured_6.29TBs": gbs / C.HBM_ACHIEVABLE_GBS,
                                "bytes_per_sample": hbm_bytes[tag], "bytes_source": C.TRAIN_HBM_BYTES_SOURCE,
                                "traffic_measured_in_this_run": False}
                if gbs / C.HBM_PEAK_GBS > entry["frac"]:
                    entry.update(frac_best_roof=gbs / C.HBM_PEAK_GBS, bound_best_roof="hbm")
            kernels.append(entry)
    worst = min(kernels, key=lambda k: k["frac"]) if kernels else None
    flop_exec = 2.0 * (fwd_exec + mac["dgrad"] + mac["wgrad"]) * products * (2 * C.S_COARSE + C.N_IMP) * n
    return {"rays_per_s": n / dt, "ms_per_step": dt * 1e3, "tflops": flop_exec / dt / 1e12, "tflops_reference_flops": flop_ref / dt / 1e12,
            "frac_of_mfma_peak": {"executed": flop_exec / dt / 1e12 / (peak * world), "peak": peak,
                                  "note": "whole step incl. losses, compositing, Adam, on the MFMA work the three MLP kernels issue (C.mac_counts); "
                                          "`tflops_reference_flops` = the same time against SURVEY 8(d)'s 1013 MFLOP/ray "
                                          "(the head re-association removed work, "
                                          "so it is not a fraction of any roof)"},
            "final_loss": float(loss.detach()),
            "batch_rays": n, "ins_num": ins_num, "rays_this_rank": n_local,
            "allreduce_bytes_per_step": int(nbytes_seen[0]), "collectives_per_step": tally["count"] / max(steps, 1),
            "collective_kinds_per_step": {k: v / max(steps, 1) for k, v in tally["kinds"].items()},
            "collective_send_bytes_per_step": tally["bytes"] / max(steps, 1),
            "roofline": None if worst is None else {"bound": worst["bound"], "unit": worst["unit"], "peak": worst["peak"], "kernel": worst["kernel"],
                                                    "kernel_ms": worst["kernel_ms"], "achieved": worst["achieved"], "frac": worst["frac"],
                                                    "frac_best_roof_worst": min(k["frac_best_roof"] for k in kernels),
                                                    "samples_per_launch": m_fine, "all": kernels,
                                                    "note": "fine-network launches (192 samples/ray), HIP events on the launch stream; "
                                                            "`kernel` = the one furthest below the MFMA roof on EXECUTED MACs (`frac`, the "
                                                            "definition of every round); each kernel also carries `hbm` (its HBM bytes per sample "
                                                            "from committed PMC passes against 8 TB/s) and `frac_best_roof` = the larger of the two; "
                                                            "algorithmic_tflops = the reference's FLOP count of the stage over the same time"},
            "note": "fwd + img2mse + Hungarian-matched object-code loss (device) + fused emptiness penalizer + bwd + Adam, perturb=1"
                    + (f"; one batch sharded over {world} ranks (sharded_train_step)" if world > 1 else "")}


def graph_train_leg(mc, mf, ro, rd, z, steps, dev, n, mfma_split=False, ins_num=None, flat_adam=False):
    """The same optimisation step as `train_leg` replayed from ONE HIP graph (dm_nerf_amd.graphed.GraphedTrainStep: forward,
    losses, every backward kernel, Adam, weight re-packing in a single launch; bit-equal to the eager step,
    tests/test_gpu_driver.py) on a batch of ``n`` rays: ms per step and the eager figure next to it."""
    C.quiesce()
    from dm_nerf_amd.graphed import GraphedTrainStep
    ins_num = C.INS_NUM if ins_num is None else ins_num
    mc.train(); mf.train()
    if flat_adam:
        from dm_nerf_amd.optim import FlatAdam
        opt = FlatAdam((mc, mf), lr=5e-4, betas=(0.9, 0.999), capturable=True)
    else:
        opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=torch.tensor(5e-4, device=dev), betas=(0.9, 0.999), capturable=True)
    args = types.SimpleNamespace(perturb=1.0, N_importance=C.N_IMP, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05,
                                 mfma_split=mfma_split)
    g = torch.Generator(device=dev).manual_seed(0)
    target = torch.rand(n, 3, device=dev, generator=g)
    labels = torch.randint(0, 9, (n,), device=dev, generator=g)
    rays = torch.stack([ro[:n], rd[:n]])
    zz = z[:n].contiguous()
    torch.manual_seed(0); torch.cuda.manual_seed(0)
    gs = GraphedTrainStep((mc, mf), opt, args, ins_num, rays, zz, target, labels)
    C.warm_up(gs.step)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = gs.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mc.eval(); mf.eval()
    return {"ms_per_step": dt * 1e3, "rays_per_s": n / dt, "batch_rays": n, "final_loss": float(loss)}


def shard_proxy_leg(mc, mf, ro, rd, z, steps, dev, t_full_ms, n_full, t_3072_ms=None):
    """What ONE GPU can say about the 8-GPU strong-scaling run (SURVEY 8(e) caveat): the complete optimisation step at the
    per-rank shard of an 8-way split of the shipped batch sizes -- 384 rays (N_train 3072 / 8) and 512 rays (4096 / 8) -- eager
    and as one HIP graph (graph_train_leg).  predicted_strong_efficiency_8 = (t_full / 8) / t_shard: the fixed per-step cost
    (launch overheads, the loss / Adam kernels that do not shrink with the batch) is what keeps it below 1; the exchange itself
    (0.2 MB gather + one 5.57 MB all-reduce) is not in it.  The step is the product's default: fused loss tail, and at 384 rays the
    two levels' network backwards on two streams (distributed.overlap_enabled: it removes the partial round there)."""
    out = {}
    for n in (384, 512):
        r = train_leg(mc, mf, ro, rd, z, steps, dev, n=n)
        gr = graph_train_leg(mc, mf, ro, rd, z, steps, dev, n)
        rf = train_leg(mc, mf, ro, rd, z, steps, dev, n=n, flat_adam=True)
        gf = graph_train_leg(mc, mf, ro, rd, z, steps, dev, n, flat_adam=True)
        out[f"n{n}"] = {"ms_per_step": r["ms_per_step"], "rays_per_s": r["rays_per_s"], "graph_ms_per_step": gr["ms_per_step"],
                        "flat_adam_ms_per_step": rf["ms_per_step"], "flat_adam_graph_ms_per_step": gf["ms_per_step"],
                        "kernel_ms": {k["kernel"].split("<")[0].split(" ")[0]: k["kernel_ms"] for k in (r["roofline"] or {}).get("all", [])}}
    out["full_batch_rays"] = n_full
    out["full_batch_ms"] = t_full_ms
    full = t_full_ms * (4096.0 / n_full)
    out["predicted_strong_efficiency_8"] = {"eager_n512_of_4096": (full / 8.0) / out["n512"]["ms_per_step"],
                                            "graph_n512_of_4096": (full / 8.0) / out["n512"]["graph_ms_per_step"]}
    if t_3072_ms:                                            # the shipped N_train: 3072 rays over 8 ranks = 384 each
        out["full_3072_ms"] = t_3072_ms
        out["predicted_strong_efficiency_8"]["eager_n384_of_3072"] = (t_3072_ms / 8.0) / out["n384"]["ms_per_step"]
        best = min(out["n384"][k] for k in ("ms_per_step", "graph_ms_per_step", "flat_adam_ms_per_step", "flat_adam_graph_ms_per_step"))
        out["predicted_strong_efficiency_8"]["best_n384_of_3072"] = (t_3072_ms / 8.0) / best
    out["note"] = ("full optimisation step (same recipe as `train`) at the per-rank shard of an 8-way strong split, eager and as one HIP graph; "
                   "flat_adam_* = the same step with the extension optimizer dm_nerf_amd.optim.FlatAdam (torch.optim.Adam's update + the weight "
                   "re-packing as two launches) instead of the reference's torch.optim.Adam; "
                   "efficiency = (t_full / 8) / t_shard with t_full = the eager step with torch.optim.Adam")
    return out


def train_loop_leg(mc, mf, dev, steps, mfma_split=False):
    """The training LOOP as shipped (configs/dmsr/train/study.txt: N_train 3072; train_dmsr.py:24-64): per iteration the
    batch selection on the reference's numpy stream -- drawn ahead by dm_nerf_amd.prefetch.TrainBatchPrefetcher on a side
    thread, dataset resident in HBM, indices through pinned memory -- then the same optimisation step as `train`.
    Reports loop ms per iteration next to the step alone on a resident batch: the difference is the host-side overhead the
    prefetcher has to hide (< 3 % is the bar).  `inline_selection_ms` = the same loop with the drop-in
    get_select_full on the critical path (what the reference's loop structure costs here)."""
    C.quiesce()
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.networks import helpers as H
    from dm_nerf_amd.prefetch import TrainBatchPrefetcher
    from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical
    n_img, N = 4, C.N_TRAIN_SHIPPED
    g = torch.Generator().manual_seed(1)
    images = torch.rand(n_img, C.H_IMG, C.W_IMG, 3, generator=g)
    labels = torch.randint(0, 9, (n_img, C.H_IMG, C.W_IMG), generator=g).to(torch.int16)
    poses = torch.stack([pose_spherical(30.0 + 40.0 * k, -65.0, 7.0) for k in range(n_img)])
    K = dmsr_intrinsics(C.H_IMG, C.W_IMG)
    mc.train(); mf.train()
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    args = types.SimpleNamespace(perturb=1.0, N_importance=C.N_IMP, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05,
                                 mfma_split=mfma_split)
    z = H.z_val_sample(N, C.NEAR, C.FAR, C.S_COARSE, device=dev)
    torch.manual_seed(0); torch.cuda.manual_seed(0)

    def step(b):
        return D.sharded_train_step(b.rays, z, b.target_c, b.target_i, (mc, mf), args, opt, C.INS_NUM)[0]

    pf = TrainBatchPrefetcher(images, labels, poses, K, np.arange(n_img), N, dev, seed=0, depth=3, max_steps=steps + 3)
    it = iter(pf)
    first = next(it)
    step(first); step(next(it)); step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in it:
        loss = step(b)
    torch.cuda.synchronize()
    loop_ms = (time.perf_counter() - t0) / steps * 1e3
    pf.close()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(first)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / steps * 1e3
    # the reference's loop structure on the drop-in functions: selection + uploads on the critical path
    di, dl, dp = images.to(dev), labels.to(dev), poses.to(dev)
    np.random.seed(0)
    t0 = time.perf_counter()
    for _ in range(steps):
        img_i = np.random.choice(n_img)
        tc, ti, rays = H.get_select_full(di[img_i], dp[img_i, :3, :4], K, dl[img_i], N)
        D.sharded_train_step(rays, z, tc, ti, (mc, mf), args, opt, C.INS_NUM)
    torch.cuda.synchronize()
    inline_ms = (time.perf_counter() - t0) / steps * 1e3
    mc.eval(); mf.eval()
    return {"rays_per_s": N / (loop_ms * 1e-3), "batch_rays": N, "loop_ms": loop_ms, "step_ms_resident_batch": step_ms,
            "overhead_ms": loop_ms - step_ms, "overhead_frac": (loop_ms - step_ms) / step_ms, "inline_selection_ms": inline_ms,
            "final_loss": float(loss.detach()), "steps": steps,
            "note": "shipped N_train=3072: prefetched batch selection (reference numpy stream, side thread, pinned index upload, "
                    "resident dataset) + full optimisation step"}
