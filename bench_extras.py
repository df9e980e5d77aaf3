"""bench_extras.py -- the SECONDARY legs of bench.py (``python bench.py --extras``): never `value`, never on the driver's default run.

`frame` (one complete 640x480 pose through the frame driver), the opt-in inference modes (`render_fused_heads`, `render_split_*`,
`frame_split_*`), `render_ins59` (BASELINE config 3's object-head width), `manipulator` / `manipulator_frame` (BASELINE config 5's
manipulation render, per chunk and as a whole sharded frame).  They live here so that bench.py stays the contract: headline, CPU
baseline, the training step and its proxies."""
import time
import types

import numpy as np
import torch

import bench_common as C


def frame_leg(mc, mf, K, c2w, dev, mfma_split=False):
    """One complete 640x480 pose through the frame driver (distributed.render_path: raygen of the band, 75 chunks of
    N_test = 4096 rays, preallocated frame buffers, device-side label / confidence of ins_eval) -- what render_test does
    per pose (networks/tester.py:58-85) minus file output and CPU metrics."""
    C.quiesce()
    from dm_nerf_amd import distributed as D
    args = types.SimpleNamespace(perturb=False, N_importance=C.N_IMP, is_train=False, N_ins=None, N_test=C.N_RAYS, N_samples=C.S_COARSE, near=C.NEAR, far=C.FAR,
                                 mfma_split=mfma_split)
    with torch.no_grad():
        D.render_path(c2w[None].to(dev), (C.H_IMG, C.W_IMG, K), (mc, mf), args, labels_only=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            out = D.render_path(c2w[None].to(dev), (C.H_IMG, C.W_IMG, K), (mc, mf), args, labels_only=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    dt = min(ts)
    return {"frames_per_s": 1.0 / dt, "seconds_per_frame": dt, "rays_per_s": C.H_IMG * C.W_IMG / dt,
            "labels_in_frame": int(len(torch.unique(out["label"]))),
            "note": "render_path, one 640x480 pose: raygen + 75 x dm_nerf(4096 rays) + label/conf kernel, labels_only"
                    + ("; opt-in split-bf16 MFMA (args.mfma_split)" if mfma_split else "")}


def render_leg(pe, ve, mc, mf, ro, rd, z, steps, rgb_ref=None, fuse_heads=False, mfma_split=False, ins_num=None):
    """Not the headline: the same render step (dm_nerf on 4096-ray chunks of the band) in another configuration.
    fuse_heads: the activation-free rgb_feature_linear / ins_feature_linear folded into the hidden layers (SURVEY 8(f)-4;
    562 432 instead of 693 504 MAC per sample at ins_num 13, results equal up to f32 re-association).  mfma_split: additionally the
    GEMMs on the 16-bit MFMA with every f32 operand split into planes -- True / "bf16x3": three bf16 planes, six products;
    "f16x2": two f16 planes, three products (f32-class accuracy either way; csrc/mlp_split_impl.h, csrc/mlp_f16_impl.h).
    ins_num: the models' object-code width (BASELINE config 3: Replica office_0 = 59).  Roofline of the fine-network launch from
    HIP events around it: executed MACs x 16-bit products against the peak of the MFMA type used."""
    C.quiesce()
    from dm_nerf_amd.networks import render as R
    ins_num = C.INS_NUM if ins_num is None else ins_num
    fused = bool(fuse_heads or mfma_split)
    args = types.SimpleNamespace(perturb=False, N_importance=C.N_IMP, is_train=False, N_ins=None, fuse_heads=fused, mfma_split=mfma_split)
    n_chunks = ro.shape[0] // C.N_RAYS
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    with torch.no_grad():
        for i in range(2):
            out = R.dm_nerf(torch.stack([ro[:C.N_RAYS], rd[:C.N_RAYS]]), pe, ve, mc, mf, z, args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            c = i % n_chunks
            out = R.dm_nerf(torch.stack([ro[c * C.N_RAYS:(c + 1) * C.N_RAYS], rd[c * C.N_RAYS:(c + 1) * C.N_RAYS]]), pe, ve, mc, mf, z, args, _events=ev[i])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    mac = C.mac_counts(ins_num)
    exec_mac = mac["fwd_fused"] if fused else mac["fwd"]
    products = C.split_products(mfma_split)
    peak = C.B16_MFMA_PEAK_TFLOPS if mfma_split else C.F32_MFMA_PEAK_TFLOPS
    k_ms = float(np.mean([b.elapsed_time(e) for b, e in ev]))
    tf = 2.0 * exec_mac * products * C.N_RAYS * (C.S_COARSE + C.N_IMP) / (k_ms * 1e-3) / 1e12
    res = {"rays_per_s": C.N_RAYS / dt, "ms_per_step": dt * 1e3, "ins_num": ins_num, "mac_per_sample": exec_mac,
           "roofline": {"bound": "mfma", "unit": "TFLOP/s", "kernel_ms": k_ms, "achieved": tf, "peak": peak, "frac": tf / peak,
                        "mfma_products_per_mac": products, "note": "fine-network MLP launch, HIP events; executed MACs x 16-bit products per MAC"},
           "note": ("opt-in (args.mfma_split = %r): fused heads + split-operand 16-bit MFMA" % (mfma_split,) if mfma_split
                    else "opt-in (args.fuse_heads)" if fuse_heads else "default f32 path") + ", not the headline metric"}
    if rgb_ref is not None:                                      # same last chunk as the headline loop
        res["max_abs_rgb_diff_vs_layerwise"] = float((out['rgb_fine'] - rgb_ref).abs().max())
    return res


def manipulator_leg(mc, mf, K, dev, steps=3):
    """BASELINE config 5's render: ``manipulator`` (networks/manipulator.py:137-205) on one 4096-ray chunk with T = 1 and T = 2
    moved objects -- per call 1 + T coarse and 1 + T fine network passes of 64 / 192 samples per ray plus 2 T passes on the merged
    64 + 128 + 128 T depths (T = 1: 1152 network samples per ray = 4.5 x a dm_nerf render), three resamplings with random u
    (sample_pdf(det=False) even at evaluation), two exchanger
    rounds, the final composite.  Rays: the bench camera for the original view; each target view is the same camera
    moved by a rigid transform (what manipulator_demo does with the edited object's pose, :346-371)."""
    C.quiesce()
    from dm_nerf_amd.networks import helpers as H, manipulator as MA
    from dm_nerf_amd.synthetic import pose_spherical
    c2w = pose_spherical(30.0, -65.0, 7.0).to(dev)
    ro, rd = H.get_rays_k(C.H_IMG, C.W_IMG, K, c2w)
    ori = torch.stack([ro.reshape(-1, 3)[:C.N_RAYS], rd.reshape(-1, 3)[:C.N_RAYS]])
    tars = []
    for k in range(2):
        c2 = pose_spherical(30.0 + 4.0 * (k + 1), -65.0, 7.0 + 0.1 * (k + 1)).to(dev)
        to, td = H.get_rays_k(C.H_IMG, C.W_IMG, K, c2)
        tars.append(torch.stack([to.reshape(-1, 3)[:C.N_RAYS], td.reshape(-1, 3)[:C.N_RAYS]]))
    out = {}
    for T in (1, 2):
        args = types.SimpleNamespace(N_samples=C.S_COARSE, N_importance=C.N_IMP, near=C.NEAR, far=C.FAR, target_labels=list(range(1, T + 1)))
        torch.manual_seed(0); torch.cuda.manual_seed(0)
        with torch.no_grad():
            MA.manipulator(None, None, mc, mf, ori, tars[:T], args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                rgb, ins, _, _ = MA.manipulator(None, None, mc, mf, ori, tars[:T], args)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        samples = (1 + T) * (2 * C.S_COARSE + C.N_IMP) + 2 * T * (C.S_COARSE + C.N_IMP + C.N_IMP * T)   # network evaluations per ray
        mac = C.mac_counts(C.INS_NUM)["fwd"]
        out[f"T{T}"] = {"rays_per_s": C.N_RAYS / dt, "ms_per_call": dt * 1e3, "network_samples_per_ray": samples,
                        "tflops": 2.0 * mac * samples * C.N_RAYS / dt / 1e12,
                        "frac_of_f32_mfma_peak": 2.0 * mac * samples * C.N_RAYS / dt / 1e12 / C.F32_MFMA_PEAK_TFLOPS,
                        "finite": bool(torch.isfinite(rgb).all() and torch.isfinite(ins).all())}
    if C.HAVE_F16X2:                                       # opt-in (args.mfma_split = "f16x2"), T = 1; not an MFMA-roof fraction: three products per MAC
        args = types.SimpleNamespace(N_samples=C.S_COARSE, N_importance=C.N_IMP, near=C.NEAR, far=C.FAR, target_labels=[1], mfma_split="f16x2")
        torch.manual_seed(0); torch.cuda.manual_seed(0)
        with torch.no_grad():
            MA.manipulator(None, None, mc, mf, ori, tars[:1], args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                rgb, ins, _, _ = MA.manipulator(None, None, mc, mf, ori, tars[:1], args)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        out["T1_split_f16x2"] = {"rays_per_s": C.N_RAYS / dt, "ms_per_call": dt * 1e3, "finite": bool(torch.isfinite(rgb).all() and torch.isfinite(ins).all()),
                                 "note": "opt-in split-f16 network kernels (f32-class, not bitwise the default); not part of the T1 / T2 numbers"}
    out["note"] = ("manipulator() on one 4096-ray chunk, 64 + 128 samples, T moved objects (default f32 kernels); network_samples_per_ray = "
                   "(1+T)(64+192) + 2T(192+128T) -- the reference re-evaluates the original rays once per target (:190-193); "
                   "frac = whole call (incl. resampling, exchanger, composites) against the f32 MFMA roof")
    return out


def manipulator_frame_leg(mc, mf, K, dev, world=1):
    """BASELINE config 5's manipulation render as a FRAME: one whole 640 x 480 pose through the product's sharded frame driver
    (distributed.manipulate_frame = the per-pose chunk loop of manipulator_eval, networks/manipulator.py:232-270; T = 1 as the
    reference evaluates it), 75 chunks of N_test = 4096 rays, target view = ``trans @ pose``, the 2 + T draws per chunk from the
    device generator, ONE all-gather of the packed band per frame at N > 1.  At N = 1 also the time of ONE band of an 8-way split
    (``rank=0, world=8``: 38 400 rays) -- what one of 8 GPUs would take for its share of the same frame."""
    C.quiesce()
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.synthetic import pose_spherical
    pose = pose_spherical(30.0, -65.0, 7.0)
    ang = 0.15
    trans = torch.tensor([[np.cos(ang), -np.sin(ang), 0., 0.3], [np.sin(ang), np.cos(ang), 0., -0.2], [0., 0., 1., 0.1], [0., 0., 0., 1.]],
                         dtype=torch.float32)
    args = types.SimpleNamespace(N_samples=C.S_COARSE, N_importance=C.N_IMP, near=C.NEAR, far=C.FAR, N_test=C.N_RAYS, target_label=1)
    T = 1
    samples = (1 + T) * (2 * C.S_COARSE + C.N_IMP) + 2 * T * (C.S_COARSE + C.N_IMP + C.N_IMP * T)
    mac = C.mac_counts(C.INS_NUM)["fwd"]

    def timed(**kw):
        torch.manual_seed(0); torch.cuda.manual_seed(0)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            frame = D.manipulate_frame(C.H_IMG, C.W_IMG, K, pose.to(dev), [trans], (mc, mf), args, **kw)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, frame
    with torch.no_grad():                                      # warm-up: the first rows of the frame as one chunk
        D.manipulate_frame(8, C.W_IMG, K, pose.to(dev), [trans], (mc, mf), args, rank=0, world=1)
    dt, frame = timed()
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    n = C.H_IMG * C.W_IMG
    out = {"rays_per_s": n / dt, "s_per_frame": dt, "n_gpus": world, "T": T, "chunks": -(-n // C.N_RAYS), "network_samples_per_ray": samples,
           "tflops": 2.0 * mac * samples * n / dt / 1e12,
           "frac_of_f32_mfma_peak": 2.0 * mac * samples * n / dt / 1e12 / (C.F32_MFMA_PEAK_TFLOPS * world),
           "finite": bool(all(torch.isfinite(t).all() for t in frame)), "labels_in_frame": int(len(torch.unique(frame[1].argmax(-1)))),
           "note": "one whole 640x480 pose through distributed.manipulate_frame (manipulator_eval's chunk loop: 75 x 4096 rays, T = 1, "
                   "default f32 kernels); the whole frame incl. raygen of both views, resampling, exchanger, composites and the band "
                   "gather against the f32 MFMA roof of the GPUs used"}
    if world == 1:
        dt8, _ = timed(rank=0, world=8)
        out["band_of_8"] = {"rays": n // 8, "s": dt8, "predicted_8gpu_rays_per_s": n / dt8, "predicted_efficiency_before_gather": dt / 8 / dt8}
    return out


def generic_shapes_leg(ro, rd, z, dev, steps, shapes=((6, 128), (8, 192), (10, 320))):
    """Not the headline: the same 4096-ray chunk on network shapes other than the shipped one (config.py:31-41 netdepth / netwidth ->
    create_nerf :126-138) -- dm_nerf_amd/generic.py on csrc/gemm_nt.hip / gemm_tn.hip, the trunk of W <= 160 networks in inference
    on csrc/gemm_chain.hip.  Per shape: render time and the WHOLE render's MACs against the f32 MFMA peak (encoding, heads, compositing
    included in the time), and forward + backward of the same chunk (3 x the forward MACs)."""
    import warnings
    from dm_nerf_amd import config as Cfg
    from dm_nerf_amd.networks import render as R
    out = {"note": "opt-in shapes (no shipped config changes netdepth / netwidth); random-init weights; fractions are of the whole "
                   "dm_nerf call, not of one kernel", "shapes": {}}
    rays = torch.stack([ro[:C.N_RAYS], rd[:C.N_RAYS]])
    for D, W in shapes:
        C.quiesce()
        a = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=D, netwidth=W, ins_num=C.INS_NUM, device=dev)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pe, ve, mc, mf, _ = Cfg.create_nerf(a)
        mac = sum(p.numel() for n, p in mc.named_parameters() if n.endswith("weight"))
        ea = types.SimpleNamespace(perturb=False, N_importance=C.N_IMP, is_train=False, N_ins=None)
        with torch.no_grad():
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.2:            # (sustained rate: the clock settles over tens of ms of continuous work)
                R.dm_nerf(rays, pe, ve, mc, mf, z, ea)
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                R.dm_nerf(rays, pe, ve, mc, mf, z, ea)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        ta = types.SimpleNamespace(perturb=1.0, N_importance=C.N_IMP, is_train=True, N_ins=None)
        mc.train(), mf.train()

        def step():
            o = R.dm_nerf(rays, pe, ve, mc, mf, z, ta)
            (o['rgb_fine'].sum() + o['rgb_coarse'].sum() + o['ins_fine'].sum()).backward()
        n_t = max(4, steps // 2)
        for _ in range(4):                                  # (the first steps size the allocator's pools; the clock settles)
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_t):
            step()
        torch.cuda.synchronize()
        dtt = (time.perf_counter() - t0) / n_t
        tf = 2.0 * mac * (C.S_COARSE + C.S_COARSE + C.N_IMP) * C.N_RAYS / dt / 1e12
        out["shapes"][f"{D}x{W}"] = {"render_ms": dt * 1e3, "rays_per_s": C.N_RAYS / dt, "render_tflops": tf,
                                     "render_frac_of_mfma_peak": tf / C.F32_MFMA_PEAK_TFLOPS, "fwd_bwd_ms": dtt * 1e3,
                                     "fwd_bwd_frac_of_mfma_peak": 3.0 * tf * dt / dtt / C.F32_MFMA_PEAK_TFLOPS,
                                     "mac_per_sample": mac, "chained_trunk": bool(W <= 160)}
        del pe, ve, mc, mf
    return out
