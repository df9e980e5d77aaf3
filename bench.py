"""bench.py -- rays/s of the DM-NeRF render hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
    python bench.py --gpus N ...          (no torchrun environment: launches exactly that command itself, see self_launch)

A "step" is one pass of the hot path (``dm_nerf``: coarse MLP -> composite -> resample -> fine MLP ->
composite) over one 4096-ray chunk of a synthetic 640x480 DM-SR 'study' frame, 64 + 128 samples,
deterministic sampling -- the chunk loop of the reference's render_test (networks/tester.py:63-72).
Inputs (rays, depth grid, packed weights) are resident in HBM before the timed region.  With N GPUs
every rank renders the chunks of its own band of image rows (``--scaling weak``, the default: 4096 rays per rank and
step; ``--scaling strong``: the 4096-ray step is split over the ranks) and the finished band is all-gathered ONCE PER
FRAME over RCCL (what distributed.render_frame does).

The LAST stdout line of rank 0 is the bounded contract line (<= 4096 bytes, strict JSON: the task statement's fields with `roofline`
and `cpu_baseline`, plus scalars of the training legs and, at N > 1, what the process group was -- `rccl`); the FULL record (every leg,
per-kernel tables, notes) goes to bench_full.json beside this file and, as one prefixed line, to stderr.  Modules: bench_common.py
(constants, MAC counts, peaks), bench_train.py (training legs), bench_extras.py (the `--extras` legs).  Secondary objects
(never part of `value`): `frame` (one complete 640x480 pose through the frame driver), `train` (one optimisation step,
its own per-kernel roofline and CPU baseline), `train_loop` (the shipped N_train = 3072 loop incl. batch selection),
`render_fused_heads`, `render_split_bf16`, `frame_split_bf16`, `train_fused_heads`, `train_split_bf16` (opt-in modes),
`render_ins59` / `train_ins59` (BASELINE config 3: Replica-width object head), `manipulator` (BASELINE config 5: the
manipulation render of networks/manipulator.py:137-205, T = 1 and 2 moved objects), `manipulator_frame` (the same as a whole
640x480 pose through the sharded frame driver, manipulator_eval's chunk loop :232-270; at N > 1 across the ranks), `train_shard_proxy` (the full step at the
384 / 512 rays one of 8 ranks sees under strong scaling: what a one-GPU box can say about the 8-GPU run).
"""
import argparse
import json
import os
import statistics
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bench_common as C        # noqa: E402  (constants, MAC counts, peaks)
import bench_extras as X        # noqa: E402  (the --extras legs)
import bench_train as T         # noqa: E402  (the training legs)
# (the experiment scripts under scripts/ read the workload and the legs through this module)
from bench_common import (FAR, H_IMG, INS_NUM, N_IMP, N_RAYS, N_TRAIN_SHIPPED, NEAR, S_COARSE, W_IMG, build_models, mac_counts,   # noqa: E402,F401
                          quiesce, warm_up)
from bench_train import graph_train_leg, train_leg, train_loop_leg         # noqa: E402,F401


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                   help="N > 1: weak = 4096 rays per rank and step; strong = one 4096-ray step (train: one 4096-ray batch, the "
                        "reference's batch semantics train_dmsr.py:24-31) split over the ranks")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-train", action="store_true", help="skip the training-step measurements")
    p.add_argument("--extras", action="store_true",
                   help="also run the secondary legs (frame driver, opt-in fused-heads / split-MFMA modes, ins-59, manipulator, "
                        "manipulation frame, graph replay): minutes of extra wall time, bench_full.json only -- never the contract line")
    p.add_argument("--no-extras", action="store_true", help="(accepted for older command lines: extras are off unless --extras)")
    p.add_argument("--train-steps", type=int, default=5)
    p.add_argument("--cpu-seconds", type=float, default=22.0,
                   help="budget of the CPU baseline per leg: BASELINE.md section 4 asks for N=4096 render / N=1024 train, median "
                        "of 3 after a warm-up (--cpu-seconds 80 does that); the default keeps the sample and runs as many repetitions "
                        "as fit -- one, on the GPU box's host -- so that the default bench stays near a minute; on a slow host the sample shrinks")
    p.add_argument("--full-record", default=os.path.join(ROOT, "bench_full.json"),
                   help="where rank 0 writes the FULL record (every leg, per-kernel tables, notes); the stdout line is the bounded contract line")
    p.add_argument("--ins-num", type=int, default=C.INS_NUM,
                   help="object-code width: 13 = DM-SR 'study' (the headline config); 59 / 93 = Replica office_0 / room_0 (BASELINE config 2)")
    return p.parse_args()


def cpu_train_baseline(mc, mf, rays_cpu, z_cpu, seconds):
    """the same optimisation step on the oracle (CPU port: PyTorch autograd, scipy assignment, torch
    Adam), N = 1024 rays of the same chunk, penalize on; median of up to 3 timed steps after a warm-up, anomaly detection
    OFF, plus one step with anomaly detection ON ("as shipped": the reference switches it on at import, dm_nerf.py:5).
    On a slow host the sample shrinks (stated) to keep the leg inside ``seconds``."""
    from oracle import ref_cpu as O
    sdc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mc.state_dict().items()}
    sdf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mf.state_dict().items()}
    opt = torch.optim.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(0)
    target = torch.rand(C.N_RAYS, 3, generator=g)
    labels = torch.randint(0, 9, (C.N_RAYS,), generator=g)

    def one(n):
        t0 = time.perf_counter()
        rays = rays_cpu[:, :n]
        o = O.dm_nerf(rays, sdc, sdf, z_cpu[:n], perturb=1.)
        loss = ((o['rgb_fine'] - target[:n]) ** 2).mean() + ((o['rgb_coarse'] - target[:n]) ** 2).mean() \
            + O.ins_criterion(o['ins_fine'], labels[:n], C.INS_NUM)[0].sum() + O.ins_criterion(o['ins_coarse'], labels[:n], C.INS_NUM)[0].sum() \
            + O.ins_penalizer(o['raw_fine'], o['z_vals_fine'], o['depth_fine'], rays[1], 0.05, 0.05).sum() \
            + O.ins_penalizer(o['raw_coarse'], o['z_vals_coarse'], o['depth_coarse'], rays[1], 0.05, 0.05).sum()
        opt.zero_grad(); loss.backward(); opt.step()
        return time.perf_counter() - t0

    torch.autograd.set_detect_anomaly(False)
    t_begin = time.perf_counter()
    one(128)                                              # warm-up (thread pools, allocator)
    t128, t256 = one(128), one(256)                       # calibration: step time ~ a + b n (the small step is mostly fixed cost)
    b = max((t256 - t128) / 128.0, 1e-4)
    a_ = max(t128 - 128.0 * b, 0.0)
    left = lambda: seconds - (time.perf_counter() - t_begin)
    n = 1024
    if (a_ + b * n) * 1.3 > left():                       # a host too slow for even one full-size step inside the budget
        n = int(max(128, min(1024, (left() / 1.3 - a_) / b // 128 * 128)))
    ts = [one(n)]
    while len(ts) < 3 and 1.05 * max(ts) <= left():       # up to three timed steps, as long as another one still fits (measured step times)
        ts.append(one(n))
    reps = len(ts)
    dt = statistics.median(ts)
    res = {"value": n / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"optimisation step on N={n} rays of the same chunk (64+128 samples, perturb=1, penalize on), oracle/ref_cpu + torch autograd + "
                     f"scipy assignment + Adam; median of {reps} after warm-up: {dt:.2f} s (all: {[round(t, 2) for t in ts]}), anomaly detection off",
           "host": C.host_info()}
    if 1.15 * max(ts) <= left():                          # budget left (--cpu-seconds 80): one step "as shipped"
        torch.autograd.set_detect_anomaly(True)
        try:
            dt_on = one(n)
        finally:
            torch.autograd.set_detect_anomaly(False)
        res["anomaly_on"] = {"value": n / dt_on, "seconds": dt_on,
                             "note": "one step with torch.autograd.set_detect_anomaly(True), as the reference ships (dm_nerf.py:5)"}
    return res


def cpu_baseline(mc, mf, rays_cpu, z_cpu, got_rgb, seconds):
    """the oracle (CPU port of the reference path) on this host's cores, the SAME 4096-ray chunk the GPU
    rendered last (64+128 samples, det): median of up to 3 timed renders after a warm-up.  On a slow host the repetitions,
    then the sample, shrink (stated) to keep the leg inside ``seconds``.  Also returns PSNR(HIP, oracle) on the sample."""
    from oracle import ref_cpu as O
    sd_c = {k: v.detach().cpu() for k, v in mc.state_dict().items()}
    sd_f = {k: v.detach().cpu() for k, v in mf.state_dict().items()}

    def one(n):
        t0 = time.perf_counter()
        with torch.no_grad():
            o = O.dm_nerf(rays_cpu[:, :n], sd_c, sd_f, z_cpu[:n], perturb=0.)
        return time.perf_counter() - t0, o

    one(512)                                              # warm-up (thread pools, page faults)
    t_small, _ = one(512)                                 # calibration
    n = C.N_RAYS
    est = t_small * n / 512 * 0.9                         # (a 512-ray render carries more fixed cost per ray than a 4096-ray one)
    if est > seconds:                                     # a host too slow for one full-size render inside the budget: a smaller sample
        n = int(max(512, min(C.N_RAYS, seconds / (t_small / 512) // 512 * 512)))
        est = t_small * n / 512
    reps = int(max(1, min(3, (seconds - 2.0 * t_small) // est)))     # as many renders of the sample (up to 3) as fit the budget
    runs = [one(n) for _ in range(reps)]
    ts = [t for t, _ in runs]
    dt = statistics.median(ts)
    want = runs[-1][1]
    mse = float(((got_rgb[:n] - want['rgb_fine']) ** 2).mean())
    psnr = float(-10 * np.log10(max(mse, 1e-20)))
    return {"value": n / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"N={n} rays = the same 4096-ray chunk (64+128 samples, det), oracle/ref_cpu.dm_nerf; median of {reps} after warm-up: "
                      f"{dt:.2f} s (all: {[round(t, 2) for t in ts]})",
            "host": C.host_info()}, psnr, n, want


def self_launch(a):
    """``python bench.py --gpus N`` with N > 1 and no torchrun environment: launch the N ranks ourselves -- the documented command
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same
    arguments>`` (one process per GPU over RCCL) -- relay rank 0's JSON line on stdout, everything else on stderr, and return the
    launcher's exit code.  The torchrun form (WORLD_SIZE set) never comes here."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or a.gpus) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    line = None
    for out in p.stdout:                                        # relay as it comes: a hung rank must not hide what was printed before
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            sys.stderr.write(out)
            sys.stderr.flush()
    rc = p.wait()
    if line is not None:
        sys.stdout.write(line if line.endswith("\n") else line + "\n")
        sys.stdout.flush()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the launched ranks exited 0 but printed no JSON line\n")
        rc = 1
    return rc


def init_world(a):
    """Rank / device / process group from torchrun's environment (one process per GPU; backend "nccl" = RCCL on ROCm).
    DMNERF_BENCH_ONE_DEVICE=1 + DMNERF_BENCH_BACKEND=gloo put every rank on the box's one GPU (the tests' dry runs)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    one_device = os.environ.get("DMNERF_BENCH_ONE_DEVICE") == "1"
    if one_device:
        # every rank on device 0: a DRY RUN of the multi-rank code path on a one-GPU box, never a scaling measurement -- only over
        # gloo (RCCL ranks sharing a device would look like an N-GPU run in the record), and the line is stamped one_device_dry_run
        if world > 1 and os.environ.get("DMNERF_BENCH_BACKEND", "nccl") != "gloo":
            raise SystemExit("bench.py: DMNERF_BENCH_ONE_DEVICE=1 is a dry run and needs DMNERF_BENCH_BACKEND=gloo "
                             "(N RCCL ranks on one GPU would be recorded as an N-GPU run)")
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DMNERF_BENCH_BACKEND", "nccl")
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    strong = a.scaling == "strong" and world > 1
    if strong and C.N_RAYS % world:
        raise SystemExit(f"--scaling strong needs a world size that divides {C.N_RAYS}")
    return types.SimpleNamespace(world=world, rank=rank, dev=dev, strong=strong, one_device=one_device)


def build_scene(w):
    """Models, camera and the product's frame driver for this rank's band (rank r owns a contiguous band of rows of the frame and
    generates its own rays -- no scatter; distributed.FrameRenderer = the body of render_frame renders it chunk by chunk into ONE
    packed band buffer and all-gathers that buffer once per frame; a "step" is one chunk of it)."""
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical
    pe, ve, mc, mf = C.build_models(w.dev)
    K = dmsr_intrinsics(C.H_IMG, C.W_IMG)
    c2w = pose_spherical(30.0, -65.0, 7.0)
    n_step = C.N_RAYS // w.world if w.strong else C.N_RAYS             # rays THIS rank renders per step
    args = types.SimpleNamespace(perturb=False, N_importance=C.N_IMP, is_train=False, N_ins=None)
    fr = D.FrameRenderer(C.H_IMG, C.W_IMG, K, c2w.to(w.dev), (mc, mf), C.NEAR, C.FAR, args, chunk=n_step, n_samples=C.S_COARSE)
    # the steps cycle through ALL chunks of the band, as render_frame does: at N > 1 the band (307 200 / N rays) is not a multiple
    # of 4096 and ends in a ragged chunk (tester.py:65-67) -- it is rendered like the others, so that every gathered frame is
    # complete, and `value` counts the rays each step really rendered (N = 1: 75 chunks of 4096, no ragged one)
    chunk_rays = [min(n_step, fr.n_local - c * n_step) for c in range(fr.n_chunks)]
    return types.SimpleNamespace(pe=pe, ve=ve, mc=mc, mf=mf, K=K, c2w=c2w, n_step=n_step, args=args, fr=fr, ro=fr.rays_o, rd=fr.rays_d,
                                 n_chunks=fr.n_chunks, chunk_rays=chunk_rays, z=fr.z_full)


def headline_leg(a, w, sc):
    """The timed region of the contract: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides;
    max over ranks; the dominant kernel's launches bracketed by HIP events the library records on its stream."""
    from dm_nerf_amd.networks import render as R
    if w.world > 1:
        import torch.distributed as dist
    fr, n_chunks = sc.fr, sc.n_chunks
    sc.mc.blob(); sc.mf.blob()                                  # packed weights resident
    # setup, not a step: load the code objects (a 32-ray render) and create the RCCL communicator (one scalar all-reduce),
    # so that --warmup 0 still times K steady-state steps
    # (with a scratch HIP-event pair: the runtime sets its timestamp machinery up on the first timed event record of the process
    # -- 35-55 ms on some boxes, measured -- and that is setup, not a step)
    ev_scratch = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    with torch.no_grad():
        R.dm_nerf(torch.stack([sc.ro[:32], sc.rd[:32]]), sc.pe, sc.ve, sc.mc, sc.mf, sc.z[:32].contiguous(), sc.args, _events=ev_scratch)
    if w.world > 1:
        dist.all_reduce(torch.zeros(1, device=w.dev))
    torch.cuda.synchronize()
    C.flush_c_stdio()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for b_, e_ in ev:                                           # create the hipEvent_t objects now (torch creates them at the first record)
        b_.record(); e_.record()
    torch.cuda.synchronize()
    gathers = [0]

    def gather_frame():
        gathers[0] += 1
        return fr.gather()

    def step(i, events=None):
        c = i % n_chunks
        rgb, ins, depth = fr.step(c, events=events)
        if w.world > 1 and c == n_chunks - 1:                   # the band is complete: one all-gather per frame
            gather_frame()
        return rgb, ins

    def barrier():
        if w.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    C.quiesce()
    out_rgb = None
    with torch.no_grad():
        for i in range(a.warmup):
            step(i, ev_scratch)
        barrier()
        gathers[0] = 0
        host_t = []
        t0 = time.perf_counter()
        for i in range(a.steps):
            out_rgb, _ = step(i, ev[i])
            host_t.append(time.perf_counter())
        if w.world > 1 and a.steps > 0 and gathers[0] == 0:
            gather_frame()                                      # fewer steps than a band has chunks: the frame's gather is still timed
        barrier()
        dt = time.perf_counter() - t0
    rays_rank = sum(sc.chunk_rays[i % n_chunks] for i in range(a.steps))      # rays THIS rank rendered in the timed region
    rays_total = rays_rank
    if w.world > 1:
        tmax = torch.tensor([dt], device=w.dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        cnt = torch.tensor([rays_rank], device=w.dev, dtype=torch.float64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        rays_total = int(cnt.item())
    return types.SimpleNamespace(dt=dt, rays_rank=rays_rank, rays_total=rays_total, ev=ev, gathers=gathers[0], host_t=host_t, t0=t0, out_rgb=out_rgb)


def device_identity(dev):
    """What tells two GPUs apart in a record: index, marketing name, PCI address (domain:bus:device) and UUID where this torch build
    exposes them, and the visibility masks the process was started with."""
    p = torch.cuda.get_device_properties(dev)
    pci = None
    if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pci = "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    uuid = getattr(p, "uuid", None)
    masks = {k: os.environ[k] for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if k in os.environ}
    return {"device_index": dev.index, "device_name": p.name, "pci_bus_id": pci, "uuid": None if uuid is None else str(uuid),
            "visible_devices": masks or None}


def rccl_evidence(w, sc, h):
    """N > 1, every rank, OUTSIDE the timed region: what the process group actually was -- backend, world size, and per rank the
    device it ran on and the rays it rendered in the timed region (one all_gather_object) -- so that a SCALE record alone answers
    "did RCCL see N ranks on N GPUs" (VERDICT r05 item 2).  ``distinct_devices`` counts distinct PCI addresses (UUIDs, else
    indices): N on an N-GPU node, 1 in the one-GPU dry runs of the tests."""
    import socket
    import torch.distributed as dist
    me = dict(device_identity(w.dev), rank=w.rank, host=socket.gethostname(), rays_rendered=int(h.rays_rank))
    ranks = [None] * w.world
    dist.all_gather_object(ranks, me)
    key = lambda r: (r["host"], r["pci_bus_id"] or r["uuid"] or r["device_index"])
    band = sc.fr.band
    ver = None
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                           # noqa: BLE001
        pass
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver if dist.get_backend() == "nccl" else None,
            "ranks": sorted(ranks, key=lambda r: r["rank"]), "distinct_devices": len({key(r) for r in ranks}),
            "one_device_dry_run": bool(w.one_device), "frame_gathers_timed": h.gathers,
            "gather_send_bytes_per_rank": None if band is None else band.numel() * band.element_size(),
            "gather_bytes_per_frame": None if band is None else band.numel() * band.element_size() * w.world}


def multi_rank_legs(a, w, sc):
    """N > 1 only, every rank takes part: the sharded training step and BASELINE config 5's manipulation frame across the ranks.
    A failure here must not cost the headline line."""
    from dm_nerf_amd.networks import helpers as H
    train_multi = mani_multi = None
    if w.world > 1 and not a.no_train:
        try:
            n_train = C.N_RAYS if w.strong else C.N_RAYS * w.world
            rows_t = -(-n_train // C.W_IMG)
            tro, trd = H.get_rays_k(C.H_IMG, C.W_IMG, sc.K, sc.c2w.to(w.dev), row0=0, nrows=rows_t)
            zt = H.z_val_sample(n_train, C.NEAR, C.FAR, C.S_COARSE, device=w.dev)
            train_multi = T.train_leg(sc.mc, sc.mf, tro.reshape(-1, 3), trd.reshape(-1, 3), zt, a.train_steps, w.dev, w.world, n=n_train)
            train_multi["scaling"] = a.scaling
        except Exception as e:                                  # noqa: BLE001
            train_multi = {"error": f"{type(e).__name__}: {e}"}
    if w.world > 1 and a.extras:
        try:                                                    # (never `value`)
            mani_multi = X.manipulator_frame_leg(sc.mc, sc.mf, sc.K, w.dev, world=w.world)
        except Exception as e:                                  # noqa: BLE001
            mani_multi = {"error": f"{type(e).__name__}: {e}"}
    return train_multi, mani_multi


def headline_record(a, w, sc, h, rccl=None):
    """The contract's JSON object from the timed region: value, roofline of the dominant kernel, config (+ ``rccl`` at N > 1)."""
    # dominant kernel = the fine-network fused PE+MLP launch (192 samples/ray): HIP events on its stream
    k_ms = float(np.mean([s.elapsed_time(e) for s, e in h.ev])) if a.steps else None       # (--steps 0: null, never NaN)
    # (average over the timed launches of rank 0; with a ragged chunk among them, the average launch is that much smaller)
    flop_per_launch = 2.0 * C.MAC_PER_SAMPLE * (C.S_COARSE + C.N_IMP) * (h.rays_rank / max(a.steps, 1))
    achieved = None if not k_ms else flop_per_launch / (k_ms * 1e-3) / 1e12
    rays_per_s = h.rays_total / h.dt if a.steps else None
    traffic, traffic_src = C.pmc_traffic() if (C.INS_NUM == 13 and sc.n_step == C.N_RAYS) else (None, None)
    res = {
        "metric": "rays/sec (render) at 640x480, 64+128 samples", "value": rays_per_s, "unit": "rays/s",
        "n_gpus": w.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": (h.dt / a.steps * 1e3) if a.steps else None,
        "higher_is_better": True, "scaling": "strong" if w.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("DM-SR 'study'" if C.INS_NUM == 13 else "Replica-width object head,")
                               + " 640x480 synthetic camera, dm_nerf render, 64 coarse + 128 fine samples, "
                               f"{sc.n_step}-ray chunk per step per GPU, det sampling, ins_num={C.INS_NUM}, random-init weights",
                   "rays_per_step_per_gpu": sc.n_step, "rays_in_timed_region": h.rays_total,
                   "chunks_per_band": sc.n_chunks,
                   "ragged_chunk_rays": (sc.chunk_rays[-1] if sc.chunk_rays and sc.chunk_rays[-1] != sc.n_step else 0),
                   "parallelism": f"ray-sharded x{w.world}" + (
                       f" + one RCCL all-gather of the rank's band per frame ({h.gathers} in the timed region)" if w.world > 1 else "")},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": C.F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": None if achieved is None else achieved / C.F32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_measured_in_this_run": False if traffic is not None else None,
                     "kernel": f"mlp_fwd_kernel<{(C.INS_NUM + 32) // 32},false,false,false> (fine network, {sc.n_step}x192 samples)", "kernel_ms": k_ms,
                     "flop_per_launch": flop_per_launch},
        "path_tflops": None if rays_per_s is None else rays_per_s * 2.0 * C.MAC_PER_SAMPLE * (2 * C.S_COARSE + C.N_IMP) / 1e12,
    }
    if rccl is not None:
        res["rccl"] = rccl
    if w.one_device:
        res["one_device_dry_run"] = True                        # every rank on device 0 (gloo): a code-path dry run, not a scaling point
    if a.steps > 1:                                             # diagnostic: how long the HOST took to enqueue each step (no sync inside the loop)
        hd = np.diff(np.array([h.t0] + h.host_t)) * 1e3
        res["host_enqueue_ms_per_step"] = {"median": float(np.median(hd)), "max": float(hd.max()), "first": float(hd[0])}
    return res


def single_gpu_render_legs(a, w, sc, h, res):
    """N = 1, rank 0: the full dict of the last timed chunk (checked against the frame driver), the CPU baseline on the same
    chunk, and the secondary render legs (frame, opt-in modes, manipulator, ins-59).  Returns what the training legs reuse."""
    from dm_nerf_amd.networks import render as R
    pe, ve, mc, mf, ro, rd, z = sc.pe, sc.ve, sc.mc, sc.mf, sc.ro, sc.rd, sc.z
    # the full 10-key dict of the chunk the timed loop rendered last (the frame driver keeps rgb / ins / depth only):
    # the same call, untimed -- what the CPU comparison and the opt-in legs are checked against
    c_last = 0 if a.steps == 0 else (a.steps - 1) % sc.n_chunks
    with torch.no_grad():
        out = R.dm_nerf(torch.stack([ro[c_last * C.N_RAYS:(c_last + 1) * C.N_RAYS], rd[c_last * C.N_RAYS:(c_last + 1) * C.N_RAYS]]), pe, ve, mc, mf, z, sc.args)
    torch.cuda.synchronize()
    if a.steps:
        assert torch.equal(out['rgb_fine'], h.out_rgb), "frame driver and dm_nerf disagree on the same chunk"
    if not a.no_cpu_baseline:
        rays_cpu = torch.stack([ro[c_last * C.N_RAYS:(c_last + 1) * C.N_RAYS], rd[c_last * C.N_RAYS:(c_last + 1) * C.N_RAYS]]).cpu()
        base, psnr, n, want = cpu_baseline(mc, mf, rays_cpu, z.cpu(), out['rgb_fine'].cpu(), a.cpu_seconds)
        res["cpu_baseline"] = base
        res["psnr_vs_oracle_db"] = psnr
        res["label_flips_vs_oracle"] = {"rays": n, "ins_fine": int((out['ins_fine'].cpu()[:n].argmax(-1) != want['ins_fine'].argmax(-1)).sum()),
                                        "ins_coarse": int((out['ins_coarse'].cpu()[:n].argmax(-1) != want['ins_coarse'].argmax(-1)).sum())}
        res["speedup_vs_cpu"] = None if res["value"] is None else res["value"] / base["value"]
    wide = None
    if a.extras:
        res["frame"] = X.frame_leg(mc, mf, sc.K, sc.c2w, w.dev)
        res["render_fused_heads"] = X.render_leg(pe, ve, mc, mf, ro, rd, z, a.steps, out['rgb_fine'], fuse_heads=True)
        res["render_split_bf16"] = X.render_leg(pe, ve, mc, mf, ro, rd, z, a.steps, out['rgb_fine'], mfma_split=True)
        res["frame_split_bf16"] = X.frame_leg(mc, mf, sc.K, sc.c2w, w.dev, mfma_split=True)
        if C.HAVE_F16X2:
            res["render_split_f16x2"] = X.render_leg(pe, ve, mc, mf, ro, rd, z, a.steps, out['rgb_fine'], mfma_split="f16x2")
            res["frame_split_f16x2"] = X.frame_leg(mc, mf, sc.K, sc.c2w, w.dev, mfma_split="f16x2")
        res["manipulator"] = X.manipulator_leg(mc, mf, sc.K, w.dev)
        res["manipulator_frame"] = X.manipulator_frame_leg(mc, mf, sc.K, w.dev)
        if C.INS_NUM != 59:                               # BASELINE config 3: Replica office_0 width (59 objects), near / far of its config
            wide = C.build_models(w.dev, 59)
            res["render_ins59"] = X.render_leg(*wide, ro, rd, z, a.steps, ins_num=59)
        res["generic_shapes"] = X.generic_shapes_leg(ro, rd, z, w.dev, max(a.steps, 8))
    return wide


def single_gpu_train_legs(a, w, sc, res, wide):
    """N = 1, rank 0: the optimisation step (`train`, with its CPU baseline), and the secondary training legs."""
    mc, mf, ro, rd, z, dev = sc.mc, sc.mf, sc.ro, sc.rd, sc.z, w.dev
    tb = None
    if not a.no_cpu_baseline:                           # (before the GPU leg: it updates the weights in place)
        tb = cpu_train_baseline(mc, mf, torch.stack([ro[:C.N_RAYS], rd[:C.N_RAYS]]).cpu(), z.cpu(), a.cpu_seconds)
    res["train"] = T.train_leg(mc, mf, ro, rd, z, a.train_steps, dev)
    if tb is not None:
        res["train"]["cpu_baseline"] = tb
        res["train"]["speedup_vs_cpu"] = res["train"]["rays_per_s"] / tb["value"]
    res["train_loop"] = T.train_loop_leg(mc, mf, dev, max(a.train_steps * 4, 20))
    if a.extras:
        res["train_graph"] = T.graph_train_leg(mc, mf, ro, rd, z, max(a.train_steps, 10), dev, C.N_RAYS)
        res["train_graph"]["note"] = "the `train` step (4096 rays) replayed from one HIP graph (GraphedTrainStep)"
        if wide is not None:
            t9 = T.train_leg(wide[2], wide[3], ro, rd, z, a.train_steps, dev, ins_num=59)
            res["train_ins59"] = {k: t9[k] for k in ("rays_per_s", "ms_per_step", "tflops", "frac_of_mfma_peak", "roofline", "ins_num")}
        tf = T.train_leg(mc, mf, ro, rd, z, a.train_steps, dev, fuse_heads=True)
        res["train_fused_heads"] = {"rays_per_s": tf["rays_per_s"], "ms_per_step": tf["ms_per_step"], "roofline": tf["roofline"],
                                    "note": "opt-in (args.fuse_heads in training): forward on the fused-heads blob, same backward; not part of `train`"}
        for key, mode in (("train_split_bf16", True),) + ((("train_split_f16x2", "f16x2"),) if C.HAVE_F16X2 else ()):
            ts = T.train_leg(mc, mf, ro, rd, z, a.train_steps, dev, mfma_split=mode)
            res[key] = {"rays_per_s": ts["rays_per_s"], "ms_per_step": ts["ms_per_step"],
                        "frac_of_mfma_peak": ts["frac_of_mfma_peak"], "roofline": ts["roofline"],
                        "note": "opt-in (args.mfma_split in training): forward, data gradients and weight gradients on the "
                                "split-operand 16-bit MFMA kernels "
                                f"(f32-class values: {C.split_products(mode)} products per f32 product, f32 accumulation); not part of `train`"}
            tl = T.train_loop_leg(mc, mf, dev, max(a.train_steps * 4, 20), mfma_split=mode)
            res[key]["train_loop"] = {k: tl[k] for k in ("rays_per_s", "batch_rays", "loop_ms", "step_ms_resident_batch", "overhead_frac")}
            res[key]["graph_ms_per_step"] = T.graph_train_leg(mc, mf, ro, rd, z, max(a.train_steps, 10), dev, C.N_RAYS, mfma_split=mode)["ms_per_step"]
    # (last: the extension optimizer re-points the models' parameters at its flat vector)
    res["train_shard_proxy"] = T.shard_proxy_leg(mc, mf, ro, rd, z, max(a.train_steps, 10), dev, res["train"]["ms_per_step"], C.N_RAYS,
                                               t_3072_ms=res["train_loop"]["step_ms_resident_batch"])


LINE_LIMIT = 4096               # bytes: the contract line the driver parses (r05's 23.7 KB line came back `parsed: null`)
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "roofline", "cpu_baseline")


def _clean(x, sig=7, text=200):
    """JSON-strict, bounded copy: NaN / inf -> None, numpy / torch scalars -> Python numbers, floats to ``sig`` significant digits,
    strings cut at ``text`` characters."""
    if isinstance(x, dict):
        return {str(k): _clean(v, sig, text) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v, sig, text) for v in x]
    if isinstance(x, (bool, type(None))):
        return x
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        return float(f"{x:.{sig}g}") if np.isfinite(x) else None
    if isinstance(x, str):
        return x if len(x) <= text else x[:text - 3] + "..."
    return _clean(str(x), sig, text)


def _pick(d, keys):
    return None if not isinstance(d, dict) else {k: d[k] for k in keys if k in d}


def contract_record(res, full_path=None):
    """The bounded object of the LAST stdout line: the task's contract fields and nothing else -- scalars, one-line strings, and at
    N > 1 the per-rank device table.  Everything else of ``res`` (per-kernel tables, opt-in legs, proxies, notes) lives in the full
    record (``full_record``).  Pure function of ``res``: tests/test_bench_launch.py runs it on canned records."""
    out = {k: res.get(k) for k in CONTRACT_KEYS if k != "cpu_baseline"}
    out["config"] = _pick(res.get("config"), ("workload", "rays_per_step_per_gpu", "rays_in_timed_region", "chunks_per_band", "ragged_chunk_rays",
                                              "parallelism"))
    out["roofline"] = _pick(res.get("roofline"), ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                  "traffic_measured_in_this_run", "kernel", "kernel_ms", "flop_per_launch"))
    if "cpu_baseline" in res:                                    # rank 0 at N = 1 only
        out["cpu_baseline"] = _pick(res["cpu_baseline"], ("value", "unit", "cores", "kind", "sample"))
        out["cpu_baseline"]["cpu_model"] = (res["cpu_baseline"].get("host") or {}).get("cpu_model")
    for k in ("psnr_vs_oracle_db", "label_flips_vs_oracle", "speedup_vs_cpu", "path_tflops", "train_ms_per_step", "train_rays_per_s",
              "train_batch_rays", "train_roofline_frac_worst", "train_step_frac_of_mfma_peak", "train_cpu_rays_per_s", "train_speedup_vs_cpu",
              "train_loop_ms", "train_loop_overhead_frac", "train_shard_n384_best_ms", "train_shard_predicted_efficiency_8",
              "one_device_dry_run", "wall_s"):
        if k in res:
            out[k] = res[k]
    t = res.get("train")
    if isinstance(t, dict) and "error" in t:
        out["train_error"] = t["error"]
    elif isinstance(t, dict) and res.get("n_gpus", 1) > 1:       # what crossed the links per optimisation step
        out["train"] = _pick(t, ("batch_rays", "rays_this_rank", "ms_per_step", "rays_per_s", "scaling", "allreduce_bytes_per_step",
                                 "collectives_per_step", "collective_kinds_per_step", "collective_send_bytes_per_step"))
    if "rccl" in res:
        r = dict(res["rccl"])
        r["ranks"] = [_pick(x, ("rank", "device_index", "device_name", "pci_bus_id", "uuid", "rays_rendered")) for x in r.get("ranks", [])]
        out["rccl"] = r
    out["full_record"] = full_path
    out = _clean(out)
    # a table of ranks is the only part that grows with N: shrink it before ever exceeding the limit (names once, then uuids away)
    if len(json.dumps(out, allow_nan=False)) > LINE_LIMIT and "rccl" in out:
        names = sorted({x.get("device_name") for x in out["rccl"]["ranks"]}, key=str)
        out["rccl"]["device_names"] = names
        for x in out["rccl"]["ranks"]:
            x.pop("device_name", None)
            if x.get("pci_bus_id"):
                x.pop("uuid", None)
    if len(json.dumps(out, allow_nan=False)) > LINE_LIMIT and "rccl" in out and len(out["rccl"]["ranks"]) > 8:
        out["rccl"]["ranks_total"] = len(out["rccl"]["ranks"])  # (the full table stays in the full record)
        out["rccl"]["ranks"] = out["rccl"]["ranks"][:8]
    if len(json.dumps(out, allow_nan=False)) > LINE_LIMIT:
        out["config"]["workload"] = out["config"]["workload"][:80]
        for k in ("traffic_source",):
            out["roofline"].pop(k, None)
        if "cpu_baseline" in out:
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:80]
    return out


def contract_line(res, full_path=None):
    """``contract_record`` as ONE strict-JSON line of at most LINE_LIMIT bytes (asserted: a longer line is a bug here, not the
    driver's problem)."""
    line = json.dumps(contract_record(res, full_path), allow_nan=False, separators=(", ", ": "))
    assert len(line.encode()) <= LINE_LIMIT and "\n" not in line, f"contract line is {len(line.encode())} bytes (limit {LINE_LIMIT})"
    return line


def lift_scalars(res):
    """Top-level scalars of the secondary legs: the training claim in the driver's parsed record."""
    t = res.get("train")
    if isinstance(t, dict) and "error" not in t:
        res["train_ms_per_step"] = t["ms_per_step"]
        res["train_rays_per_s"] = t["rays_per_s"]
        res["train_batch_rays"] = t["batch_rays"]
        res["train_roofline_frac_worst"] = None if not t.get("roofline") else t["roofline"]["frac"]
        res["train_step_frac_of_mfma_peak"] = t["frac_of_mfma_peak"]["executed"]
        if isinstance(t.get("cpu_baseline"), dict):
            res["train_cpu_rays_per_s"] = t["cpu_baseline"]["value"]
            res["train_speedup_vs_cpu"] = t.get("speedup_vs_cpu")
    tl = res.get("train_loop")
    if isinstance(tl, dict):
        res["train_loop_ms"] = tl["loop_ms"]
        res["train_loop_overhead_frac"] = tl["overhead_frac"]
    sp = res.get("train_shard_proxy")
    if isinstance(sp, dict) and "n384" in sp:
        res["train_shard_n384_best_ms"] = min(v for k, v in sp["n384"].items() if k.endswith("ms_per_step"))
        res["train_shard_predicted_efficiency_8"] = sp["predicted_strong_efficiency_8"].get("best_n384_of_3072")


def write_full_record(res, path):
    """The FULL record: to ``path`` (default bench_full.json beside bench.py) and, as one line, to stderr -- BEFORE the contract line
    goes to stdout, so that the last line of any capture is the contract line.  Returns the path written (or None)."""
    full = _clean(res, sig=9, text=2000)
    txt = json.dumps(full, allow_nan=False)
    sys.stderr.write("bench.py full record: " + txt + "\n")
    sys.stderr.flush()
    for cand in (path, os.path.join("/tmp", "bench_full.json")):
        try:
            with open(cand, "w") as f:
                json.dump(full, f, allow_nan=False, indent=1)
                f.write("\n")
            return cand
        except OSError:
            continue
    return None


def main():
    a = parse()
    t_wall = time.perf_counter()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    C.INS_NUM = a.ins_num
    C.MAC_PER_SAMPLE = 691712 + 128 * (C.INS_NUM + 1)
    w = init_world(a)
    from dm_nerf_amd import _lib
    C.HAVE_F16X2 = "dmnerf_mlp_fwd_rays_f16" in _lib.SIGNATURES
    sc = build_scene(w)
    h = headline_leg(a, w, sc)
    rccl = rccl_evidence(w, sc, h) if w.world > 1 else None
    train_multi, mani_multi = multi_rank_legs(a, w, sc)
    if w.rank == 0:
        res = headline_record(a, w, sc, h, rccl)
        if w.world == 1:
            wide = single_gpu_render_legs(a, w, sc, h, res)
            if not a.no_train:
                single_gpu_train_legs(a, w, sc, res, wide)
        if train_multi is not None:
            res["train"] = train_multi
        if mani_multi is not None:
            res["manipulator_frame"] = mani_multi
        lift_scalars(res)
        res["wall_s"] = time.perf_counter() - t_wall
    if w.world > 1:
        import torch.distributed as dist
        dist.barrier()                                          # nobody is still printing
    C.flush_c_stdio()
    if w.rank == 0:
        path = write_full_record(res, a.full_record)
        rel = None if path is None else (os.path.relpath(path, ROOT) if path.startswith(ROOT) else path)
        print(contract_line(res, rel), flush=True)               # the LAST stdout line: the contract, <= 4096 bytes, strict JSON
    if w.world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
