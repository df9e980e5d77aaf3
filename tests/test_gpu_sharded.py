"""The ray-sharded training step (dm_nerf_amd.distributed.sharded_train_step, SURVEY 8(e) / BASELINE config 5) with the
real HIP kernels: two ranks -- two processes sharing the box's one GPU, exchanging through gloo, which runs the same
code path as RCCL up to the transport -- reach the parameters a single process reaches on the whole batch."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

INS, N, N_INS = 13, 131, 70          # 131 rays -> slices of 66 + 65; the last 70 rays carry labels (ScanNet convention)


def _two_steps(flat_adam=False):
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.networks import dm_nerf as M
    models = []
    for seed in (71, 72):
        m = M.DM_NeRF(8, 256, 63, 27, [4], INS)
        m.load_state_dict(O.make_weights(seed, INS, gain=1.7, sigma_bias=0.3))
        models.append(m.cuda().train())
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(35.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(9).choice(480 * 640, N, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]]).cuda()
    z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous().cuda()
    g = torch.Generator().manual_seed(73)
    target = torch.rand(N, 3, generator=g).cuda()
    labels = torch.randint(0, 7, (N_INS,), generator=g).cuda()
    if flat_adam:                                           # the extension optimizer: update + re-pack on the (all-reduced) arena
        from dm_nerf_amd.optim import FlatAdam
        opt = FlatAdam(models, lr=5e-4)
    else:
        opt = torch.optim.SGD([p for m in models for p in m.parameters()], lr=2e-2)
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=N_INS, penalize=True, tolerance=0.05, deta_w=0.05)
    torch.manual_seed(7)
    torch.cuda.manual_seed(7)                               # the jitter stream: identical on every rank
    losses = []
    for _ in range(2):
        loss, nbytes = D.sharded_train_step(rays, z, target, labels, models, args, opt, INS)
        losses.append(float(loss))
    flat = torch.cat([p.detach().reshape(-1) for m in models for p in m.parameters()]).cpu().numpy()
    # multi-rank: the gradients of BOTH models live in one arena that was all-reduced in place (no cat / copy_)
    from dm_nerf_amd import autograd as G
    arena = G.arena_slot(models[0])
    in_place = None if arena is None else bool(arena[0].resident() and arena[0].flat.numel() * 4 == nbytes
                                               and models[1].mlps[0].weight.grad.data_ptr() == arena[0].slots[1].data_ptr())
    return losses, flat, nbytes, in_place


def _worker(rank, world, port, q, flat_adam=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _two_steps(flat_adam))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("flat_adam", [False, True])
def test_two_rank_sharded_training_step_equals_single_process(flat_adam):
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    want_losses, want, nb0, in_place0 = _two_steps(flat_adam)
    assert nb0 == 0 and (in_place0 is None or flat_adam)            # (FlatAdam owns an arena at world 1 too; nothing is all-reduced)
    start = torch.cat([v.reshape(-1) for seed in (71, 72) for v in O.make_weights(seed, INS, gain=1.7, sigma_bias=0.3).values()]).numpy()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, flat_adam)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, losses, flat, nbytes, in_place in res:
        assert in_place is True
        assert nbytes == 4 * want.size
        assert np.allclose(losses, want_losses, rtol=2e-5), (losses, want_losses)
        # SGD: the update is linear in the gradient (2e-6 = lr x the f32 summation-order difference); Adam's first steps are
        # sign-like (m / sqrt(v)): a gradient element that differs in its last bits moves the parameter by up to 2 lr
        assert np.abs(flat - want).max() <= (2e-6 if not flat_adam else 2.1e-3), np.abs(flat - want).max()
        if flat_adam:
            assert np.mean(np.abs(flat - want) <= 2e-6) >= 0.98
    assert np.abs(want - start).max() >= (1e-3 if not flat_adam else 9e-4)
    assert np.array_equal(res[0][2], res[1][2])


def _strict_line(stdout, only_line=True):
    """The LAST stdout line, parsed strictly (no NaN / Infinity), bounded at 4096 bytes -- what the driver's reader sees -- plus the
    full record it points at.  ``only_line``: stdout holds nothing else (the self-launching form filters its ranks' stdout; under a
    bare torch.distributed.run the gloo transport of these dry runs prints its own "[Gloo] Rank ..." banners before it)."""
    import json

    def no_constants(name):
        raise AssertionError(f"non-strict JSON constant {name}")
    lines = [l for l in stdout.strip().split("\n") if l.strip()]
    assert lines and lines[-1].startswith("{"), lines[-3:]              # the contract line is the LAST line
    if only_line:
        assert len(lines) == 1, lines                                   # ... and stdout carries nothing else
    else:
        # (eight ranks print their "[Gloo] Rank r is connected to ..." banners concurrently and the pieces interleave arbitrarily:
        # whatever they look like, none of it may look like a second record)
        assert all("{" not in l and '"metric"' not in l for l in lines[:-1]), lines[:-1]
    assert len(lines[-1].encode()) <= 4096, len(lines[-1].encode())
    r = json.loads(lines[-1], parse_constant=no_constants)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = r["full_record"] if os.path.isabs(r["full_record"]) else os.path.join(root, r["full_record"])
    with open(path) as f:
        full = json.loads(f.read(), parse_constant=no_constants)
    return r, full


def _run_bench(world, steps, scaling, train_steps=1, timeout=800, tmp=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DMNERF_BENCH_ONE_DEVICE="1", DMNERF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # the driver's command line (task statement), with the two environment switches that put every rank on the box's one GPU
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", "1",
           "--train-steps", str(train_steps), "--scaling", scaling, "--full-record", os.path.join(str(tmp), "bench_full.json")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    return _strict_line(p.stdout, only_line=False)


def _run_bench_plain(world, steps, timeout=800, tmp=None):
    """The PLAIN command ``python bench.py --gpus N ...`` with no torchrun environment (what the driver's BENCH run looks like
    at N = 1): bench.py launches its own ranks (bench.self_launch) and relays rank 0's line and the exit code."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DMNERF_BENCH_ONE_DEVICE="1", DMNERF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", "1", "--train-steps", "1",
           "--full-record", os.path.join(str(tmp), "bench_full.json")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    return _strict_line(p.stdout)


def _check_line(rf, world, steps, scaling, rays_expected, gathers):
    r, full = rf                                                     # the contract line and the full record behind it
    assert r["n_gpus"] == world and r["steps"] == steps and r["warmup"] == 1 and r["unit"] == "rays/s" and r["scaling"] == scaling
    per_rank = 4096 if scaling == "weak" else 4096 // world
    assert r["config"]["rays_per_step_per_gpu"] == per_rank
    assert r["config"]["rays_in_timed_region"] == rays_expected
    assert abs(r["value"] - rays_expected / (r["ms_per_step"] * steps * 1e-3)) <= 1e-5 * r["value"]
    assert f"all-gather of the rank's band per frame ({gathers} in the timed region)" in full["config"]["parallelism"]
    assert r["roofline"]["bound"] == "mfma" and 0.0 < r["roofline"]["frac"] <= 1.0
    assert "cpu_baseline" not in r and "cpu_baseline" not in full    # rank 0 at N = 1 only
    # what the process group was: these are DRY RUNS -- N gloo ranks on the box's one GPU -- and the line says so
    rc = r["rccl"]
    assert r["one_device_dry_run"] is True and rc["one_device_dry_run"] is True
    assert rc["backend"] == "gloo" and rc["world_size"] == world and rc["distinct_devices"] == 1 and rc["rccl_version"] is None
    assert [x["rank"] for x in rc["ranks"]] == list(range(world))
    assert all(x["device_index"] == 0 and x["device_name"] for x in rc["ranks"])
    assert sum(x["rays_rendered"] for x in rc["ranks"]) == rays_expected
    assert rc["frame_gathers_timed"] == gathers
    band_rows = -(-480 // world) * 640                               # every rank sends the largest band's buffer: rgb | ins 13 | depth
    assert rc["gather_send_bytes_per_rank"] == band_rows * 17 * 4 and rc["gather_bytes_per_frame"] == world * band_rows * 17 * 4
    t, tc = full["train"], r["train"]
    assert "error" not in t, t
    assert t["batch_rays"] == (4096 * world if scaling == "weak" else 4096) and t["rays_per_s"] > 0 and np.isfinite(t["final_loss"])
    assert t["roofline"]["samples_per_launch"] == t["batch_rays"] // world * 192
    assert abs(r["train_ms_per_step"] - t["ms_per_step"]) <= 1e-5 * t["ms_per_step"]            # top-level scalars
    assert abs(r["train_rays_per_s"] - t["rays_per_s"]) <= 1e-5 * t["rays_per_s"]
    assert 0.0 < r["train_roofline_frac_worst"] <= 1.0
    # per optimisation step: one packed all-gather, the 64-B penalizer sums, the in-place arena all-reduce of both models' gradients
    assert tc["collectives_per_step"] == 3 and tc["collective_kinds_per_step"] == {"all_gather": 1, "all_reduce_sums": 1, "all_reduce_grads": 1}
    assert tc["allreduce_bytes_per_step"] == 2 * 696338 * 4 and tc["rays_this_rank"] == t["batch_rays"] // world


def test_one_device_mode_is_refused_over_rccl():
    """N RCCL ranks on ONE GPU would be recorded as an N-GPU run: bench.py refuses the combination outright."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMNERF_BENCH_ONE_DEVICE="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    env.pop("DMNERF_BENCH_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert p.returncode != 0 and not p.stdout.strip() and "needs DMNERF_BENCH_BACKEND=gloo" in p.stderr


@pytest.mark.timeout(900)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_multi_rank_code_path_prints_a_valid_line(scaling, tmp_path):
    """The driver's SCALE run launches ``bench.py --gpus N`` under torch.distributed.run on an 8-GPU node this build never
    sees: exercise that exact code path here with two ranks sharing the box's one GPU over gloo
    (DMNERF_BENCH_ONE_DEVICE / DMNERF_BENCH_BACKEND) and validate the JSON line -- per-frame band gather, sharded training
    step with the in-place gradient arena, max-over-ranks timing, both scaling modes."""
    r = _run_bench(2, 4, scaling, tmp=tmp_path)
    per_rank = 4096 if scaling == "weak" else 2048
    _check_line(r, 2, 4, scaling, 2 * per_rank * 4, 1)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world", [2, 8])
def test_plain_bench_command_launches_its_own_ranks(world, tmp_path):
    """``python bench.py --gpus N`` WITHOUT torch.distributed.run around it (the form of the driver's N = 1
    command) must not die on the world-size check: it re-launches itself under torch.distributed.run and relays the line."""
    steps = 4 if world == 2 else 10
    r = _run_bench_plain(world, steps, timeout=1400, tmp=tmp_path)
    _check_line(r, world, steps, "weak", (2 * 4096 * 4) if world == 2 else 480 * 640, 1)


def test_plain_bench_command_relays_a_failing_exit_code():
    """... and a failure of the launched ranks is the launcher's exit code, not a silent 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DMNERF_BENCH_ONE_DEVICE="1", DMNERF_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0", "--scaling", "strong",
                        "--no-train"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode != 0 and not p.stdout.strip(), (p.returncode, p.stdout[-500:])     # 4096 rays do not split three ways


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_eight_rank_dry_run_covers_a_whole_band(scaling, tmp_path):
    """The 8-GPU run before it exists: EIGHT ranks on the one GPU over gloo, the driver's command line, and enough steps for one
    whole band per rank -- weak: 60 rows = 38 400 rays = nine 4096-ray chunks and the ragged 1536-ray one (tester.py:65-67), then
    the frame's single all-gather; strong: 75 chunks of 512 rays.  Every ray of the 640 x 480 frame is rendered exactly once in
    the timed region (``rays_in_timed_region`` = 307 200), by 8 ranks, with one gather -- and the sharded training step runs at
    world 8 (weak: a 32 768-ray batch, 4096 per rank; strong: the 4096-ray batch in 512-ray slices)."""
    steps = 10 if scaling == "weak" else 75
    r = _run_bench(8, steps, scaling, timeout=1400, tmp=tmp_path)
    _check_line(r, 8, steps, scaling, 480 * 640, 1)
    assert r[0]["config"]["chunks_per_band"] == steps
    assert r[0]["config"]["ragged_chunk_rays"] == (1536 if scaling == "weak" else 0)
