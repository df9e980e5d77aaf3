"""World-size-2 tests of the ray-sharded multi-GPU logic on CPU (gloo): the sharded result must equal
the single-process result.  The HIP kernels are not involved: the chunk renderer / ray generator are
injected (the CPU oracle plays the renderer), exactly the hooks dm_nerf_amd.distributed exposes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dm_nerf_amd import distributed as D
from oracle import ref_cpu as O

H, W, CHUNK, INS = 10, 12, 32, 5          # 120 rays, bands of 5 rows = 60 rays -> chunks 32 + 28 (ragged)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene():
    K = O.dmsr_intrinsics(H, W)
    c2w = O.pose_spherical(30.0, -65.0, 7.0)
    sd_c = O.make_weights(1, INS, gain=1.7, sigma_bias=0.3)
    sd_f = O.make_weights(2, INS, gain=1.7, sigma_bias=0.3)
    return K, c2w, sd_c, sd_f


def _raygen(H_, W_, K, c2w, row0, nrows):
    o, d = O.get_rays_k(H_, W_, K, c2w)
    return o[row0:row0 + nrows].contiguous(), d[row0:row0 + nrows].contiguous()


def _render_chunk(rays_o, rays_d, z, models, args):
    with torch.no_grad():
        out = O.dm_nerf(torch.stack([rays_o, rays_d]), models[0], models[1], z, perturb=0., N_importance=16)
    return out['rgb_fine'], out['ins_fine'], out['depth_fine']


def _z_fn(n, dev):
    return O.z_val_sample(n, 4.0, 15.0, 16).contiguous()


def _frame(h=H, labels_only=False):
    K, c2w, sd_c, sd_f = _scene()
    kw = dict(labels_only=True, label_conf=lambda x: (x.argmax(-1), x.max(-1).values)) if labels_only else {}
    return D.render_frame(h, W, K, c2w, (sd_c, sd_f), 4.0, 15.0, None, chunk=CHUNK, n_samples=16,
                          raygen=_raygen, render_chunk=_render_chunk, z_fn=_z_fn, ins_num=INS, **kw)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []
        real_gather = dist.all_gather
        dist.all_gather = lambda *a, **k: (calls.append(1), real_gather(*a, **k))[1]
        rgb, ins, depth = _frame()
        n_coll = len(calls)                                  # ONE collective per frame: the packed band
        # bands of unequal height (7 rows over 2 ranks = 4 + 3): padded gather + one indexed row copy; labels_only packing
        odd = _frame(7)
        odd_lab = _frame(7, labels_only=True)
        n_coll_odd = len(calls) - n_coll
        dist.all_gather = real_gather
        # gradient bucket all-reduce: rank r contributes (r+1) * ones
        lin = [torch.nn.Linear(3, 2), torch.nn.Linear(2, 1)]
        for m in lin:
            for p in m.parameters():
                p.grad = torch.full_like(p, float(rank + 1))
        nbytes = D.allreduce_grads(lin)
        g_ok = all(bool((p.grad == 3.0).all()) for m in lin for p in m.parameters())
        # a rank WITHOUT a gradient for some tensor (its slice produced none) must still contribute the same number of
        # elements: the bucket spans all parameters, missing gradients count as zero (unequal buckets hang)
        for m in lin:
            for p in m.parameters():
                p.grad = torch.full_like(p, float(rank + 1))
        if rank == 1:
            lin[1].bias.grad = None
            lin[0].weight.grad = None
        nb2 = D.allreduce_grads(lin)
        g_ok = g_ok and nb2 == nbytes and bool((lin[1].bias.grad == 1.0).all()) and bool((lin[0].weight.grad == 1.0).all()) \
            and bool((lin[0].bias.grad == 3.0).all())
        # a tensor NO rank has a gradient for keeps grad = None (single-process semantics: the optimizer skips it), and a
        # parameter that does not require a gradient takes no part at all
        for m in lin:
            for p in m.parameters():
                p.grad = torch.full_like(p, float(rank + 1))
        lin[1].weight.grad = None
        lin[0].bias.requires_grad_(False)
        lin[0].bias.grad = None
        nb3 = D.allreduce_grads(lin)
        g_ok = g_ok and lin[1].weight.grad is None and lin[0].bias.grad is None and bool((lin[0].weight.grad == 3.0).all()) \
            and nb3 == nbytes - 2 * 4
        lin[0].bias.requires_grad_(True)
        # ray-sharded training step == single-process gradients (mean-squared error over the global batch)
        torch.manual_seed(0)
        lin2 = torch.nn.Linear(3, 3)
        g = torch.Generator().manual_seed(5)
        xs, ys = torch.randn(10, 3, generator=g), torch.randn(10, 3, generator=g)
        s0, cnt = D.ray_slice(10, rank, world)
        D.data_parallel_backward(((lin2(xs[s0:s0 + cnt]) - ys[s0:s0 + cnt]) ** 2).sum(), [lin2], cnt, 10)
        ref = torch.nn.Linear(3, 3)
        ref.load_state_dict(lin2.state_dict())
        (((ref(xs) - ys) ** 2).sum() / 10).backward()
        g_ok = g_ok and all(torch.allclose(p.grad, q.grad, atol=1e-6) for p, q in zip(lin2.parameters(), ref.parameters()))
        # uneven all_gather_cat
        t = torch.arange(rank + 2, dtype=torch.float32)[:, None] + 10 * rank
        cat = D.all_gather_cat(t, sizes=[2, 3])
        q.put((rank, rgb.numpy(), ins.numpy(), depth.numpy(), g_ok, nbytes, cat.numpy(), n_coll, n_coll_odd,
               [t.numpy() for t in odd], [t.numpy() for t in odd_lab]))
    finally:
        dist.destroy_process_group()


def test_row_bands_partition_the_frame():
    for Hh, world in ((480, 8), (10, 3), (7, 8), (480, 1)):
        bands = [D.row_band(Hh, r, world) for r in range(world)]
        assert bands[0][0] == 0 and sum(n for _, n in bands) == Hh
        for (a, n), (b, _) in zip(bands[:-1], bands[1:]):
            assert a + n == b
        assert max(n for _, n in bands) - min(n for _, n in bands) <= 1
    assert D.row_band(480, 3, 8) == (180, 60)            # 60 rows = 38 400 rays per rank


def test_two_stream_backward_is_chosen_where_it_removes_a_partial_round(monkeypatch):
    """distributed.overlap_enabled: 128-sample workgroups on 256 CUs -- the 384-ray shard of an 8-way split is 576 + 192 workgroups
    (3 + 1 rounds one after the other, 3 side by side); 512, 3072 and 4096 rays are whole rounds either way."""
    monkeypatch.delenv("DMNERF_OVERLAP_BWD", raising=False)
    dev = torch.device("cpu")
    pick = lambda n: D.overlap_enabled(n, 64, 192, dev, cus=256)
    assert pick(384) and pick(96) and pick(64)
    assert not pick(131)                                            # 197 + 66 workgroups: 1 + 1 rounds one after the other, 2 side by side
    assert not pick(512) and not pick(1024) and not pick(3072) and not pick(4096)
    assert D.overlap_enabled(384, 64, 192, dev) is False             # no GPU: nothing to overlap
    monkeypatch.setenv("DMNERF_OVERLAP_BWD", "1")
    assert D.overlap_enabled(4096, 64, 192, dev) is True
    monkeypatch.setenv("DMNERF_OVERLAP_BWD", "0")
    assert D.overlap_enabled(384, 64, 192, dev, cus=256) is False


def test_single_process_frame_matches_direct_oracle():
    rgb, ins, depth = _frame()
    K, c2w, sd_c, sd_f = _scene()
    o, d = O.get_rays_k(H, W, K, c2w)
    with torch.no_grad():
        want = O.dm_nerf(torch.stack([o.reshape(-1, 3), d.reshape(-1, 3)]), sd_c, sd_f, _z_fn(H * W, None),
                         perturb=0., N_importance=16)
    assert rgb.shape == (H, W, 3) and ins.shape == (H, W, INS) and depth.shape == (H, W)
    assert torch.allclose(rgb.reshape(-1, 3), want['rgb_fine'], atol=1e-5)


def test_render_path_poses_crop_and_psnr():
    """The pose loop with a crop mask and ground truth: frames equal render_frame's, the crop keeps the masked pixels
    in order, PSNR is that of the cropped frame (oracle as the chunk renderer, single process)."""
    import types
    K, c2w, sd_c, sd_f = _scene()
    poses = torch.stack([c2w, O.pose_spherical(60.0, -65.0, 7.0)])
    mask = torch.zeros(H, W, dtype=torch.int64)
    mask[2:8, 3:11] = 1                                           # a 6 x 8 window
    args = types.SimpleNamespace(N_test=CHUNK, N_samples=16, near=4.0, far=15.0, crop_height=6, crop_width=8)
    gt = torch.rand(2, 6, 8, 3, generator=torch.Generator().manual_seed(1))
    kw = dict(raygen=_raygen, render_chunk=_render_chunk, z_fn=_z_fn)
    out = D.render_path(poses, (H, W, K), (sd_c, sd_f), args, gt_imgs=gt, crop_mask=mask, **kw)
    assert out["rgb"].shape == (2, 6, 8, 3) and out["ins"].shape == (2, 6, 8, INS) and out["depth"].shape == (2, 6, 8)
    full = D.render_frame(H, W, K, poses[1], (sd_c, sd_f), 4.0, 15.0, args, chunk=CHUNK, n_samples=16, **kw)
    assert torch.equal(out["rgb"][1], full[0][2:8, 3:11]) and torch.equal(out["depth"][1], full[2][2:8, 3:11])
    want = -10 * torch.log10(((full[0][2:8, 3:11] - gt[1]) ** 2).mean())
    assert torch.allclose(out["psnr"][1], want)
    lab = D.render_path(poses[:1], (H, W, K), (sd_c, sd_f), args, labels_only=True,
                        label_conf=lambda x: (x.argmax(-1), x.max(-1).values), **kw)
    assert lab["label"].shape == (1, H, W) and lab["label"].dtype == torch.int64
    assert torch.equal(lab["label"][0], D.render_frame(H, W, K, poses[0], (sd_c, sd_f), 4.0, 15.0, args, chunk=CHUNK,
                                                       n_samples=16, **kw)[1].argmax(-1))


@pytest.mark.timeout(300)
def test_two_rank_sharded_frame_equals_single_process():
    single = [t.numpy() for t in _frame()]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    odd1 = [t.numpy() for t in _frame(7)]
    lab1 = [t.numpy() for t in _frame(7, labels_only=True)]
    for rank, rgb, ins, depth, g_ok, nbytes, cat, n_coll, n_coll_odd, odd, odd_lab in res:
        # rays are independent: sharding must not change a single bit
        assert np.array_equal(rgb, single[0]) and np.array_equal(ins, single[1]) and np.array_equal(depth, single[2])
        assert n_coll == 1 and n_coll_odd == 2                         # one all-gather per frame, whatever the band heights
        assert all(np.array_equal(a, b) for a, b in zip(odd, odd1))
        assert all(np.array_equal(a, b) for a, b in zip(odd_lab, lab1)) and odd_lab[1].dtype == np.int64
        assert g_ok and nbytes == (3 * 2 + 2 + 2 + 1) * 4
        assert np.array_equal(cat[:, 0], np.array([0, 1, 10, 11, 12], dtype=np.float32))


# ---- ray-sharded training step (BASELINE config 5) --------------------------------------------------
TN, TS, TIMP, TNINS = 25, 16, 16, 10      # 25 rays -> slices of 13 + 12; only the last 10 rays carry labels


class _SD:
    """The oracle's weights (a state_dict of leaf tensors) behind the two methods the step needs."""

    def __init__(self, sd):
        self.sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}

    def parameters(self):
        return list(self.sd.values())


def _train_two_steps(penalize=True):
    import types
    K, c2w, sd_c, sd_f = _scene()
    o, d = O.get_rays_k(H, W, K, c2w)
    sel = torch.from_numpy(np.random.RandomState(3).choice(H * W, TN, replace=False))
    rays = torch.stack([o.reshape(-1, 3)[sel], d.reshape(-1, 3)[sel]])
    z = O.z_val_sample(TN, 4.0, 15.0, TS).contiguous()
    g = torch.Generator().manual_seed(11)
    target = torch.rand(TN, 3, generator=g)
    labels = torch.randint(0, 4, (TNINS,), generator=g)
    mc, mf = _SD(sd_c), _SD(sd_f)
    # (plain gradient steps: the update is proportional to the gradient, so parameter differences measure gradient
    # differences; Adam would turn float noise on near-zero gradients into +-lr steps)
    opt = torch.optim.SGD(mc.parameters() + mf.parameters(), lr=2e-2)
    # args.penalize: the reference adds the emptiness term only under --penalize (train_dmsr.py:51; default off, tolerance /
    # deta_w then None): with it off the step must not touch the penalizer at all
    args = types.SimpleNamespace(perturb=1.0, N_importance=TIMP, is_train=True, N_ins=TNINS, penalize=penalize,
                                 tolerance=0.05 if penalize else None, deta_w=0.05 if penalize else None)
    sizes = [D.ray_slice(TN, r, D.world_info()[1])[1] for r in range(D.world_info()[1])]

    seen = []

    def render(r, zz, a, tr, uu):
        assert a.N_ins is None                               # the label slice belongs to the gathered batch
        seen.append((tr.clone(), uu.clone()))                # the jitter rows this rank was handed
        return O.dm_nerf(r, mc.sd, mf.sd, zz, perturb=1., N_importance=TIMP, is_train=True, N_ins=None, t_rand=tr, u=uu)

    def penalizer(out, lvl, rays_d):                         # exact batch-global semantics through gather_batch
        assert penalize, "the emptiness penalizer must only run under args.penalize"
        full = [D.gather_batch(t, sizes) for t in (out['raw_' + lvl], out['z_vals_' + lvl], out['depth_' + lvl], rays_d)]
        return O.ins_penalizer(full[0], full[1], full[2], full[3], 0.05, 0.05)

    torch.manual_seed(7)                                     # the jitter stream: identical on every rank
    losses = []
    calls = {"gather": 0, "reduce": 0}
    real_gather, real_reduce = dist.all_gather, dist.all_reduce
    if dist.is_initialized():
        dist.all_gather = lambda *a, **k: (calls.__setitem__("gather", calls["gather"] + 1), real_gather(*a, **k))[1]
        dist.all_reduce = lambda *a, **k: (calls.__setitem__("reduce", calls["reduce"] + 1), real_reduce(*a, **k))[1]
    D.collective_tally(reset=True)
    try:
        losses = _two_steps(rays, z, target, labels, mc, mf, args, opt, render, penalizer)
    finally:
        dist.all_gather, dist.all_reduce = real_gather, real_reduce
    calls["tally"] = D.collective_tally(reset=True)          # the module's own count (what bench.py reports per step at N > 1)
    flat = torch.cat([p.detach().reshape(-1) for p in mc.parameters() + mf.parameters()])
    draws = [torch.cat([a, b], 1).numpy() for a, b in seen]  # per step: [n_local, S + N_importance]
    return losses, flat.numpy(), _two_steps.nbytes, calls, draws


def _two_steps(rays, z, target, labels, mc, mf, args, opt, render, penalizer):
    losses = []
    for _ in range(2):
        loss, nbytes = D.sharded_train_step(rays, z, target, labels, (mc, mf), args, opt, INS, render=render,
                                            mse=lambda a, b: ((a - b) ** 2).mean(),
                                            criterion=lambda p, gt: O.ins_criterion(p, gt, INS)[0].sum(), penalizer=penalizer)
        losses.append(float(loss))
    _two_steps.nbytes = nbytes
    return losses


def _train_worker(rank, world, port, q, penalize=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _train_two_steps(penalize))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("penalize", [True, False])
def test_two_rank_sharded_training_equals_single_process(penalize):
    """Two gradient steps with jitter, a partially labelled batch (N_ins) and uneven slices: both ranks end with the
    parameters a single process reaches on the whole batch (differences: f32 summation order of the all-reduce).
    With and without the optional emptiness term (``args.penalize``, train_dmsr.py:51-58)."""
    want_losses, want, nb0, _, draws1 = _train_two_steps(penalize)
    assert nb0 == 0
    _, _, sd_c, sd_f = _scene()
    start = torch.cat([v.reshape(-1) for v in list(sd_c.values()) + list(sd_f.values())]).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q, penalize)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n_param = want.size
    # ray i's jitter does not depend on the world size: the ranks' rows, in rank order, ARE the single-process draws (full-size
    # draws from the identically seeded generator, then sliced; render.py:46 / helpers.py:135 order)
    by_rank = sorted(res, key=lambda r: r[0])
    for step in range(2):
        assert np.array_equal(np.concatenate([r[5][step] for r in by_rank], 0), draws1[step])
    for rank, losses, flat, nbytes, calls, _ in res:
        assert nbytes == 4 * n_param                        # ONE flat bucket with both models' gradients
        if not penalize:                                    # (the injected penalizer of this test gathers on its own)
            # per step: ONE packed all-gather (rgb | ins of both levels) and ONE gradient all-reduce
            tally = calls.pop("tally")
            assert calls == {"gather": 2, "reduce": 2}, calls
            # ... and the module's own tally (bench.py's `collectives_per_step` / `allreduce_bytes_per_step`) says the same
            assert tally["count"] == 4 and tally["kinds"] == {"all_gather": 2, "all_reduce_grads": 2}, tally
            assert tally["bytes"] >= 2 * 4 * n_param
        assert np.allclose(losses, want_losses, rtol=1e-5), (losses, want_losses)
        assert np.abs(flat - want).max() <= 1e-6, np.abs(flat - want).max()
        assert np.abs(flat - start).max() >= 1e-3           # ... of steps that did move the weights
    assert np.array_equal(res[0][2], res[1][2])            # the replicas stay bit-identical


# ---- the 8-GPU node's world size (and a non-divisor) on CPU ---------------------------------------------------------
def _wide_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames = {h: [t.numpy() for t in _frame(h)] for h in (H, 5)}       # 10 rows: uneven bands; 5 rows: some ranks own NO row
        lab = [t.numpy() for t in _frame(5, labels_only=True)]
        losses, flat, nbytes, calls, draws = _train_two_steps(False)
        q.put((rank, frames, lab, losses, flat, nbytes, calls, draws))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [8, 7])
def test_wide_worlds_frame_and_training_equal_single_process(world):
    """World size 8 (the node the scaling bench runs on) and 7 (divides nothing): row bands of 2/1 rows, a 5-row frame on which
    ranks 5.. own no ray at all and still take part in the frame's one all-gather (that once raised on the empty rank
    and hung the others), ray slices of 4/3 rays with the labelled tail (N_ins) spread over the last ranks.  Every rank ends with
    the same frame bit for bit (the single-process one to the oracle's chunk-position ulp) and the single-process parameters to summation order."""
    single = {h: [t.numpy() for t in _frame(h)] for h in (H, 5)}
    lab1 = [t.numpy() for t in _frame(5, labels_only=True)]
    want_losses, want, _, _, draws1 = _train_two_steps(False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wide_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    by_rank = sorted(res, key=lambda r: r[0])
    for step in range(2):                                                   # jitter of ray i independent of the world size
        assert np.array_equal(np.concatenate([r[7][step] for r in by_rank], 0), draws1[step])
    for rank, frames, lab, losses, flat, nbytes, calls, _ in res:
        for h in (H, 5):
            # (the renderer injected here is the CPU oracle, whose vectorised sigmoid / exp are not bit-invariant to where a ray
            # falls in a chunk: one ulp on a few elements when the chunk sizes differ from the single-process ones.  The sharding
            # itself moves bytes: every rank holds the same frame, bit for bit.)
            assert all(np.allclose(a, b, rtol=0, atol=2.5e-7) for a, b in zip(frames[h], single[h])), (rank, h)
            assert all(np.array_equal(a, b) for a, b in zip(frames[h], res[0][1][h])), (rank, h)
        assert np.array_equal(lab[1], lab1[1]) and lab[1].dtype == np.int64
        assert all(np.allclose(a, b, rtol=0, atol=2.5e-7) for a, b in zip(lab, lab1))
        assert calls.pop("tally")["kinds"] == {"all_gather": 2, "all_reduce_grads": 2}
        assert calls == {"gather": 2, "reduce": 2}, calls
        assert np.allclose(losses, want_losses, rtol=1e-5), (losses, want_losses)
        assert np.abs(flat - want).max() <= 2e-6, np.abs(flat - want).max()
        assert np.array_equal(flat, res[0][4])                              # replicas bit-identical
    for r in range(world):                                                  # the index helpers at these sizes
        assert D.ray_slice(TN, r, world)[1] in (TN // world, TN // world + 1)
    idx = D._compact_index([D.row_band(5, r, world)[1] * W for r in range(world)], W, "cpu")
    assert idx.numel() == 5 * W and torch.equal(idx, torch.arange(5 * W))   # (ranks 0..4 own one row each, contiguous at the front)


# ---------------------------------------------------------------------------------------------------------------------
# the manipulation render's frame driver (distributed.ManipulationFrameRenderer; networks/manipulator.py:232-270)
# ---------------------------------------------------------------------------------------------------------------------
MH, MW, MCHUNK, MINS, MIMP = 10, 12, 32, 5, 8          # 120 rays: chunks of 32, 32, 32 and a ragged 24 that straddle the bands


def _mani_scene():
    K = O.dmsr_intrinsics(MH, MW)
    pose = O.pose_spherical(75.0, -65.0, 7.0)
    trans = [torch.tensor([[1., 0., 0., 0.3], [0., 1., 0., -0.2], [0., 0., 1., 0.1], [0., 0., 0., 1.]]),
             torch.tensor([[0., -1., 0., 0.], [1., 0., 0., 0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]])]
    sd_c = O.make_weights(31, MINS, W=32, **O.PEAKY)
    sd_f = O.make_weights(32, MINS, W=32, **O.PEAKY)
    return K, pose, trans, sd_c, sd_f


def _mani_chunk(ori, tars, models, args, us):
    with torch.no_grad():
        return O.manipulator(models[0], models[1], ori, list(tars), 8, MIMP, 4.0, 15.0, args.target_labels, us=us)


def _mani_chunk_exact(ori, tars, models, args, us):
    """A stand-in chunk renderer built from single IEEE operations only (bitwise independent of which rows share a call -- the CPU
    oracle's MKL GEMMs are not: 1 ulp between a 24-row and a 32-row call), touching every input the driver routes: original rays,
    each target's rays, each of the 2 + T draws."""
    C = MINS + 1
    mix = sum(u[:, :3] for u in us)
    return (ori[0] + ori[1] * us[0][:, :3], us[-1][:, :C] * ori[1][:, :1] + us[1][:, 1:C + 1],
            tars[-1][0] * mix + tars[-1][1], us[len(tars)][:, :C] - tars[0][1][:, 2:3])


def _mani_frame(T, log=None, exact=False, **kw):
    import types
    K, pose, trans, sd_c, sd_f = _mani_scene()
    gen = torch.Generator().manual_seed(77)                 # every rank owns an identically seeded generator, as on the device

    def draws(n, n_imp, count, dev):
        us = [torch.rand(n, n_imp, generator=gen) for _ in range(count)]
        if log is not None:
            log.append((n, count))
        return us
    args = types.SimpleNamespace(N_samples=8, N_importance=MIMP, near=4.0, far=15.0, N_test=MCHUNK, target_label=2)
    return D.manipulate_frame(MH, MW, K, pose, trans[:T], (sd_c, sd_f), args, raygen=_raygen,
                              manipulate_chunk=_mani_chunk_exact if exact else _mani_chunk, draws=draws, ins_num=MINS, **kw)


def _mani_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []
        real_gather = dist.all_gather
        dist.all_gather = lambda *a, **k: (calls.append(1), real_gather(*a, **k))[1]
        log = []
        f1 = _mani_frame(1, log)
        n1 = len(calls)
        f2 = _mani_frame(2)
        n2 = len(calls) - n1
        e1, e2 = _mani_frame(1, exact=True), _mani_frame(2, exact=True)
        dist.all_gather = real_gather
        q.put((rank, [t.numpy() for t in f1], [t.numpy() for t in f2], n1, n2, log, [t.numpy() for t in e1], [t.numpy() for t in e2]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 7])
def test_sharded_manipulation_frame_equals_single_process(world):
    """World 2 (bands of 5 rows = 60 rays: the chunks of 32 straddle the band boundary) and world 7 (bands of 2 / 1 rows, ranks that
    own a fraction of one chunk): the gathered frame equals the single-process frame -- every rank makes the ``2 + T`` draws of
    EVERY chunk of the frame in the reference's order and uses its rows -- with ONE collective per frame.  Bit for bit with a chunk
    renderer that is bitwise row-independent (as the HIP kernels are: tests/test_gpu_manipulator_frame.py); with the CPU oracle's
    ``manipulator`` as the renderer to 1 ulp (its GEMMs round differently for different row counts) and with identical labels."""
    torch.set_num_threads(1)
    want1, want2 = _mani_frame(1), _mani_frame(2)
    wante1, wante2 = _mani_frame(1, exact=True), _mani_frame(2, exact=True)
    assert want1[0].shape == (MH, MW, 3) and want1[1].shape == (MH, MW, MINS + 1) and want1[3].shape == (MH, MW, MINS + 1)
    assert len(np.unique(want1[1].argmax(-1).numpy())) >= 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mani_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, f1, f2, n1, n2, log, e1, e2 in res:
        assert n1 == 1 and n2 == 1                                                   # ONE all-gather per frame
        assert log == [(32, 3), (32, 3), (32, 3), (24, 3)]                           # every rank: all four chunks, 2 + T draws each
        for got, want in zip(e1 + e2, wante1 + wante2):
            assert np.array_equal(got, want.numpy())
        for got, want in zip(f1 + f2, want1 + want2):
            assert np.abs(got - want.numpy()).max() <= 2.5e-7                        # (a misrouted draw or ray moves values by 1e-2 .. 1)
        for f, w in ((f1, want1), (f2, want2)):
            assert np.array_equal(f[1].argmax(-1), w[1].argmax(-1).numpy())


def test_manipulation_frame_band_override_renders_rows_of_the_frame():
    """``rank=`` / ``world=``: one process renders band r of N without a process group; its rows are the frame's rows."""
    whole = _mani_frame(1, exact=True)
    for world, rank in ((3, 1), (4, 3)):
        r0, nr = D.row_band(MH, rank, world)
        part = _mani_frame(1, exact=True, rank=rank, world=world)
        for got, want in zip(part, whole):
            assert got.shape[0] == nr and torch.equal(got, want[r0:r0 + nr])
