"""One process per GPU (torchrun), RCCL over xGMI via ``torch.distributed`` (backend "nccl" on ROCm).

Rays are independent units, so the path shards with no data-path collective:

* render (reference: the serial chunk loop of ``render_test``, networks/tester.py:55-85): rank r owns
  a contiguous band of image rows, generates its own rays (no scatter), renders them in chunks,
  and ONE all-gather per frame of one packed band buffer assembles ``rgb [H,W,3]``, ``ins [H,W,ins_num]``,
  ``depth [H,W]`` (``FrameRenderer``; replaces the O(chunks^2) ``torch.cat`` accumulation, tester.py:73-77).
* training: every rank renders a slice of the same ray batch; ``allreduce_grads`` sums the gradients
  of both models in one flat 5.57 MB bucket per step (after ``total_loss.backward()``,
  train_dmsr.py:63); ``all_gather_cat`` exchanges the small per-ray outputs that batch-global
  losses (Hungarian matching, soft-IoU; networks/evaluator.py:19-74) need.

Nothing here touches kernels: the collectives move finished tiles / gradients only.  The chunk
renderer and ray generator are injectable so the sharding logic is tested on CPU with gloo.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (idempotent).
    Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(local)
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = {"device_id": device} if use_cuda else {}
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world, **kw)
    return rank, world, device


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


_force = [False]


def force_collectives(on=None):
    """``DMNERF_FORCE_COLLECTIVES=1`` (or ``force_collectives(True)``): issue every collective of this module even in a world of
    ONE rank, where they are otherwise skipped as no-ops.  A one-GPU box can then drive the exact RCCL calls of the 8-GPU run --
    the packed all-gather, the 64-B all-reduce, the in-place arena all-reduce, the frame gather -- through the product code with
    ``init_process_group("nccl", world_size=1, device_id=...)`` and compare against the skipped path bit for bit
    (tests/test_gpu_rccl.py).  Needs an initialised process group; returns the current setting."""
    if on is not None:
        _force[0] = bool(on)
    return _force[0] or os.environ.get("DMNERF_FORCE_COLLECTIVES") == "1"


def _active(world):
    """Whether collectives are issued: more than one rank, or forced in an initialised world of one."""
    return world > 1 or (force_collectives() and dist.is_available() and dist.is_initialized())


# Tally of the collectives this module issued (host-side counters, no synchronisation): what a bench line reports as
# ``collectives_per_step`` / bytes so that a multi-rank record says what crossed the links, not only how long it took.
_tally = {"count": 0, "bytes": 0, "kinds": {}}


def _count(kind, t):
    _tally["count"] += 1
    _tally["bytes"] += t.numel() * t.element_size()
    _tally["kinds"][kind] = _tally["kinds"].get(kind, 0) + 1


def collective_tally(reset=False):
    """``{"count", "bytes", "kinds"}`` of the collectives issued through this module since the last reset (``bytes`` = the size of
    each call's local send buffer; an all-gather delivers ``world`` times that to every rank)."""
    out = {"count": _tally["count"], "bytes": _tally["bytes"], "kinds": dict(_tally["kinds"])}
    if reset:
        _tally.update(count=0, bytes=0, kinds={})
    return out


def row_band(H, rank, world):
    """Contiguous band of image rows of rank ``rank``: (row0, nrows); bands differ by at most one row."""
    base, rem = divmod(int(H), int(world))
    row0 = rank * base + min(rank, rem)
    return row0, base + (1 if rank < rem else 0)


def ray_slice(N, rank, world):
    """Contiguous slice of a ray batch (training): (start, count)."""
    return row_band(N, rank, world)


def all_gather_cat(t, sizes=None):
    """Concatenate ``t`` (dim 0) over ranks.  ``sizes``: per-rank dim-0 lengths when they differ."""
    rank, world = world_info()
    if not _active(world):
        return t
    t = t.contiguous()

    # The collective is chosen UP FRONT from the backend, identically on every rank -- never by catching an exception
    # (a failure on some ranks only would make the ranks issue different collectives and hang; a real RCCL error must
    # surface): RCCL ("nccl") has the single-buffer form, gloo gathers into a list.
    tensor_form = dist.get_backend() == "nccl"

    def gather(x):                                       # [n, ...] -> [world * n, ...]
        if tensor_form:
            out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(out, x)          # one RCCL all-gather
            _count("all_gather", x)
            return out
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x)
        _count("all_gather", x)
        return torch.cat(parts, 0)

    if sizes is None or len(set(sizes)) == 1:
        return gather(t)
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = gather(pad)
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], 0)


def allreduce_grads(models, average=False, arena=None):
    """Sum (or average) the gradients of ``models`` across ranks in ONE flat message (5.57 MB at ins_num 13).

    With a ``GradArena`` (dm_nerf_amd.autograd) whose slots the backward kernels filled, the parameters' ``.grad`` ARE
    views of one buffer and that buffer is all-reduced in place: no ``cat``, no copies.  Otherwise the bucket is built
    over all parameters that require a gradient, zero-filling those this rank has none for, so that every rank contributes
    the same number of elements whatever it rendered (a rank whose slice produced no gradient for some tensor must not shrink
    its bucket); one flag per parameter rides in the same message, and a tensor no rank had a gradient for keeps ``grad = None``
    (single-process semantics: the optimizer skips it).  The result is copied back."""
    rank, world = world_info()
    if not _active(world):
        return 0
    if arena is not None:
        from . import autograd
        autograd.join_side()                             # the gradients must be complete on this stream before they are reduced
    if arena is not None and arena.resident():
        dist.all_reduce(arena.flat, op=dist.ReduceOp.SUM)
        _count("all_reduce_grads", arena.flat)
        if average:
            arena.flat /= world
        return arena.flat.numel() * arena.flat.element_size()
    # parameters that do not require a gradient take no part (identically on every rank: requires_grad is model structure)
    params = [p for m in models for p in m.parameters() if p.requires_grad]
    if not params:
        return 0
    # one flat message: the gradients (zeros where this rank has none, so that every rank contributes the same count whatever
    # it rendered) followed by one flag per parameter -- after the sum a flag of zero means NO rank had a gradient for that
    # tensor, and it stays None as in a single process (an optimizer skips it; zero-filling would let Adam's moments or weight
    # decay move a parameter nobody differentiated)
    # The message is formed in the widest floating type among the parameters (f32 for the shipped models; f64 parameters -- the
    # CPU / gloo path, oracle-style double models -- are summed in f64 as a single process would); the flags are small integers,
    # exact in any of them.  (Reading the flags back is a host synchronisation: this path is the CPU / injected-renderer one;
    # the HIP path reduces the arena above, in place and graph-capturably.)
    wide = params[0].dtype
    for p in params[1:]:
        wide = torch.promote_types(wide, p.dtype)
    if not wide.is_floating_point:
        wide = torch.float32
    has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=wide, device=params[0].device)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(wide) for p in params] + [has])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    _count("all_reduce_grads", flat)
    n_grad = flat.numel() - len(params)
    any_rank = flat[n_grad:].tolist()
    if average:
        flat[:n_grad] /= world
    o = 0
    for p, flag in zip(params, any_rank):
        n = p.numel()
        if flag > 0:
            if p.grad is None:
                p.grad = flat[o:o + n].view_as(p).to(p.dtype).clone()
            else:
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
    return n_grad * flat.element_size()


def data_parallel_backward(loss_local, models, n_local, n_global):
    """Ray-sharded training step: every rank rendered ``n_local`` of the ``n_global`` rays of the SAME
    batch and formed ``loss_local`` as a SUM over its rays of per-ray terms (e.g. ``((rgb - target)**2).sum()``).
    Scales it so that the all-reduced gradients equal those of the mean over the global batch, runs backward and
    sums the gradients over ranks in one bucket.  Batch-global losses (Hungarian cost, penalizer normalisers) need
    ``all_gather_cat`` of the small per-ray outputs first -- they are not a sum of per-ray terms."""
    del n_local
    (loss_local / float(n_global)).backward()
    return allreduce_grads(models)


class _GatherBatch(torch.autograd.Function):
    """Forward: the full batch tensor on every rank (one all-gather of the detached local slices).
    Backward: the gradient w.r.t. the full tensor is identical on every rank (the losses that consume it are
    evaluated redundantly); each rank keeps the rows of its own slice."""

    @staticmethod
    def forward(ctx, local, sizes):
        rank, world = world_info()
        ctx.s0, ctx.cnt = sum(sizes[:rank]), sizes[rank]
        return all_gather_cat(local.detach(), sizes)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.s0:ctx.s0 + ctx.cnt].contiguous(), None


def gather_batch(local, sizes):
    """``[n_local, ...] -> [N, ...]`` on every rank, differentiable w.r.t. the local rows (the small
    per-ray outputs -- rgb ``[N,3]``, ins ``[N,ins_num]`` -- that batch-global losses need: ≈0.2 MB per step)."""
    rank, world = world_info()
    if not _active(world):
        return local
    return _GatherBatch.apply(local, list(sizes))


def allreduce_sums(t):
    """In-place sum over ranks of a small tensor of batch-global partial sums (the emptiness penalizer's mask
    normalisers, networks/penalizer.py:43,52).  No-op in a single process."""
    rank, world = world_info()
    if _active(world):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        _count("all_reduce_sums", t)
    return t


def overlap_enabled(n_local, n_coarse, n_fine, device, cus=None):
    """Whether the step runs the two levels' network backwards on two streams (autograd.overlapped_backward).

    Worth it exactly when it removes a partial round: the data-gradient kernel runs one 128-sample workgroup per CU at a time, so
    the fine launch takes ceil(wg_fine / CUs) rounds and the coarse one ceil(wg_coarse / CUs) after it; side by side they take
    ceil((wg_fine + wg_coarse) / CUs).  At the 384-ray shard of an 8-way split: 576 + 192 workgroups on 256 CUs, 3 + 1 rounds
    against 3 (measured, profiles/r04: 3.88 -> 3.72 ms per step); at 512 / 3072 / 4096 rays both launches are whole rounds and two
    streams only add contention (+1 .. 2 %, same measurement), so they stay on one.  ``DMNERF_OVERLAP_BWD`` = 0 / 1 forces it."""
    env = os.environ.get("DMNERF_OVERLAP_BWD")
    if env is not None:
        return env != "0"
    if cus is None:
        if device.type != "cuda":
            return False
        cus = torch.cuda.get_device_properties(device).multi_processor_count
    wg_f, wg_c = -(-n_local * n_fine // 128), -(-n_local * n_coarse // 128)
    rounds = lambda w: -(-w // cus)
    return rounds(wg_f) + rounds(wg_c) > rounds(wg_f + wg_c)


def sharded_train_step(rays, z_vals, target, labels, models, args, optimizer, ins_num,
                       render=None, mse=None, criterion=None, penalizer=None, t_rand=None, u=None):
    """One optimisation step of train_dmsr.py:26-64 with the ray batch sharded over the ranks, producing the SAME
    update as a single process on the whole batch (BASELINE config 5):

    * every rank holds the same ``rays [2,N,3]``, ``z_vals [N,S]``, ``target [N,3]``, ``labels`` (the loaders draw the
      batch from a numpy stream that is seeded identically on every rank) and renders rows ``ray_slice(N)``;
      with ``args.perturb > 0`` the two jitter tensors are drawn FULL-size from the (identically seeded) device
      generator, in the reference's order (render.py:46, helpers.py:135), and sliced, so the jitter of ray i does not
      depend on the world size;
    * the per-ray outputs the batch-global losses need are all-gathered in ONE packed ``gather_batch`` (rgb and ins of both
      levels side by side, ≈0.2 MB):
      img2mse's mean and the Hungarian cost matrices / soft-IoU sums of ins_criterion (evaluator.py:19-74) are then
      evaluated identically on every rank, and autograd hands each rank the gradient rows of its own slice;
      ``args.N_ins`` (ScanNet: only the LAST N_ins rays carry labels, render.py:88-90) is applied to the gathered tensor;
    * with ``args.penalize`` (train_dmsr.py:51-58; off by default in config.py:86, on in every shipped train config) the
      emptiness penalizer is added: a ratio of batch sums, numerators and mask counts summed over ranks
      (``allreduce_sums``) before the division, inside the penalizer (``sharded=True``);
    * ``loss.backward()`` then yields on each rank the gradient contribution of its rays to the GLOBAL loss, one flat
      all-reduce (sum) of both models' gradients completes them, and every rank takes the same optimizer step.

    ``render / mse / criterion / penalizer`` default to the HIP path (networks.render.dm_nerf, evaluator.img2mse,
    evaluator.ins_criterion, penalizer.ins_penalizer); they are injectable so that the sharding logic itself is
    covered on CPU with gloo.  Returns the (global) loss and the number of bytes all-reduced."""
    from .networks import evaluator as E, penalizer as P, render as R
    rank, world = world_info()
    multi = _active(world)                               # (a forced world of one takes every multi-rank branch below)
    arena = None
    render_is_hip = render is None
    if render is None and (multi or getattr(optimizer, "wants_arena", False)):                     # HIP path: backward writes both models' gradients into one buffer
        from . import autograd
        arena = autograd.grad_arena(models)
    N = rays.shape[1]
    sizes = [ray_slice(N, r, world)[1] for r in range(world)]
    s0, cnt = ray_slice(N, rank, world)
    sl = slice(s0, s0 + cnt)
    n_imp = int(args.N_importance)
    if float(args.perturb) > 0.:
        if t_rand is None:
            t_rand = torch.rand(z_vals.shape, device=z_vals.device)
        if u is None:
            u = torch.rand([N, n_imp], device=z_vals.device)
    import copy
    largs = copy.copy(args)
    n_ins = getattr(args, "N_ins", None)
    largs.N_ins = None                                   # the label slice is taken on the gathered batch
    render = render or (lambda r, z, a, tr, uu: R.dm_nerf(r, None, None, models[0], models[1], z, a, t_rand=tr, u=uu))
    local_rays = rays[:, sl].contiguous()                # (kept: the loss tail recognises the tensors the render was called with)
    out = render(local_rays, z_vals[sl].contiguous(), largs,
                 None if t_rand is None else t_rand[sl].contiguous(),
                 None if u is None else (u if u.dim() == 1 else u[sl].contiguous()))      # a 1-D u is the grid shared by all rays
    penalize = bool(getattr(args, "penalize", False))    # train_dmsr.py:51: the emptiness term is optional (--penalize)
    # ONE exchange for everything the batch-global losses need: rgb_fine | rgb_coarse | ins_fine | ins_coarse packed side by side
    # into [n_local, 6 + 2 * ins_num] and all-gathered once (160 B per ray at ins_num 13; at the 384-ray shard of an 8-rank step
    # four latency-bound collectives would cost more than the 61 KB they carry).  Packing and un-packing move values, not
    # arithmetic: the gathered columns are the same floats, and autograd routes each rank the rows of its own slice.
    levels = ("fine", "coarse")
    widths = [out['rgb_' + l].shape[-1] for l in levels] + [out['ins_' + l].shape[-1] for l in levels]
    if multi:
        packed = gather_batch(torch.cat([out['rgb_' + l] for l in levels] + [out['ins_' + l] for l in levels], -1), sizes)
        rgb_f, rgb_c, ins_f, ins_c = torch.split(packed, widths, -1)
    else:
        rgb_f, rgb_c, ins_f, ins_c = (out['rgb_fine'], out['rgb_coarse'], out['ins_fine'], out['ins_coarse'])
    if render_is_hip and mse is None and criterion is None and penalizer is None and os.environ.get("DMNERF_FUSED_TAIL", "1") != "0":
        # the HIP path's own loss tail: the same sum as the loop below as ONE autograd node (dm_nerf_amd/losses.py) -- the
        # object-code loss of both levels per launch, the scalar arithmetic in two kernels instead of some forty
        from . import losses
        if n_ins is not None:
            ins_f, ins_c = ins_f[-n_ins:], ins_c[-n_ins:]
        loss, _ = losses.train_losses(out, local_rays[1], target, labels, ins_num, largs, rgb_ins=(rgb_f, rgb_c, ins_f, ins_c),
                                      sharded=multi)
    else:
        mse = mse or E.img2mse
        criterion = criterion or (lambda pred, gt: E.ins_criterion(pred, gt, ins_num)[0])
        penalizer = penalizer or (lambda o, lvl, rays_d: P.ins_penalizer(o['raw_' + lvl], o['z_vals_' + lvl], o['depth_' + lvl],
                                                                         rays_d, largs, sharded=True))
        loss = 0.
        for lvl, rgb, ins in (("fine", rgb_f, ins_f), ("coarse", rgb_c, ins_c)):
            if n_ins is not None:
                ins = ins[-n_ins:]
            loss = loss + mse(rgb, target) + criterion(ins, labels)
            if penalize:
                loss = loss + penalizer(out, lvl, rays[1, sl]).sum()
    optimizer.zero_grad(set_to_none=True)
    if arena is not None:
        arena.begin_step()
    if render_is_hip:
        from . import autograd
        with autograd.overlapped_backward(overlap_enabled(cnt, z_vals.shape[1], z_vals.shape[1] + n_imp, rays.device)):   # coarse and fine network backwards side by side (two streams)
            loss.backward()
    else:
        loss.backward()
    nbytes = allreduce_grads(models, arena=arena)
    optimizer.step()
    return loss.detach(), nbytes


def _default_raygen(H, W, K, c2w, row0, nrows):
    from .networks import helpers
    return helpers.get_rays_k(H, W, K, c2w, row0=row0, nrows=nrows)


def _default_render_chunk(rays_o, rays_d, z, models, args, events=None):
    from .networks import render
    out = render.dm_nerf(torch.stack([rays_o, rays_d]), None, None, models[0], models[1], z, args, _events=events)
    return out['rgb_fine'], out['ins_fine'], out['depth_fine']


_compact_index_cache = {}


def _compact_index(sizes, mx, device):
    """Row index that drops the padding of a gathered ``[world * mx, ...]`` tensor whose rank r contributed ``sizes[r]`` rows."""
    key = (tuple(sizes), mx, str(device))
    if key not in _compact_index_cache:
        _compact_index_cache[key] = torch.cat([torch.arange(r * mx, r * mx + n) for r, n in enumerate(sizes)]).to(device)
    return _compact_index_cache[key]


def _gather_band(band, sizes, world, dev):
    """The frame from the ranks' packed bands: ONE all-gather of ``band [max(sizes), cols]`` (every rank allocates the largest
    band's size, so uneven bands gather without a padding copy), then one indexed row copy that drops the padding rows."""
    if not _active(world):
        return band
    full = all_gather_cat(band)
    if len(set(sizes)) != 1:
        full = full.index_select(0, _compact_index(sizes, max(sizes), dev))
    return full


class FrameRenderer:
    """One pose, rows sharded over the ranks: the per-pose body of ``render_test`` (networks/tester.py:58-85) as a resumable
    object -- ``step(i)`` renders chunk i of this rank's band into ONE packed band buffer
    ``[band rays, 3 + ins_num + 1]`` (rgb | ins | depth; ``labels_only``: ``[band rays, 6]`` = rgb | label | conf | depth, the
    label stored as an exactly representable float), ``gather()`` assembles the frame with ONE all-gather of that buffer.
    The buffer is allocated at the largest band's size, so ranks whose band is one row shorter (H not divisible by the world
    size) gather without a padding copy and the padding rows are dropped by one indexed row copy after the collective.
    ``render_frame`` is ``step`` over all chunks + ``gather``; bench.py drives the same object chunk by chunk."""

    def __init__(self, H, W, K, c2w, models, near, far, args, chunk=4096, n_samples=64,
                 raygen=None, render_chunk=None, z_fn=None, labels_only=False, label_conf=None, ins_num=None, dtype=None):
        self.rank, self.world = world_info()
        self.dtype = dtype                                         # band dtype (default f32, what every kernel writes)
        self.H, self.W, self.models, self.args, self.chunk = int(H), int(W), models, args, int(chunk)
        self.labels_only = bool(labels_only)
        self.render_chunk = render_chunk or _default_render_chunk
        self._pass_events = render_chunk is None
        if z_fn is None:
            from .networks import helpers
            z_fn = lambda n, dev: helpers.z_val_sample(n, near, far, n_samples, device=dev)
        self.z_fn = z_fn
        if self.labels_only and label_conf is None:
            from .networks import evaluator
            label_conf = evaluator.ins_label_conf
        self.label_conf = label_conf
        row0, nrows = row_band(H, self.rank, self.world)
        ro, rd = (raygen or _default_raygen)(H, W, K, c2w, row0, nrows)
        self.rays_o, self.rays_d = ro.reshape(-1, 3), rd.reshape(-1, 3)
        self.n_local = self.rays_o.shape[0]
        self.sizes = [row_band(H, r, self.world)[1] * self.W for r in range(self.world)]
        self.dev = self.rays_o.device
        self.n_chunks = -(-self.n_local // self.chunk)
        self.z_full = self.z_fn(min(self.chunk, max(self.n_local, 1)), self.dev) if self.n_local else None
        # The packed band is sized HERE (object-code width from the model, f32 like every kernel output), not by the first rendered
        # chunk: a rank whose band has no rays (H < world size) never renders one, and must still enter the frame's all-gather with
        # a buffer of the common size -- raising on that rank alone would leave the others waiting in the collective.
        self.band = None
        if ins_num is None and models is not None:
            ins_num = getattr(models[-1], "ins_num", None)
        if ins_num is not None:
            self._alloc_band(int(ins_num), self.dtype or torch.float32)

    def _alloc_band(self, n_ins, dtype):
        self.n_ins = n_ins
        cols = 6 if self.labels_only else 3 + n_ins + 1
        self.band = torch.empty(max(max(self.sizes), 1), cols, dtype=dtype, device=self.dev)

    def step(self, i, events=None):
        """Render chunk ``i`` (0 .. n_chunks - 1; the last one may be ragged, tester.py:65-67) into the band."""
        s = i * self.chunk
        e = min(s + self.chunk, self.n_local)
        z = self.z_full if e - s == self.z_full.shape[0] else self.z_fn(e - s, self.dev)
        if self._pass_events:
            c_rgb, c_ins, c_depth = self.render_chunk(self.rays_o[s:e], self.rays_d[s:e], z, self.models, self.args, events=events)
        else:
            c_rgb, c_ins, c_depth = self.render_chunk(self.rays_o[s:e], self.rays_d[s:e], z, self.models, self.args)
        if self.band is None:
            self._alloc_band(c_ins.shape[-1], self.dtype or c_rgb.dtype)
        elif self.n_ins != c_ins.shape[-1]:
            # (a WIDTH the band was not sized for cannot be stored; every rank that renders sees the same renderer and raises the
            # same error.  A different DTYPE is converted by the assignments below instead -- ranks never disagree on the buffer.)
            if self.world > 1:
                raise RuntimeError(f"FrameRenderer: the chunk renderer returned object-code width {c_ins.shape[-1]}, the band was sized "
                                   f"for {self.n_ins} (pass ins_num= / a model with .ins_num matching the renderer)")
            self._alloc_band(c_ins.shape[-1], self.dtype or c_rgb.dtype)
        t = self.band[s:e]
        t[:, :3] = c_rgb
        if self.labels_only:
            label, conf = self.label_conf(c_ins)
            t[:, 3] = label.to(t.dtype)                            # (< 2^24: exact)
            t[:, 4] = conf
        else:
            t[:, 3:3 + self.n_ins] = c_ins
        t[:, -1] = c_depth
        return c_rgb, c_ins, c_depth

    def gather(self):
        """ONE all-gather of the packed band -> the frame's tensors on every rank."""
        H, W = self.H, self.W
        if self.band is None:                                      # no model width, no ins_num=, and no chunk rendered yet
            raise RuntimeError("FrameRenderer.gather(): the band was never sized -- render a chunk first or pass ins_num= "
                               "(required when a rank's band can be empty)")
        full = _gather_band(self.band, self.sizes, self.world, self.dev)
        rgb, depth = full[:, :3].reshape(H, W, 3), full[:, -1].reshape(H, W)
        if self.labels_only:
            return rgb, full[:, 3].to(torch.int64).reshape(H, W), full[:, 4].reshape(H, W), depth
        return rgb, full[:, 3:3 + self.n_ins].reshape(H, W, -1), depth


def render_frame(H, W, K, c2w, models, near, far, args, chunk=4096, n_samples=64,
                 raygen=None, render_chunk=None, z_fn=None, labels_only=False, label_conf=None, ins_num=None):
    """Full-frame render, rows sharded over ranks, ONE all-gather per frame (``FrameRenderer``).

    Mirrors the per-pose body of ``render_test`` (networks/tester.py:58-85): same chunking
    (``chunk`` = N_test rays, ragged last chunk), ``args.perturb`` is the caller's business
    (test scripts set it False, test_dmsr.py:86).  Returns ``rgb [H,W,3]``, ``ins [H,W,ins_num]``,
    ``depth [H,W]`` on every rank (views of the one gathered buffer).  ``labels_only=True``: the object map is reduced on the
    device to what ``ins_eval`` consumes (evaluator.py:127-137) -- ``label [H,W]`` int64 = argmax, ``conf [H,W]`` = max --
    before the gather, and ``(rgb, label, conf, depth)`` is returned: 24 instead of 16 + 4*ins_num bytes per pixel cross the links.
    """
    fr = FrameRenderer(H, W, K, c2w, models, near, far, args, chunk=chunk, n_samples=n_samples, raygen=raygen,
                       render_chunk=render_chunk, z_fn=z_fn, labels_only=labels_only, label_conf=label_conf, ins_num=ins_num)
    for i in range(fr.n_chunks):
        fr.step(i)
    return fr.gather()


def render_path(render_poses, hwk, models, args, gt_imgs=None, crop_mask=None, labels_only=False, **frame_kw):
    """The pose loop of ``render_test`` (networks/tester.py:55-90) without its file output and CPU metrics: every pose of
    ``render_poses [P,3or4,4]`` through ``render_frame`` (rows sharded over the ranks, chunks of ``args.N_test`` rays),
    the ScanNet ``crop_mask`` applied as the reference applies it (:78-82: the pixels with mask 1, reshaped to
    ``args.crop_height x args.crop_width``), and, given ``gt_imgs [P,h,w,3]``, the per-pose PSNR on the device
    (data_range 1, what ``skimage.metrics.peak_signal_noise_ratio`` computes at :89).

    Returns a dict of stacked device tensors: ``rgb [P,h,w,3]``, ``depth [P,h,w]``, and either ``ins [P,h,w,ins_num]`` or,
    with ``labels_only=True``, ``label [P,h,w]`` (int64) and ``conf [P,h,w]``; ``psnr [P]`` when ``gt_imgs`` is given."""
    H, W, K = hwk
    chunk = int(getattr(args, "N_test", 4096))
    n_samples = int(getattr(args, "N_samples", 64))
    keep = None
    if crop_mask is not None:
        keep = torch.as_tensor(crop_mask).reshape(-1) == 1
        h, w = int(args.crop_height), int(args.crop_width)
    else:
        h, w = H, W
    cols = {}
    for i, c2w in enumerate(render_poses):
        frame = render_frame(H, W, K, c2w, models, args.near, args.far, args, chunk=chunk, n_samples=n_samples,
                             labels_only=labels_only, **frame_kw)
        names = ("rgb", "label", "conf", "depth") if labels_only else ("rgb", "ins", "depth")
        for name, t in zip(names, frame):
            flat = t.reshape(H * W, *t.shape[2:])
            if keep is not None:
                flat = flat[keep.to(flat.device)]
            cols.setdefault(name, []).append(flat.reshape(h, w, *t.shape[2:]))
        if gt_imgs is not None:
            gt = torch.as_tensor(gt_imgs[i]).to(cols["rgb"][-1])
            mse = torch.mean((cols["rgb"][-1] - gt) ** 2)
            cols.setdefault("psnr", []).append(-10.0 * torch.log10(mse))
    return {k: torch.stack(v, 0) for k, v in cols.items()}


def _matmul4_f32(a, b):
    """``a @ b`` for two 4 x 4 f32 matrices with a FIXED evaluation order: products and sums rounded to f32 one at a time,
    k = 0..3 in sequence.  The reference forms ``trans @ ori_pose`` with ``torch.matmul`` on the host (manipulator.py:235), whose
    BLAS kernel -- and with it the last bit of the result -- depends on the CPU model; this one gives the same target pose on
    every host (within 1 ulp per entry of whatever the reference's host computes)."""
    import numpy as np
    a, b = a.numpy().astype(np.float32), b.numpy().astype(np.float32)
    out = np.zeros((4, 4), dtype=np.float32)
    for k in range(4):
        out = (out + (a[:, k:k + 1] * b[k:k + 1, :]).astype(np.float32)).astype(np.float32)
    return torch.from_numpy(out)


def _default_manipulate_chunk(ori_rays, tar_rays, models, args, us):
    from .networks import manipulator as Mn
    return Mn.manipulator(None, None, models[0], models[1], ori_rays, tar_rays, args, us=us)


def _default_draws(n, n_imp, count, device):
    return [torch.rand([n, n_imp], device=device) for _ in range(count)]


class ManipulationFrameRenderer:
    """One pose of the manipulation render, rows sharded over the ranks: the per-pose body of ``manipulator_eval``
    (networks/manipulator.py:232-270) -- original rays of ``ori_pose``, target rays of ``trans @ ori_pose``, the chunk loop around
    ``manipulator()`` (:137-205) and its four O(chunks^2) ``torch.cat`` accumulations -- as a resumable object (BASELINE config 5).

    * rank r owns a contiguous band of image rows and generates the original AND the ``T = len(trans_list)`` target rays of that
      band itself (raygen kernel, no scatter; the reference evaluates one transformation per call, ``trans_list`` generalises it
      the way ``manipulator()``'s ``f_tar_rays`` list does);
    * the chunks are those of the WHOLE frame -- ``[c N_test, (c + 1) N_test)``, ragged last chunk (:241-244) -- and a rank renders
      the part of each chunk that falls into its band.  ``step(c)`` makes, on EVERY rank, the ``2 + T`` draws
      ``torch.rand([chunk rays, N_importance])`` that ``manipulator()`` makes for chunk c in a single process (it resamples with
      ``det=False`` even at evaluation, :148,:170,:187), in the reference's order, and uses the rows of its own part: the device
      generator advances identically on all ranks, and **the assembled frame is bit-identical whatever the world size**
      (tests/test_gpu_manipulator_frame.py) -- which a per-rank chunking of the band could not be;
    * each part's four outputs go into ONE packed band ``[band rays, 2 (3 + C)]`` = ``final_rgb | final_ins | tar_rgb | tar_ins``
      (C = ins_num + 1: the manipulation render keeps the last object channel, :101-102), and ``gather()`` is ONE all-gather per
      frame.

    ``manipulate_chunk(ori_rays [2,n,3], tar_rays [T,2,n,3], models, args, us)`` and ``raygen`` / ``draws`` are injectable
    (CPU / gloo tests of the sharding logic); ``rank=`` / ``world=`` override the process group's view for the band arithmetic
    (a single process can then render band r of N, without collectives)."""

    def __init__(self, H, W, K, ori_pose, trans_list, models, args, chunk=None, raygen=None, manipulate_chunk=None, draws=None,
                 ins_num=None, rank=None, world=None, dtype=torch.float32):
        r_, w_ = world_info()
        self.rank, self.world = (r_ if rank is None else int(rank)), (w_ if world is None else int(world))
        self._collective = rank is None and world is None
        self.H, self.W, self.models = int(H), int(W), models
        self.chunk = int(chunk if chunk is not None else getattr(args, "N_test", 4096))
        import copy
        self.args = copy.copy(args)
        if not hasattr(self.args, "target_labels"):                 # manipulator.py:229
            self.args.target_labels = [self.args.target_label]
        self.n_imp = int(self.args.N_importance)
        self.manipulate_chunk = manipulate_chunk or _default_manipulate_chunk
        self.draws = draws or _default_draws
        raygen = raygen or _default_raygen
        row0, nrows = row_band(H, self.rank, self.world)
        pose = torch.as_tensor(ori_pose, dtype=torch.float32)
        dev_pose = pose.device
        pose_h = pose.detach().cpu()
        if pose_h.shape[0] == 3:                                    # [3,4] pose: the homogeneous row the 4 x 4 product needs
            pose_h = torch.cat([pose_h, torch.tensor([[0., 0., 0., 1.]])], 0)
        ro, rd = raygen(H, W, K, pose.to(dev_pose), row0, nrows)
        self.ori = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)])                      # [2, band, 3]
        tars = []
        for trans in trans_list:
            tar_pose = _matmul4_f32(torch.as_tensor(trans, dtype=torch.float32).cpu(), pose_h)   # manipulator.py:235
            to, td = raygen(H, W, K, tar_pose.to(dev_pose), row0, nrows)
            tars.append(torch.stack([to.reshape(-1, 3), td.reshape(-1, 3)]))
        self.T = len(tars)
        if self.T == 0:
            raise ValueError("ManipulationFrameRenderer: at least one transformation")
        self.tar = torch.stack(tars)                                                        # [T, 2, band, 3]
        self.dev = self.ori.device
        self.start = row0 * self.W                                                          # first frame ray of this band
        self.n_local = self.ori.shape[1]
        self.sizes = [row_band(H, r, self.world)[1] * self.W for r in range(self.world)]
        self.n_chunks = -(-(self.H * self.W) // self.chunk)
        if ins_num is None and models is not None:
            ins_num = getattr(models[-1], "ins_num", None)
        if ins_num is None:
            raise ValueError("ManipulationFrameRenderer: pass ins_num= (no model with .ins_num given)")
        self.C = int(ins_num) + 1
        self.band = torch.empty(max(max(self.sizes), 1), 2 * (3 + self.C), dtype=dtype, device=self.dev)

    def owned(self, c):
        """Rows of the band that chunk ``c`` of the frame covers: (first, last + 1) in band coordinates; empty if first >= last."""
        a = max(c * self.chunk, self.start) - self.start
        b = min(min((c + 1) * self.chunk, self.H * self.W), self.start + self.n_local) - self.start
        return a, b

    def step(self, c):
        """Chunk ``c`` of the frame: the ``2 + T`` draws on every rank, the render of this rank's part of it (if any)."""
        s = c * self.chunk
        n = min(self.chunk, self.H * self.W - s)
        us = self.draws(n, self.n_imp, 2 + self.T, self.dev)
        a, b = self.owned(c)
        if b <= a:
            return None
        off = self.start + a - s                                    # this part's first row inside the chunk's draws
        us = [u[off:off + (b - a)].contiguous() for u in us]
        out = self.manipulate_chunk(self.ori[:, a:b].contiguous(), self.tar[:, :, a:b].contiguous(), self.models, self.args, us)
        t, C = self.band[a:b], self.C
        t[:, 0:3], t[:, 3:3 + C], t[:, 3 + C:6 + C], t[:, 6 + C:6 + 2 * C] = out
        return out

    def gather(self):
        """ONE all-gather of the packed band -> ``final_rgb [H,W,3], final_ins [H,W,C], tar_rgb [H,W,3], tar_ins [H,W,C]``."""
        full = _gather_band(self.band, self.sizes, self.world, self.dev) if self._collective else self.band[:self.n_local]
        rows = self.H if self._collective else self.n_local // self.W
        C = self.C
        return (full[:, 0:3].reshape(rows, self.W, 3), full[:, 3:3 + C].reshape(rows, self.W, C),
                full[:, 3 + C:6 + C].reshape(rows, self.W, 3), full[:, 6 + C:6 + 2 * C].reshape(rows, self.W, C))


def manipulate_frame(H, W, K, ori_pose, trans_list, models, args, **kw):
    """One manipulated pose, rows sharded over the ranks, ONE all-gather (``ManipulationFrameRenderer``): what the chunk loop of
    ``manipulator_eval`` (networks/manipulator.py:232-270) leaves in ``full_rgb, full_ins, full_tar_rgb, full_tar_ins``, reshaped
    ``[H, W, .]`` as :273-274 does.  ``args``: N_samples, N_importance, near, far, N_test, target_labels (or target_label)."""
    fr = ManipulationFrameRenderer(H, W, K, ori_pose, trans_list, models, args, **kw)
    for c in range(fr.n_chunks):
        fr.step(c)
    return fr.gather()
