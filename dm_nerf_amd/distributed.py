"""One process per GPU (torchrun), RCCL over xGMI via ``torch.distributed`` (backend "nccl" on ROCm).

Rays are independent units, so the path shards with no data-path collective:

* render (reference: the serial chunk loop of ``render_test``, networks/tester.py:55-85): rank r owns
  a contiguous band of image rows, generates its own rays (no scatter), renders them in chunks,
  and ONE all-gather per frame assembles ``rgb [H,W,3]``, ``ins [H,W,ins_num]``, ``depth [H,W]``
  (replaces the O(chunks^2) ``torch.cat`` accumulation, tester.py:73-77).
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
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if use_cuda else {}
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world, **kw)
    return rank, world, device


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


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
    if world == 1:
        return t
    t = t.contiguous()
    if sizes is None or len(set(sizes)) == 1:
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        try:
            dist.all_gather_into_tensor(out, t)          # one RCCL all-gather
        except (RuntimeError, NotImplementedError):      # backends without the tensor form (gloo on device tensors)
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            out = torch.cat(parts, 0)
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], 0)


def allreduce_grads(models, average=False):
    """Sum (or average) the gradients of ``models`` across ranks in ONE flat bucket."""
    rank, world = world_info()
    params = [p for m in models for p in m.parameters() if p.grad is not None]
    if world == 1 or not params:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= world
    o = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
    return flat.numel() * flat.element_size()


def data_parallel_backward(loss_local, models, n_local, n_global):
    """Ray-sharded training step: every rank rendered ``n_local`` of the ``n_global`` rays of the SAME
    batch and formed ``loss_local`` as a SUM over its rays of per-ray terms (e.g. ``((rgb - target)**2).sum()``).
    Scales it so that the all-reduced gradients equal those of the mean over the global batch, runs backward and
    sums the gradients over ranks in one bucket.  Batch-global losses (Hungarian cost, penalizer normalisers) need
    ``all_gather_cat`` of the small per-ray outputs first -- they are not a sum of per-ray terms."""
    del n_local
    (loss_local / float(n_global)).backward()
    return allreduce_grads(models)


def _default_raygen(H, W, K, c2w, row0, nrows):
    from .networks import helpers
    return helpers.get_rays_k(H, W, K, c2w, row0=row0, nrows=nrows)


def _default_render_chunk(rays_o, rays_d, z, models, args):
    from .networks import render
    out = render.dm_nerf(torch.stack([rays_o, rays_d]), None, None, models[0], models[1], z, args)
    return out['rgb_fine'], out['ins_fine'], out['depth_fine']


def render_frame(H, W, K, c2w, models, near, far, args, chunk=4096, n_samples=64,
                 raygen=None, render_chunk=None, z_fn=None):
    """Full-frame render, rows sharded over ranks, one all-gather per output.

    Mirrors the per-pose body of ``render_test`` (networks/tester.py:58-85): same chunking
    (``chunk`` = N_test rays, ragged last chunk), ``args.perturb`` is the caller's business
    (test scripts set it False, test_dmsr.py:86).  Returns ``rgb [H,W,3]``, ``ins [H,W,ins_num]``,
    ``depth [H,W]`` on every rank.
    """
    rank, world = world_info()
    raygen = raygen or _default_raygen
    render_chunk = render_chunk or _default_render_chunk
    if z_fn is None:
        from .networks import helpers
        z_fn = lambda n, dev: helpers.z_val_sample(n, near, far, n_samples, device=dev)
    row0, nrows = row_band(H, rank, world)
    rays_o, rays_d = raygen(H, W, K, c2w, row0, nrows)
    rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    n_local = rays_o.shape[0]
    dev = rays_o.device
    rgb = ins = depth = None
    z_full = z_fn(chunk, dev)
    for s in range(0, n_local, chunk):
        e = min(s + chunk, n_local)
        z = z_full if e - s == chunk else z_fn(e - s, dev)       # ragged last chunk (tester.py:65-67)
        c_rgb, c_ins, c_depth = render_chunk(rays_o[s:e], rays_d[s:e], z, models, args)
        if rgb is None:                                          # preallocated band buffers, no repeated cat
            rgb = torch.empty(n_local, 3, dtype=c_rgb.dtype, device=dev)
            ins = torch.empty(n_local, c_ins.shape[-1], dtype=c_ins.dtype, device=dev)
            depth = torch.empty(n_local, dtype=c_depth.dtype, device=dev)
        rgb[s:e], ins[s:e], depth[s:e] = c_rgb, c_ins, c_depth
    sizes = [row_band(H, r, world)[1] * W for r in range(world)]
    rgb, ins, depth = all_gather_cat(rgb, sizes), all_gather_cat(ins, sizes), all_gather_cat(depth, sizes)
    return rgb.reshape(H, W, 3), ins.reshape(H, W, -1), depth.reshape(H, W)
