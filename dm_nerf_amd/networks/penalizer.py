"""Drop-in for the reference's ``networks/penalizer.py``: the emptiness regulariser that
consumes ``raw_*``, ``z_vals_*`` and ``depth_*`` of the dm_nerf dict, fused into two HIP kernels
(forward partial sums, backward) behind a ``torch.autograd.Function``."""
import numpy as np
import torch

from .. import _lib


def _consts(deta_w):
    """float32 constants formed exactly as the reference forms them (penalizer.py:7-10)."""
    two_w2 = float((2 * (torch.tensor([deta_w]) ** 2)).item())
    norm = float((torch.tensor([0.4]) * torch.sqrt(torch.tensor([2 * np.pi]))).item())
    return two_w2, norm


class _Penalizer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z, depth, rays_d, tolerance, deta_w, sharded=False):
        lib = _lib.load()
        N, S, ch = raw.shape
        C = ch - 4
        k2w, kh = _consts(deta_w)
        part = torch.empty(N, 4, dtype=torch.float64, device=raw.device)
        _lib.check(lib.dmnerf_penalizer_fwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(depth), _lib.ptr(rays_d), N, S, C,
                                            float(tolerance), k2w, kh, _lib.ptr(part), _lib.stream()), "dmnerf_penalizer_fwd")
        s = torch.empty(4, dtype=torch.float64, device=raw.device)    # the four batch sums; stay on the device
        _lib.check(lib.dmnerf_penalizer_sums(_lib.ptr(part), N, _lib.ptr(s), _lib.stream()), "dmnerf_penalizer_sums")
        if sharded:                                                   # ray-sharded batch: the four sums are batch-global
            from .. import distributed
            distributed.allreduce_sums(s)
        loss = torch.empty(1, dtype=torch.float32, device=raw.device)  # the reference returns a 1-element tensor
        inv = torch.empty(2, dtype=torch.float32, device=raw.device)   # 1 / (C max(sum m_b, 1e-8)), 1 / max(sum m_m, 1e-8)
        _lib.check(lib.dmnerf_penalizer_finish(_lib.ptr(s), C, _lib.ptr(loss), _lib.ptr(inv), _lib.stream()), "dmnerf_penalizer_finish")
        ctx.save_for_backward(raw, z, depth, rays_d, inv)
        ctx.consts = (float(tolerance), k2w, kh, C)
        return loss

    @staticmethod
    def backward(ctx, up):
        lib = _lib.load()
        raw, z, depth, rays_d, inv = ctx.saved_tensors
        tol, k2w, kh, C = ctx.consts
        N, S, ch = raw.shape
        scales = (inv * up.reshape(()).to(torch.float32)).contiguous()
        d_raw = torch.empty_like(raw)
        _lib.check(lib.dmnerf_penalizer_bwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(depth), _lib.ptr(rays_d), N, S, C, tol, k2w, kh,
                                            _lib.ptr(scales), _lib.ptr(d_raw), _lib.stream()), "dmnerf_penalizer_bwd")
        return d_raw, None, None, None, None, None, None


class _PenalizerFromPartials(torch.autograd.Function):
    """The scalar tail of the penalizer on per-ray partial sums that the fused compositing pass already produced
    (autograd.CompositePenFunction): sums -> [all-reduce] -> normalised loss; backward: the gradient of the partials (the same
    4-vector for every ray), which that node turns into d raw inside its own backward kernel."""

    @staticmethod
    def forward(ctx, part, C, sharded):
        lib = _lib.load()
        N = part.shape[0]
        s = torch.empty(4, dtype=torch.float64, device=part.device)
        _lib.check(lib.dmnerf_penalizer_sums(_lib.ptr(part), N, _lib.ptr(s), _lib.stream()), "dmnerf_penalizer_sums")
        if sharded:
            from .. import distributed
            distributed.allreduce_sums(s)
        loss = torch.empty(1, dtype=torch.float32, device=part.device)
        inv = torch.empty(2, dtype=torch.float32, device=part.device)
        _lib.check(lib.dmnerf_penalizer_finish(_lib.ptr(s), C, _lib.ptr(loss), _lib.ptr(inv), _lib.stream()), "dmnerf_penalizer_finish")
        ctx.save_for_backward(inv)
        ctx.N = N
        return loss

    @staticmethod
    def backward(ctx, up):
        inv, = ctx.saved_tensors
        scales = (inv * up.reshape(()).to(torch.float32)).to(torch.float64)      # the f32 products of _Penalizer.backward, exact in f64
        row = torch.zeros(4, dtype=torch.float64, device=inv.device)
        row[0::2] = scales
        return row.expand(ctx.N, 4), None, None


def emptiness_penalizer(raw, z_vals, depths, rays_d, tolerance, deta_w, sharded=False):
    """``emptiness_penalizer`` (networks/penalizer.py:5-55).  ``depths``: [N,1] or [N]; no gradient flows to it
    (the reference's only caller detaches it), nor to ``z_vals`` / ``rays_d``.  ``sharded`` (extension): the rays are
    one rank's slice of a batch; the loss and its gradient are those of the whole batch (distributed.py)."""
    raw = _lib.f32(raw)
    z, depth, d = _lib.f32(z_vals.detach()), _lib.f32(depths.detach().reshape(-1)), _lib.f32(rays_d.detach())
    _lib.require_gpu(raw, z, depth, d)
    return _Penalizer.apply(raw, z, depth, d, tolerance, deta_w, sharded)


def ins_penalizer(raw, z_vals, depth, rays_d, args, sharded=False):
    """``ins_penalizer`` (networks/penalizer.py:58-62).  When ``raw`` / ``z_vals`` / ``depth`` are a level of a training-mode
    ``dm_nerf`` dict rendered with ``args.penalize`` set, the per-ray sums were already formed by the compositing pass
    (autograd.CompositePenFunction): only the scalar tail runs here, and the gradient is produced inside that pass's backward
    kernel -- same values, bit for bit."""
    if (torch.is_tensor(depth) and depth.is_cuda and torch.is_tensor(raw) and raw.dtype == torch.float32
            and getattr(args, "deta_w", None) is not None and getattr(args, "tolerance", None) is not None):
        from .. import autograd
        k2w, kh = _consts(args.deta_w)
        part = autograd.pen_partials(depth, raw, _lib.f32(z_vals.detach()), _lib.f32(rays_d.detach()), (float(args.tolerance), k2w, kh))
        if part is not None:
            return _PenalizerFromPartials.apply(part, raw.shape[-1] - 4, sharded)
    depth = depth[..., None].detach()
    return emptiness_penalizer(raw, z_vals, depth, rays_d, args.tolerance, args.deta_w, sharded)
