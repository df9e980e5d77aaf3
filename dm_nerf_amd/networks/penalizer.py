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


def emptiness_penalizer(raw, z_vals, depths, rays_d, tolerance, deta_w, sharded=False):
    """``emptiness_penalizer`` (networks/penalizer.py:5-55).  ``depths``: [N,1] or [N]; no gradient flows to it
    (the reference's only caller detaches it), nor to ``z_vals`` / ``rays_d``.  ``sharded`` (extension): the rays are
    one rank's slice of a batch; the loss and its gradient are those of the whole batch (distributed.py)."""
    raw = _lib.f32(raw)
    z, depth, d = _lib.f32(z_vals.detach()), _lib.f32(depths.detach().reshape(-1)), _lib.f32(rays_d.detach())
    _lib.require_gpu(raw, z, depth, d)
    return _Penalizer.apply(raw, z, depth, d, tolerance, deta_w, sharded)


def ins_penalizer(raw, z_vals, depth, rays_d, args, sharded=False):
    """``ins_penalizer`` (networks/penalizer.py:58-62)."""
    depth = depth[..., None].detach()
    return emptiness_penalizer(raw, z_vals, depth, rays_d, args.tolerance, args.deta_w, sharded)
