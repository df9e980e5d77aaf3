"""Training-mode entry points: ``torch.autograd.Function`` wrappers around the HIP forward /
backward kernels, so ``total_loss.backward()`` + ``torch.optim.Adam`` of the reference training
loop (train_dmsr.py:56-64) work unchanged.

Gradient barriers of the reference are reproduced structurally:
  * ``weights_ins.detach()`` (render.py:22-23)  -- composite_bwd gives the ins logits no path into sigma;
  * ``h.detach()`` on the ins branch (dm_nerf.py:95) -- mlp_bwd_data does not add d(ins_feature) to dh_7;
  * ``z_samples.detach()`` (render.py:68) -- the resampled depths are computed outside autograd.

Weight gradients dW = dy . x^T (GEMMs over the M samples of the batch, both operands stored
feature-major by the kernels) and the bias row sums run in the split-K f32-MFMA kernel of
csrc/wgrad.hip (dmnerf_mlp_bwd_weights); no vendor BLAS is involved.
"""
import torch

from . import _lib
from .networks import helpers

W = 256
HW = 128


def _row_len(M):
    """Rows of the training workspace are padded to a multiple of 32 samples (csrc/layout.h::save_row_len)."""
    return (M + 31) & ~31


def _views(buf, M):
    """Feature-major [rows, row_len] COPIES of the tensors of a SaveLayout workspace (csrc/layout.h) --
    diagnostics / tests only.  On the device every tensor is block-major [block][rows][32 samples];
    padding columns hold duplicates of the last sample (activations) or exact zeros (gradients).  The
    accumulator-layout tensors (everything but the encodings) keep the memory row order 0,4,1,5,2,6,3,7
    inside each group of 8 features (csrc/layout.h::row_feature); the copies are in feature order."""
    Mp = _row_len(M)
    o = 0
    out = {}
    for name, rows, n in (("pe", 63, 1), ("de", 27, 1), ("h", W, 8), ("f", W, 1), ("q", W, 1), ("g1", HW, 1), ("g2", HW, 1)):
        ts = []
        for _ in range(n):
            t = buf[o:o + rows * Mp].view(Mp // 32, rows, 32).permute(1, 0, 2).reshape(rows, Mp)
            if name not in ("pe", "de"):
                rho = torch.arange(rows, device=buf.device)
                feat = (rho & ~7) | ((rho & 7) >> 1) | ((rho & 1) << 2)
                t = t[torch.argsort(feat)]
            ts.append(t)
            o += rows * Mp
        out[name] = ts[0] if n == 1 else torch.stack(ts)
    return out


_plan_cache = {}


def wgrad_plan(ins_num, M, device, max_wgs=None):
    """Device-resident split-K plan of the weight-gradient kernel for (ins_num, M): (jobs, n_jobs, outs, n_outs, part_floats)."""
    import ctypes

    import numpy as np
    if max_wgs is None:
        max_wgs = torch.cuda.get_device_properties(device).multi_processor_count     # one workgroup per CU
    key = (ins_num, M, str(device), max_wgs)
    if key not in _plan_cache:
        lib = _lib.load()
        jb, ob, pf = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        nj, no = ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.dmnerf_wgrad_plan_sizes(ins_num, M, max_wgs, ctypes.byref(jb), ctypes.byref(ob), ctypes.byref(pf),
                                               ctypes.byref(nj), ctypes.byref(no)), "dmnerf_wgrad_plan_sizes")
        hj, ho = np.empty(jb.value, dtype=np.uint8), np.empty(ob.value, dtype=np.uint8)
        _lib.check(lib.dmnerf_wgrad_plan(ins_num, M, max_wgs, hj.ctypes.data_as(ctypes.c_void_p), jb.value,
                                         ho.ctypes.data_as(ctypes.c_void_p), ob.value), "dmnerf_wgrad_plan")
        _plan_cache[key] = (torch.from_numpy(hj).to(device), nj.value, torch.from_numpy(ho).to(device), no.value, pf.value)
    return _plan_cache[key]


def split_flat_grads(model, flat):
    """Views of the flat gradient vector (reference parameter order) as the model's parameter tensors."""
    out, o = [], 0
    for _, p in model.named_parameters():
        n = p.numel()
        out.append(flat[o:o + n].view_as(p))
        o += n
    return out


class MLPRaysFunction(torch.autograd.Function):
    """raw[N,S,4+C] = DM_NeRF(embed(o + d z) | embed(d/|d|)); parameters are inputs 3.. in state_dict order."""

    @staticmethod
    def forward(ctx, model, rays_o, rays_d, z, *params):
        lib = _lib.load()
        N, S = z.shape
        M = N * S
        ins_num = model.ins_num
        raw = torch.empty(N, S, 4 + ins_num + 1, dtype=torch.float32, device=z.device)
        save = torch.empty(lib.dmnerf_train_save_floats(M), dtype=torch.float32, device=z.device)
        _lib.check(lib.dmnerf_mlp_fwd_rays_train(_lib.ptr(model.blob()), ins_num, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z),
                                                 N, S, _lib.ptr(raw), _lib.ptr(save), _lib.stream()), "dmnerf_mlp_fwd_rays_train")
        ctx.model, ctx.M, ctx.save = model, M, save
        ctx.blob, ctx.blob_t = model.blob(), model.blob_t()      # the weights this forward used
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        lib = _lib.load()
        model, M = ctx.model, ctx.M
        ins_num = model.ins_num
        C = ins_num + 1
        g = _lib.f32(g_raw).reshape(M, 4 + C)
        dsave = torch.empty_like(ctx.save)
        Mp = _row_len(M)
        gt = torch.empty(Mp // 32, 4 + C, 32, dtype=torch.float32, device=g.device)   # d raw, block-major, written by the kernel
        _lib.check(lib.dmnerf_mlp_bwd_data(_lib.ptr(ctx.blob), _lib.ptr(ctx.blob_t), ins_num, _lib.ptr(ctx.save), _lib.ptr(g), M,
                                           _lib.ptr(dsave), _lib.ptr(gt), _lib.stream()), "dmnerf_mlp_bwd_data")
        jobs, n_jobs, outs, n_outs, part_floats = wgrad_plan(ins_num, M, g.device)
        part = torch.empty(part_floats, dtype=torch.float32, device=g.device)
        flat = torch.empty(lib.dmnerf_param_count(ins_num), dtype=torch.float32, device=g.device)
        _lib.check(lib.dmnerf_mlp_bwd_weights(_lib.ptr(ctx.save), _lib.ptr(dsave), _lib.ptr(gt), M, _lib.ptr(jobs), n_jobs,
                                              _lib.ptr(outs), n_outs, _lib.ptr(part), _lib.ptr(flat), _lib.stream()),
                   "dmnerf_mlp_bwd_weights")
        ctx.save = None
        return (None, None, None, None) + tuple(split_flat_grads(model, flat))


class CompositeFunction(torch.autograd.Function):
    """render_train (networks/render.py:6-28) with its analytic backward."""

    @staticmethod
    def forward(ctx, raw, z, rays_d):
        lib = _lib.load()
        N, S, ch = raw.shape
        C = ch - 4
        f = dict(dtype=torch.float32, device=raw.device)
        rgb, w = torch.empty(N, 3, **f), torch.empty(N, S, **f)
        depth, ins = torch.empty(N, **f), torch.empty(N, C - 1, **f)
        _lib.check(lib.dmnerf_composite_fwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays_d), N, S, C, _lib.ptr(rgb), _lib.ptr(w),
                                            _lib.ptr(depth), _lib.ptr(ins), _lib.stream()), "dmnerf_composite_fwd")
        ctx.save_for_backward(raw, z, rays_d, ins)
        return rgb, w, depth, ins

    @staticmethod
    def backward(ctx, g_rgb, g_w, g_depth, g_ins):
        lib = _lib.load()
        raw, z, rays_d, ins = ctx.saved_tensors
        N, S, ch = raw.shape
        C = ch - 4

        def prep(g, shape):
            if g is None:
                return None
            return _lib.f32(g.expand(shape) if g.shape != torch.Size(shape) else g)
        g_rgb = prep(g_rgb, (N, 3)) if g_rgb is not None else torch.zeros(N, 3, device=raw.device)
        g_ins = prep(g_ins, (N, C - 1)) if g_ins is not None else torch.zeros(N, C - 1, device=raw.device)
        g_w, g_depth = prep(g_w, (N, S)), prep(g_depth, (N,))
        d_raw = torch.empty_like(raw)
        _lib.check(lib.dmnerf_composite_bwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays_d), _lib.ptr(ins), _lib.ptr(g_rgb), _lib.ptr(g_ins),
                                            _lib.ptr(g_depth), _lib.ptr(g_w), N, S, C, _lib.ptr(d_raw), _lib.stream()), "dmnerf_composite_bwd")
        return d_raw, None, None


MAX_TRAIN_SAMPLES = 1048576          # DMNERF_MAX_TRAIN_SAMPLES (include/dmnerf_hip.h)


def _params(model):
    return [p for _, p in model.named_parameters()]


def run_network_train(model, rays_o, rays_d, z):
    """Differentiable (w.r.t. the parameters) fused points + encoding + MLP."""
    rays_o, rays_d, z = _lib.f32(rays_o.reshape(-1, 3)), _lib.f32(rays_d.reshape(-1, 3)), _lib.f32(z)
    _lib.require_gpu(rays_o, rays_d, z)
    model._check_supported()
    N, S = z.shape
    max_rays = max(1, MAX_TRAIN_SAMPLES // S)
    if N <= max_rays:
        return MLPRaysFunction.apply(model, rays_o, rays_d, z, *_params(model))
    # a training launch addresses its saved-activation workspace with 32-bit byte offsets (DMNERF_MAX_TRAIN_SAMPLES in
    # include/dmnerf_hip.h): larger batches run as several launches, each with its own workspace; autograd adds the
    # parameter gradients of the pieces (rays are independent, so this is the same sum in a different order)
    params = _params(model)
    return torch.cat([MLPRaysFunction.apply(model, rays_o[s:s + max_rays], rays_d[s:s + max_rays], z[s:s + max_rays], *params)
                      for s in range(0, N, max_rays)], 0)


def render_train_train(raw, z_vals, rays_d):
    raw, z, d = _lib.f32(raw), _lib.f32(z_vals.detach()), _lib.f32(rays_d.detach())
    _lib.require_gpu(raw, z, d)
    return CompositeFunction.apply(raw, z, d)


def mlp_forward_train(model, x):
    raise NotImplementedError(
        "dm_nerf_amd: DM_NeRF.forward on pre-embedded rows is inference-only; training goes through dm_nerf() / "
        "run_network_train (the reference's train loops only ever call the model through dm_nerf, train_dmsr.py:32)")


def dm_nerf_train(rays, model_coarse, model_fine, z_vals_coarse, args, t_rand=None, u=None):
    """Training-mode ``dm_nerf`` (networks/render.py:31-96): same dict, differentiable w.r.t. both models."""
    rays_o, rays_d = rays
    rays_o, rays_d = _lib.f32(rays_o.reshape(-1, 3)).detach(), _lib.f32(rays_d.reshape(-1, 3)).detach()
    z_in = _lib.f32(z_vals_coarse).detach()
    _lib.require_gpu(rays_o, rays_d, z_in)
    N, S = z_in.shape
    n_imp = int(args.N_importance)
    perturb = float(args.perturb)
    if perturb > 0.:                                   # RNG order of the reference: [N,S] then [N,n_imp]
        if t_rand is None:
            t_rand = torch.rand(z_in.shape, device=z_in.device)
        if u is None:
            u = torch.rand([N, n_imp], device=z_in.device)
        z_coarse = helpers.stratify(z_in, t_rand)
    else:
        z_coarse = z_in
    raw_coarse = run_network_train(model_coarse, rays_o, rays_d, z_coarse)
    rgb_coarse, weights_coarse, depth_coarse, ins_coarse = CompositeFunction.apply(raw_coarse, z_coarse, rays_d)
    with torch.no_grad():                              # z_samples.detach()  (render.py:68)
        z_fine = helpers.importance_resample(z_coarse, weights_coarse.detach(), n_imp, det=(perturb == 0.), u=u)
    raw_fine = run_network_train(model_fine, rays_o, rays_d, z_fine)
    rgb_fine, weights_fine, depth_fine, ins_fine = CompositeFunction.apply(raw_fine, z_fine, rays_d)
    if getattr(args, "is_train", False) and getattr(args, "N_ins", None) is not None:
        ins_fine = ins_fine[-args.N_ins:]
        ins_coarse = ins_coarse[-args.N_ins:]
    return {'rgb_fine': rgb_fine, 'ins_fine': ins_fine, 'z_vals_fine': z_fine, 'raw_fine': raw_fine,
            'raw_coarse': raw_coarse, 'rgb_coarse': rgb_coarse, 'ins_coarse': ins_coarse,
            'z_vals_coarse': z_coarse, 'depth_fine': depth_fine, 'depth_coarse': depth_coarse}
