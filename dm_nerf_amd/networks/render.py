"""Drop-in for the reference's ``networks/render.py``: render_train, dm_nerf."""
import ctypes

import torch

from .. import _lib, weights
from . import helpers


def render_train(raw, z_vals, rays_d):
    """``render_train`` (networks/render.py:6-28) -> (rgb_map, weights, depth_map, ins_map)."""
    if torch.is_grad_enabled() and raw.requires_grad:
        from .. import autograd
        return autograd.render_train_train(raw, z_vals, rays_d)
    raw, z, d = _lib.f32(raw), _lib.f32(z_vals), _lib.f32(rays_d)
    _lib.require_gpu(raw, z, d)
    N, S, ch = raw.shape
    C = ch - 4
    dev = raw.device
    rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
    w = torch.empty(N, S, dtype=torch.float32, device=dev)
    depth = torch.empty(N, dtype=torch.float32, device=dev)
    ins = torch.empty(N, C - 1, dtype=torch.float32, device=dev)
    _lib.check(_lib.load().dmnerf_composite_fwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(d), N, S, C, _lib.ptr(rgb), _lib.ptr(w),
                                                _lib.ptr(depth), _lib.ptr(ins), _lib.stream()), "dmnerf_composite_fwd")
    return rgb, w, depth, ins


def run_network(model, rays_o, rays_d, z_vals, split=None):
    """pts = o + d z -> embed(pts) | embed(d/|d|) -> ``model`` (networks/render.py:49-61 / :71-83)
    as one fused kernel: ``[N,3], [N,3], [N,S] -> raw [N,S,4+C]`` (inference only).  ``split``: None (the f32 kernels) or
    ``weights.split_mode(args)`` = "bf16x3" / "f16x2", the opt-in split-operand kernels (f32-class, not bitwise)."""
    if not model._fused_ok():                              # another network shape: layer by layer (dm_nerf_amd/generic.py)
        if split:
            raise ValueError("args.mfma_split: the split-operand kernels exist for the 8 x 256 network only")
        from .. import generic
        return generic.run_network(model, rays_o, rays_d, z_vals, train=False)
    rays_o, rays_d, z = _lib.f32(rays_o.reshape(-1, 3)), _lib.f32(rays_d.reshape(-1, 3)), _lib.f32(z_vals)
    _lib.require_gpu(rays_o, rays_d, z)
    N, S = z.shape
    raw = torch.empty(N, S, 4 + model.ins_num + 1, dtype=torch.float32, device=z.device)
    lib = _lib.load()
    fn, blob, name = {None: (lib.dmnerf_mlp_fwd_rays, model.blob, "dmnerf_mlp_fwd_rays"),
                      "bf16x3": (lib.dmnerf_mlp_fwd_rays_split, model.blob_split, "dmnerf_mlp_fwd_rays_split"),
                      "f16x2": (lib.dmnerf_mlp_fwd_rays_f16, model.blob_f16, "dmnerf_mlp_fwd_rays_f16")}[split or None]
    _lib.check(fn(_lib.ptr(blob()), model.ins_num, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z), N, S, _lib.ptr(raw), _lib.stream()), name)
    return raw


def check_draws(t_rand, u, N, S, n_imp, perturb, dev):
    """The two random tensors of one ``dm_nerf`` call, validated before raw pointers reach the kernels.
    ``perturb > 0``: ``t_rand [N,S]`` (render.py:46) then ``u [N,n_imp]`` (helpers.py:135), drawn here in the reference's
    order unless passed in.  Otherwise no jitter and ``u`` = the deterministic grid ``linspace(0,1,n_imp)`` shared by all
    rays (``det = (perturb == 0.)``, render.py:67) -- or a caller-supplied ``[n_imp]`` / ``[N,n_imp]`` tensor.
    Returns ``(t_rand or None, u, u_row_stride)``; a wrong shape raises instead of reading out of bounds."""
    if perturb > 0.:
        if t_rand is None:
            t_rand = torch.rand([N, S], device=dev)
        if u is None:
            u = torch.rand([N, n_imp], device=dev)
    else:
        t_rand = None
        if u is None:
            u = helpers.linspace01(n_imp, dev)
    if t_rand is not None:
        if tuple(t_rand.shape) != (N, S):
            raise ValueError(f"dm_nerf: t_rand must be [{N}, {S}] (one draw per coarse sample), got {tuple(t_rand.shape)}")
        t_rand = _lib.f32(t_rand)
        _lib.require_gpu(t_rand)
    if tuple(u.shape) not in ((n_imp,), (N, n_imp)):
        raise ValueError(f"dm_nerf: u must be [{n_imp}] or [{N}, {n_imp}], got {tuple(u.shape)}")
    u = _lib.f32(u)
    _lib.require_gpu(u)
    return t_rand, u, (0 if u.dim() == 1 else n_imp)


def dm_nerf(rays, position_embedder, view_embedder, model_coarse, model_fine, z_vals_coarse, args,
            t_rand=None, u=None, _events=None):
    """``dm_nerf`` (networks/render.py:31-96) -> the reference's 10-key dict.

    ``position_embedder`` / ``view_embedder`` are accepted for signature compatibility; the
    encoding (multires 10 / 4, the only values create_nerf's configs use) is computed inside the
    fused kernel.  RNG parity: with ``args.perturb > 0`` the reference draws ``torch.rand([N,S])``
    (render.py:46) and then ``torch.rand([N,N_importance])`` (helpers.py:135); the same two draws
    are made here, in that order, on the rays' device -- or pass ``t_rand`` / ``u`` (extension).
    ``_events``: optional (begin, end) ``torch.cuda.Event`` pair recorded around the fine MLP kernel.
    """
    # training = gradients are wanted for EITHER model (a frozen fine model with a trainable coarse one still trains)
    training = torch.is_grad_enabled() and any(p.requires_grad for m in (model_coarse, model_fine) for p in m.parameters())
    if training:
        from .. import autograd
        return autograd.dm_nerf_train(rays, model_coarse, model_fine, z_vals_coarse, args, t_rand=t_rand, u=u)
    for emb, want in ((position_embedder, model_fine.input_ch_pts), (view_embedder, model_fine.input_ch_views)):
        if getattr(emb, "out_dim", want) != want:
            raise ValueError("dm_nerf: the embedders' out_dim does not match the models' input channels")
    rays_o, rays_d = rays
    rays_o, rays_d = _lib.f32(rays_o.reshape(-1, 3)), _lib.f32(rays_d.reshape(-1, 3))
    z_in = _lib.f32(z_vals_coarse)
    _lib.require_gpu(rays_o, rays_d, z_in)
    dev = rays_o.device
    N, S = z_in.shape
    n_imp = int(args.N_importance)
    if n_imp < 0:
        raise ValueError("dm_nerf: N_importance must be >= 0")
    ins_num = model_fine.ins_num
    C = ins_num + 1
    perturb = float(args.perturb)
    t_rand, u, u_stride = check_draws(t_rand, u, N, S, n_imp, perturb, dev)
    if n_imp == 0 or not (model_coarse._fused_ok() and model_fine._fused_ok()):
        # composed from the stage kernels instead of the one fused call.  N_importance = 0 (config.py:43 allows it; no
        # shipped config uses it): sample_pdf returns [N, 0], the merged depths are the coarse ones (render.py:66-70) and the
        # fine network is evaluated on them.  A network shape other than the shipped one: run_network goes layer by layer.
        z_c = helpers.stratify(z_in, t_rand) if t_rand is not None else z_in
        raw_c = run_network(model_coarse, rays_o, rays_d, z_c)
        rgb_c, w_c, dep_c, ins_c = render_train(raw_c, z_c, rays_d)
        z_f = z_c.clone() if n_imp == 0 else helpers.importance_resample(z_c, w_c, n_imp, u=u)
        raw_f = run_network(model_fine, rays_o, rays_d, z_f)
        rgb_f, _, dep_f, ins_f = render_train(raw_f, z_f, rays_d)
        if getattr(args, "is_train", False) and getattr(args, "N_ins", None) is not None:
            ins_f, ins_c = ins_f[-args.N_ins:], ins_c[-args.N_ins:]
        return {'rgb_fine': rgb_f, 'ins_fine': ins_f, 'z_vals_fine': z_f, 'raw_fine': raw_f, 'raw_coarse': raw_c, 'rgb_coarse': rgb_c,
                'ins_coarse': ins_c, 'z_vals_coarse': z_c, 'depth_fine': dep_f, 'depth_coarse': dep_c}
    SF = S + n_imp
    f = dict(dtype=torch.float32, device=dev)
    out = {
        'rgb_fine': torch.empty(N, 3, **f), 'ins_fine': torch.empty(N, C - 1, **f),
        'z_vals_fine': torch.empty(N, SF, **f), 'raw_fine': torch.empty(N, SF, 4 + C, **f),
        'raw_coarse': torch.empty(N, S, 4 + C, **f), 'rgb_coarse': torch.empty(N, 3, **f),
        'ins_coarse': torch.empty(N, C - 1, **f),
        # without jitter the reference hands its input grid back (render.py:40-47 is skipped): alias, no copy
        'z_vals_coarse': torch.empty(N, S, **f) if t_rand is not None else z_in,
        'depth_fine': torch.empty(N, **f), 'depth_coarse': torch.empty(N, **f),
    }
    ws = torch.empty(N, SF, **f)
    a = _lib.RenderArgs()
    # args.fuse_heads (extension, default off): inference with the activation-free feature linears folded into the
    # hidden layers (-19 % MACs; results equal up to f32 re-association)
    # args.mfma_split (extension, default off): split-operand 16-bit MFMA inference (f32-class accuracy, not bitwise the f32
    # chain): True / "bf16x3" = three bf16 planes, six products; "f16x2" = two f16 planes, three products
    fused = bool(getattr(args, "fuse_heads", False))
    split = weights.split_mode(args)
    a.fused_heads = {"bf16x3": 2, "f16x2": 3}[split] if split else (1 if fused else 0)
    pick = {"bf16x3": (lambda mdl: mdl.blob_split()), "f16x2": (lambda mdl: mdl.blob_f16())}[split] if split else \
        ((lambda mdl: mdl.blob_fused()) if fused else (lambda mdl: mdl.blob()))
    a.d_blob_coarse = pick(model_coarse).data_ptr()
    a.d_blob_fine = pick(model_fine).data_ptr()
    a.ins_num = ins_num
    a.d_rays_o, a.d_rays_d, a.d_z_in = rays_o.data_ptr(), rays_d.data_ptr(), z_in.data_ptr()
    a.d_t_rand = t_rand.data_ptr() if t_rand is not None else None
    a.d_u, a.u_row_stride = u.data_ptr(), u_stride
    a.N, a.S, a.n_imp = N, S, n_imp
    a.d_z_coarse, a.d_raw_coarse = out['z_vals_coarse'].data_ptr(), out['raw_coarse'].data_ptr()
    a.d_rgb_coarse, a.d_depth_coarse = out['rgb_coarse'].data_ptr(), out['depth_coarse'].data_ptr()
    a.d_ins_coarse = out['ins_coarse'].data_ptr()
    a.d_z_fine, a.d_raw_fine = out['z_vals_fine'].data_ptr(), out['raw_fine'].data_ptr()
    a.d_rgb_fine, a.d_depth_fine = out['rgb_fine'].data_ptr(), out['depth_fine'].data_ptr()
    a.d_ins_fine = out['ins_fine'].data_ptr()
    a.d_weights_ws = ws.data_ptr()
    if _events is not None:          # (begin, end) torch.cuda.Event pair around the fine-network MLP kernel
        for e in _events:
            e.record()               # torch creates the hipEvent_t lazily; the library re-records it in place
        a.ev_fine_mlp_begin, a.ev_fine_mlp_end = _events[0].cuda_event, _events[1].cuda_event
    _lib.check(_lib.load().dmnerf_render_rays_fwd(ctypes.byref(a), _lib.stream()), "dmnerf_render_rays_fwd")
    if split == "f16x2":
        from .. import autograd
        if autograd.f16_check_enabled(args):             # opt-in diagnostic (DMNERF_CHECK_F16=1 / args.check_f16): did a conversion saturate?
            autograd.f16x2_probe(model_coarse, rays_o, rays_d, out['z_vals_coarse'])
            autograd.f16x2_probe(model_fine, rays_o, rays_d, out['z_vals_fine'])
            autograd.check_f16x2(dev)
    if getattr(args, "is_train", False) and getattr(args, "N_ins", None) is not None:
        out['ins_fine'] = out['ins_fine'][-args.N_ins:]          # render.py:88-90
        out['ins_coarse'] = out['ins_coarse'][-args.N_ins:]
    return out
