"""DM_NeRF for network shapes other than the one the fused kernels are specialised for (D = 8, W = 256, skips = [4],
multires 10 / 4): the layer-by-layer path on the kernels of ``csrc/generic.hip``.

``create_nerf`` (config.py:126-138) passes ``args.netdepth / netwidth / multires / multires_views`` through; no shipped config
changes them, but a configuration that does must run, not raise.  This module chains one strided f32-MFMA GEMM per linear
layer exactly as ``DM_NeRF.forward`` does (networks/dm_nerf.py:80-106) -- skip concat ``[h, pts]`` (:87), activation-free
feature linears (:89,96), ``h.detach()`` on the ins branch (:95), output ``cat[rgb, density, ins]`` (:105) -- and, for
training, as its autograd would (data gradient with the ReLU mask in the GEMM epilogue, split-K weight gradients, column-sum
bias gradients).  Slower than the fused path (one pass over HBM per layer) and used ONLY when ``DM_NeRF._fused_ok()`` is
false; the shipped shape never comes here.  No torch compute ops: tensors are allocated with torch, every FLOP runs in
``libdmnerf_hip.so``.
"""
import torch

from . import _lib


def _gemm(A, sai, sak, B, sbk, sbj, C, ldc, I, J, K, bias=None, relu=False, mask=None, ldm=0, accumulate=False, splits=1):
    lib = _lib.load()
    ws = None
    if splits > 1:
        ws = torch.empty(splits * I * J, dtype=torch.float32, device=C.device)
    _lib.check(lib.dmnerf_gemm(_lib.ptr(A), sai, sak, _lib.ptr(B), sbk, sbj, _lib.ptr(C), ldc, I, J, K, _lib.ptr(bias), int(relu),
                               _lib.ptr(mask), ldm, int(accumulate), _lib.ptr(ws), splits, _lib.stream()), "dmnerf_gemm")


def _splits(I, J, K):
    """Split-K factor of a weight-gradient GEMM (reduction over the K = M samples): enough slices to fill the chip."""
    tiles = ((I + 127) // 128) * ((J + 127) // 128)
    return int(max(1, min((K + 2047) // 2048, max(1, 1024 // tiles), 4096)))


def _linear(x, ldx, Wt, b, out, ldo, M, relu=False):
    """out[:, :N] = act(x[:, :K] W^T + b);  W [N, K] row-major."""
    N, K = Wt.shape
    _gemm(x, ldx, 1, Wt, 1, K, out, ldo, M, N, K, bias=b, relu=relu)


def _dgrad(dy, ldy, Wt, n_in, out, ldo, M, mask=None, ldm=0, accumulate=False):
    """out[:, :n_in] (+)= dy W[:, :n_in], then . [mask > 0];  W [N, K] row-major, n_in <= K."""
    N, K = Wt.shape
    _gemm(dy, ldy, 1, Wt, K, 1, out, ldo, M, n_in, N, mask=mask, ldm=ldm, accumulate=accumulate)


def _wgrad(dy, ldy, n_out, x, ldx, n_in, M):
    """-> dW [n_out, n_in] = dy[:, :n_out]^T x[:, :n_in] over the M samples, db [n_out] = column sums of dy."""
    lib = _lib.load()
    dW = torch.empty(n_out, n_in, dtype=torch.float32, device=x.device)
    _gemm(dy, 1, ldy, x, ldx, 1, dW, n_in, n_out, n_in, M, splits=_splits(n_out, n_in, M))
    db = torch.empty(n_out, dtype=torch.float32, device=x.device)
    slices = int(max(1, min(256, (M + 4095) // 4096)))
    ws = torch.empty(slices * n_out, dtype=torch.float32, device=x.device)
    _lib.check(lib.dmnerf_colsum(_lib.ptr(dy), ldy, M, n_out, _lib.ptr(db), _lib.ptr(ws), slices, _lib.stream()), "dmnerf_colsum")
    return dW, db


def _copy_cols(src, lds, dst, ldd, M, n):
    _lib.check(_lib.load().dmnerf_copy_cols(_lib.ptr(src), lds, _lib.ptr(dst), ldd, M, n, _lib.stream()), "dmnerf_copy_cols")


def _col(t, c):
    """Pointer to column ``c`` of a contiguous 2-D tensor (a view: plumbing, no copy)."""
    return t[:, c:]


class _Net:
    """The parameters of one DM_NeRF in forward order, with the dimensions the chain needs."""

    def __init__(self, model, params):
        names = [n for n, _ in model.named_parameters()]
        p = dict(zip(names, params))
        self.D, self.W, self.inp, self.inv = model.D, model.W, model.input_ch_pts, model.input_ch_views
        self.skips = set(model.skips)
        self.HW = self.W // 2
        self.C = model.ins_num + 1
        self.trunk = [(p[f"mlps.{i}.weight"], p[f"mlps.{i}.bias"]) for i in range(self.D)]
        g = lambda n: (p[n + ".weight"], p[n + ".bias"])
        self.rf, self.inf_, self.rh, self.ih = g("rgb_feature_linear"), g("ins_feature_linear"), g("rgb_feature_linears.0"), g("ins_feature_linears.0")
        self.den, self.io, self.ro = g("density_linear"), g("ins_linear"), g("rgb_linear")
        self.names = names
        if (self.D - 1) in self.skips:
            raise ValueError("DM_NeRF: a skip after the last trunk layer feeds W + input_ch_pts columns into W-column heads (the reference fails too)")


def forward_layers(net, x_pos, x_dir, save):
    """raw [M, 4 + C] for embedded inputs x_pos [M, inp], x_dir [M, inv]; ``save``: dict filled with what backward needs."""
    M = x_pos.shape[0]
    dev = x_pos.device
    f = dict(dtype=torch.float32, device=dev)
    W, inp, inv, HW, C = net.W, net.inp, net.inv, net.HW, net.C
    hs = [(x_pos, inp)]                                     # (buffer, leading dim) of every trunk layer's INPUT, then the trunk output
    for i, (Wi, bi) in enumerate(net.trunk):
        cols = W + (inp if i in net.skips else 0)
        buf = torch.empty(M, cols, **f)
        h, ldh = hs[-1]
        _linear(h, ldh, Wi, bi, buf, cols, M, relu=True)
        if i in net.skips:                                   # cat[h, pts]  (dm_nerf.py:87)
            _copy_cols(x_pos, inp, _col(buf, W), cols, M, inp)
        hs.append((buf, cols))
    h, ldh = hs[-1]
    xr = torch.empty(M, W + inv, **f)                        # cat[rgb_feature, dirs]  (:90)
    _linear(h, ldh, net.rf[0], net.rf[1], xr, W + inv, M)
    _copy_cols(x_dir, inv, _col(xr, W), W + inv, M, inv)
    g1 = torch.empty(M, HW, **f)
    _linear(xr, W + inv, net.rh[0], net.rh[1], g1, HW, M, relu=True)
    q = torch.empty(M, W, **f)                               # ins_feature(h.detach())  (:95-96)
    _linear(h, ldh, net.inf_[0], net.inf_[1], q, W, M)
    g2 = torch.empty(M, HW, **f)
    _linear(q, W, net.ih[0], net.ih[1], g2, HW, M, relu=True)
    out = torch.empty(M, 4 + C, **f)                         # cat[rgb, density, ins]  (:105)
    _linear(g1, HW, net.ro[0], net.ro[1], out, 4 + C, M)
    _linear(h, ldh, net.den[0], net.den[1], _col(out, 3), 4 + C, M)
    _linear(g2, HW, net.io[0], net.io[1], _col(out, 4), 4 + C, M)
    if save is not None:
        save.update(hs=hs, xr=xr, g1=g1, q=q, g2=g2)
    return out


def backward_layers(net, save, g_out):
    """Parameter gradients (list in ``named_parameters`` order) for the upstream gradient g_out [M, 4 + C]."""
    g_out = _lib.f32(g_out)
    M = g_out.shape[0]
    dev = g_out.device
    f = dict(dtype=torch.float32, device=dev)
    W, inp, inv, HW, C = net.W, net.inp, net.inv, net.HW, net.C
    ld = 4 + C
    hs, xr, g1, q, g2 = save["hs"], save["xr"], save["g1"], save["q"], save["g2"]
    h, ldh = hs[-1]
    g_rgb, g_den, g_ins = g_out, _col(g_out, 3), _col(g_out, 4)
    grads = {}
    # ins branch (no gradient into h: h.detach(), :95)
    dg2 = torch.empty(M, HW, **f)
    _dgrad(g_ins, ld, net.io[0], HW, dg2, HW, M, mask=g2, ldm=HW)
    grads["ins_linear"] = _wgrad(g_ins, ld, C, g2, HW, HW, M)
    dq = torch.empty(M, W, **f)
    _dgrad(dg2, HW, net.ih[0], W, dq, W, M)
    grads["ins_feature_linears.0"] = _wgrad(dg2, HW, HW, q, W, W, M)
    grads["ins_feature_linear"] = _wgrad(dq, W, W, h, ldh, W, M)
    # rgb branch
    dg1 = torch.empty(M, HW, **f)
    _dgrad(g_rgb, ld, net.ro[0], HW, dg1, HW, M, mask=g1, ldm=HW)
    grads["rgb_linear"] = _wgrad(g_rgb, ld, 3, g1, HW, HW, M)
    dxr = torch.empty(M, W + inv, **f)
    _dgrad(dg1, HW, net.rh[0], W + inv, dxr, W + inv, M)
    grads["rgb_feature_linears.0"] = _wgrad(dg1, HW, HW, xr, W + inv, W + inv, M)
    grads["rgb_feature_linear"] = _wgrad(dxr, W + inv, W, h, ldh, W, M)
    grads["density_linear"] = _wgrad(g_den, ld, 1, h, ldh, W, M)
    # d h_D = df W_rf + g_sigma w_d, masked by relu'(h_D)
    dy = torch.empty(M, W, **f)
    _dgrad(dxr, W + inv, net.rf[0], W, dy, W, M)
    _dgrad(g_den, ld, net.den[0], W, dy, W, M, mask=h, ldm=ldh, accumulate=True)
    for i in range(net.D - 1, -1, -1):
        Wi, _ = net.trunk[i]
        x, ldx = hs[i]
        n_in = Wi.shape[1]
        grads[f"mlps.{i}"] = _wgrad(dy, W, W, x, ldx, n_in, M)
        if i > 0:                                            # gradient w.r.t. the relu part of the previous layer's output
            nxt = torch.empty(M, W, **f)
            _dgrad(dy, W, Wi, W, nxt, W, M, mask=x, ldm=ldx)
            dy = nxt
    out = []
    for n in net.names:
        mod, kind = n.rsplit(".", 1)
        dW, db = grads[mod]
        out.append(dW if kind == "weight" else db)
    return out


class GenericMLPFunction(torch.autograd.Function):
    """raw = DM_NeRF(x_pos | x_dir) layer by layer; parameters are inputs 3.. in state_dict order."""

    @staticmethod
    def forward(ctx, model, x_pos, x_dir, *params):
        net = _Net(model, [p.detach() for p in params])
        save = {}
        out = forward_layers(net, x_pos, x_dir, save)
        ctx.net, ctx.saved = net, save
        return out

    @staticmethod
    def backward(ctx, g_out):
        grads = backward_layers(ctx.net, ctx.saved, g_out.contiguous())
        ctx.saved = None
        return (None, None, None) + tuple(grads)


def mlp_embedded(model, x, train):
    """``DM_NeRF.forward`` on pre-embedded rows [..., inp + inv] through the generic path."""
    x2 = _lib.f32(x.reshape(-1, x.shape[-1]))
    _lib.require_gpu(x2)
    if x2.shape[-1] != model.input_ch_pts + model.input_ch_views:
        raise ValueError(f"DM_NeRF.forward expects {model.input_ch_pts + model.input_ch_views} input channels")
    x_pos = x2[:, :model.input_ch_pts].contiguous()
    x_dir = x2[:, model.input_ch_pts:].contiguous()
    out = _run(model, x_pos, x_dir, train)
    return out.reshape(*x.shape[:-1], out.shape[-1])


def _run(model, x_pos, x_dir, train):
    params = [p for _, p in model.named_parameters()]
    if train:
        return GenericMLPFunction.apply(model, x_pos, x_dir, *params)
    return forward_layers(_Net(model, [p.detach() for p in params]), x_pos, x_dir, None)


def run_network(model, rays_o, rays_d, z, train=False):
    """pts = o + d z -> embed(pts) | embed(d/|d|) -> model, for any network shape: [N,3], [N,3], [N,S] -> raw [N,S,4+C]."""
    lib = _lib.load()
    rays_o, rays_d, z = _lib.f32(rays_o.reshape(-1, 3)), _lib.f32(rays_d.reshape(-1, 3)), _lib.f32(z)
    _lib.require_gpu(rays_o, rays_d, z)
    N, S = z.shape
    M = N * S
    f = dict(dtype=torch.float32, device=z.device)
    pts, dirs = torch.empty(M, 3, **f), torch.empty(M, 3, **f)
    _lib.check(lib.dmnerf_ray_points(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z), N, S, _lib.ptr(pts), _lib.ptr(dirs), _lib.stream()),
               "dmnerf_ray_points")
    Lp, Lv = (model.input_ch_pts - 3) // 6, (model.input_ch_views - 3) // 6
    if 3 + 6 * Lp != model.input_ch_pts or 3 + 6 * Lv != model.input_ch_views:
        raise NotImplementedError("dm_nerf: the encoders must be get_embedder(multires, 0) outputs (3 + 6 L channels)")
    x_pos, x_dir = torch.empty(M, model.input_ch_pts, **f), torch.empty(M, model.input_ch_views, **f)
    _lib.check(lib.dmnerf_embed(_lib.ptr(pts), M, Lp, _lib.ptr(x_pos), _lib.stream()), "dmnerf_embed")
    _lib.check(lib.dmnerf_embed(_lib.ptr(dirs), M, Lv, _lib.ptr(x_dir), _lib.stream()), "dmnerf_embed")
    out = _run(model, x_pos, x_dir, train)
    return out.reshape(N, S, out.shape[-1])
