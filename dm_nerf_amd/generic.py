"""DM_NeRF for network shapes other than the one the fused kernels are specialised for (D = 8, W = 256, skips = [4],
multires 10 / 4): the layer-by-layer path on ``csrc/gemm_nt.hip`` / ``gemm_tn.hip`` / ``gemm_chain.hip``.

``create_nerf`` (config.py:126-138) passes ``args.netdepth / netwidth / multires / multires_views`` through; no shipped config
changes them, but a configuration that does must run, not raise.  This module chains one f32-MFMA GEMM per linear
layer exactly as ``DM_NeRF.forward`` does (networks/dm_nerf.py:80-106) -- skip concat ``[h, pts]`` (:87) and ``[rgb_feature, dirs]``
(:90) as a second K range of the layer that reads them, activation-free feature linears (:89,96), ``h.detach()`` on the ins
branch (:95), output ``cat[rgb, density, ins]`` (:105) -- and, for training, as its autograd would (data gradient with the ReLU
mask in the GEMM epilogue, split-K weight gradients, column-sum bias gradients).  Forward and data-gradient products run on
``csrc/gemm_nt.hip`` (LDS-DMA operand ring, ``ds_read_b128``, one workgroup = 128 samples x all outputs); weight and bias gradients
on ``csrc/gemm_tn.hip`` (both sample-major operands as 32-sample LDS-DMA chunks, bias sums on the dy registers, deterministic split
over the samples); in inference the trunk of a network up to 160 wide is ONE launch of ``csrc/gemm_chain.hip`` (activations
LDS-resident from layer to layer, bit-equal to the layer-by-layer trunk; ``DMNERF_GENERIC_CHAIN=0`` turns it off).  The strided
kernel of ``csrc/generic.hip`` remains for operands whose rows are not 16-byte aligned.  Used ONLY when
``DM_NeRF._fused_ok()`` is false; the shipped shape never comes here.  No torch compute ops: tensors are allocated with torch, every FLOP runs in
``libdmnerf_hip.so``.
"""
import os

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


def _ld(cols):
    """Row length of an activation matrix with ``cols`` columns: rows stay 16-byte aligned (the LDS-DMA fetches 16 bytes per lane)."""
    return (cols + 3) // 4 * 4


class _Act:
    """An activation matrix [M, ld] with ``cols`` logical columns; the pad columns [cols, ld) hold zeros (the GEMM that reads it may
    run over them; they meet zero weights, but a NaN bit pattern left in uninitialised memory would not vanish)."""

    def __init__(self, buf, cols):
        self.buf, self.cols, self.ld = buf, cols, buf.shape[1]

    @staticmethod
    def empty(M, cols, device):
        return _Act(torch.empty(M, _ld(cols), dtype=torch.float32, device=device), cols)


class _Packed:
    """A layer's weight matrix in gemm_nt's form (csrc/gemm_nt.hip::pack_nt_kernel): rows padded to the kernel's tile, the K
    ranges of a cat input each padded to 32, zeros elsewhere; the bias padded likewise."""

    def __init__(self, W, bias, ranges, n_rows=None, transposed=False):
        lib = _lib.load()
        (c0, k0), (c1, k1) = (list(ranges) + [(0, 0)])[:2]
        n_rows = W.shape[0] if n_rows is None else n_rows
        nbb = int(lib.dmnerf_gemm_nt_blocks(n_rows))
        tiles = ((n_rows + 31) // 32 + nbb - 1) // nbb
        self.rows_pad, self.ldb = tiles * nbb * 32, 32 * ((k0 + 31) // 32 + (k1 + 31) // 32)
        self.k0, self.k1, self.n = k0, k1, n_rows
        self.w = torch.empty(self.rows_pad * self.ldb, dtype=torch.float32, device=W.device)
        self.b = torch.empty(self.rows_pad, dtype=torch.float32, device=W.device) if bias is not None else None
        _lib.check(lib.dmnerf_pack_nt(_lib.ptr(W), W.stride(0), n_rows, c0, k0, c1, k1, int(transposed), _lib.ptr(bias), _lib.ptr(self.w),
                                      self.rows_pad, self.ldb, _lib.ptr(self.b), _lib.stream()), "dmnerf_pack_nt")


def _floats_from(t, col=0):
    """Floats from element [0, col] of a contiguous 2-D tensor to the end of its allocation (the kernel's descriptor bound)."""
    return t.numel() - col


def _linear_nt(a0, pk, out, ldc, n_store, n_zero, M, a1=None, relu=False, mask=None, ldm=0, accumulate=False, out_floats=None):
    """out[:, :n_store] = act(A W^T + b) with A = a0 (| a1: the second K range of a cat input), W packed (``_Packed``); columns
    [n_store, n_zero) of ``out`` are zeroed (its pad columns)."""
    assert pk.k0 == a0.cols and pk.k1 == (a1.cols if a1 is not None else 0)
    _lib.check(_lib.load().dmnerf_gemm_nt(_lib.ptr(a0.buf), a0.ld, _floats_from(a0.buf), pk.k0,
                                          _lib.ptr(a1.buf) if a1 is not None else None, a1.ld if a1 is not None else 0,
                                          _floats_from(a1.buf) if a1 is not None else 0, pk.k1,
                                          _lib.ptr(pk.w), pk.w.numel(), pk.ldb, _lib.ptr(pk.b), _lib.ptr(out), ldc, n_store, n_zero, M, int(relu),
                                          _lib.ptr(mask), ldm, int(accumulate), _lib.stream()), "dmnerf_gemm_nt")


def _tn_ok(t):
    """Whether an operand's rows are what gemm_tn's LDS-DMA needs (16-byte aligned: every ``_Act`` is)."""
    return t.ld % 4 == 0 and t.buf.data_ptr() % 16 == 0


def _wgrad_tn(dy, n_out, x, n_in, M, dW, ldw, want_bias):
    """dW = dy^T x (+ db = column sums of dy) on csrc/gemm_tn.hip: one product launch + one reduction."""
    lib = _lib.load()
    dev = x.buf.device
    db = torch.empty(n_out, dtype=torch.float32, device=dev) if want_bias else None
    n_ws = int(lib.dmnerf_gemm_tn_ws_floats(n_out, n_in, M))
    if n_ws < 0:
        raise RuntimeError("dmnerf_gemm_tn_ws_floats: " + _lib.last_error())
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dev)
    _lib.check(lib.dmnerf_gemm_tn(_lib.ptr(dy.buf), dy.ld, _floats_from(dy.buf), n_out, _lib.ptr(x.buf), x.ld, _floats_from(x.buf), n_in, M,
                                  _lib.ptr(dW), ldw, _lib.ptr(db), _lib.ptr(ws), ws.numel(), _lib.stream()), "dmnerf_gemm_tn")
    return db


def _wgrad_w(dy, n_out, x, n_in, M, dW=None, ldw=None, want_bias=False):
    """dW [n_out, n_in] = dy[:, :n_out]^T x[:, :n_in] over the M samples (``dy`` / ``x``: ``_Act``); into ``dW`` with row stride ``ldw``
    when given (one column range of a cat input's weight).  ``want_bias``: -> (dW, db) with db = column sums of dy."""
    if dW is None:
        dW = torch.empty(n_out, n_in, dtype=torch.float32, device=x.buf.device)
        ldw = n_in
    if M > 0 and _tn_ok(dy) and _tn_ok(x):
        db = _wgrad_tn(dy, n_out, x, n_in, M, dW, ldw, want_bias)
    else:                                                    # rows that are not 16-byte aligned: the strided kernel
        _gemm(dy.buf, 1, dy.ld, x.buf, x.ld, 1, dW, ldw, n_out, n_in, M, splits=_splits(n_out, n_in, M))
        db = _colsum(dy, n_out, M) if want_bias else None
    return (dW, db) if want_bias else dW


def _colsum(dy, n_out, M):
    """db [n_out] = column sums of dy (the bias gradient)."""
    db = torch.empty(n_out, dtype=torch.float32, device=dy.buf.device)
    slices = int(max(1, min(256, (M + 4095) // 4096)))
    ws = torch.empty(slices * n_out, dtype=torch.float32, device=dy.buf.device)
    _lib.check(_lib.load().dmnerf_colsum(_lib.ptr(dy.buf), dy.ld, M, n_out, _lib.ptr(db), _lib.ptr(ws), slices, _lib.stream()), "dmnerf_colsum")
    return db


def _wgrad_act(dy, n_out, x, n_in, M):
    """-> (dW, db) of a layer: one gemm_tn launch (the bias gradient rides on the dy operand's registers)."""
    return _wgrad_w(dy, n_out, x, n_in, M, want_bias=True)


# ---- the three products of a linear layer on the STRIDED kernel (csrc/generic.hip::gemm_kernel), plain row-major operands with
# any leading dimension: the weight gradient's form (both operands sample-major) and the reference point of gemm_nt's tests
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
    a, b = _Act(dy, n_out), _Act(x, n_in)
    a.ld, b.ld = ldy, ldx
    return _wgrad_w(a, n_out, b, n_in, M, want_bias=True)


def _slice_act(src, ld_src, col, n, M):
    """Columns [col, col + n) of a row-major matrix as a gemm_nt operand: its own row-padded buffer, pad columns zero."""
    t = _Act.empty(M, n, src.device)
    _lib.check(_lib.load().dmnerf_copy_cols_pad(_lib.ptr(src[:, col:]), ld_src, _lib.ptr(t.buf), t.ld, M, n, t.ld, _lib.stream()), "dmnerf_copy_cols_pad")
    return t


def _col(t, c):
    """Pointer to column ``c`` of a contiguous 2-D tensor (a view: plumbing, no copy)."""
    return t[:, c:]


class _Net:
    """The parameters of one DM_NeRF in forward order, with the dimensions the chain needs, and their packed forms."""

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
        self._fwd = self._bwd = None

    def after_skip(self, i):
        """Whether trunk layer ``i`` reads cat[h, pts] (the layer BEFORE it is in ``skips``, dm_nerf.py:86-87)."""
        return (i - 1) in self.skips

    def packed_forward(self):
        """Every layer's weights in gemm_nt's form; a cat input is two K ranges (h | pts, rgb_feature | dirs)."""
        if self._fwd is None:
            W, inp, inv = self.W, self.inp, self.inv
            f = {}
            for i, (Wi, bi) in enumerate(self.trunk):
                f[f"mlps.{i}"] = _Packed(Wi, bi, [(0, W), (W, inp)] if self.after_skip(i) else [(0, Wi.shape[1])])
            f["rf"] = _Packed(*self.rf, [(0, W)])
            f["inf"] = _Packed(*self.inf_, [(0, W)])
            f["rh"] = _Packed(*self.rh, [(0, W), (W, inv)])
            f["ih"] = _Packed(*self.ih, [(0, W)])
            f["den"] = _Packed(*self.den, [(0, W)])
            f["ro"] = _Packed(*self.ro, [(0, self.HW)])
            f["io"] = _Packed(*self.io, [(0, self.HW)])
            self._fwd = f
        return self._fwd

    def packed_backward(self):
        """W^T of the layers the data gradient passes through (rows = the layer's W-wide input, K = its outputs)."""
        if self._bwd is None:
            W, HW, C = self.W, self.HW, self.C
            t = lambda Wt, n_in, n_out: _Packed(Wt, None, [(0, n_out)], n_rows=n_in, transposed=True)
            b = {"io": t(self.io[0], HW, C), "ih": t(self.ih[0], W, HW), "ro": t(self.ro[0], HW, 3), "rh": t(self.rh[0], W, HW),
                 "rf": t(self.rf[0], W, W), "den": t(self.den[0], W, 1)}
            for i in range(1, self.D):
                b[f"mlps.{i}"] = t(self.trunk[i][0], W, W)
            self._bwd = b
        return self._bwd


def forward_layers(net, x_pos, x_dir, save):
    """raw [M, 4 + C] for embedded inputs ``x_pos`` / ``x_dir`` (``_Act``: row-padded, pad columns zero); ``save``: dict filled with what
    backward needs.  One gemm_nt launch per linear layer; the cats of dm_nerf.py:87,90 are the second K range of the layer that reads
    them, never materialised."""
    M = x_pos.buf.shape[0]
    dev = x_pos.buf.device
    W, HW, C = net.W, net.HW, net.C
    pk = net.packed_forward()
    h = None
    hs = []                                                  # the ReLU output of every trunk layer
    lib = _lib.load()
    chained = save is None and bool(lib.dmnerf_mlp_chain_supported(W, net.inp)) and net.D <= 16 \
        and os.environ.get("DMNERF_GENERIC_CHAIN", "1") != "0"
    if chained:
        # inference on a narrow network: the whole trunk as ONE launch, activations LDS-resident from layer to layer (csrc/gemm_chain.hip)
        arr = (_lib.ChainLayer * net.D)()
        for i in range(net.D):
            p_ = pk[f"mlps.{i}"]
            arr[i] = _lib.ChainLayer(p_.w.data_ptr(), p_.b.data_ptr(), p_.ldb, int(i > 0), int(i == 0 or net.after_skip(i)), 1)
        h = _Act.empty(M, W, dev)
        _lib.check(lib.dmnerf_mlp_chain(_lib.ptr(x_pos.buf), x_pos.ld, _floats_from(x_pos.buf), net.inp, arr, net.D, W, _lib.ptr(h.buf), h.ld, M,
                                        _lib.stream()), "dmnerf_mlp_chain")
    for i in range(net.D if not chained else 0):
        out = _Act.empty(M, W, dev)
        if i == 0:
            _linear_nt(x_pos, pk["mlps.0"], out.buf, out.ld, W, out.ld, M, relu=True)
        else:
            _linear_nt(h, pk[f"mlps.{i}"], out.buf, out.ld, W, out.ld, M, a1=x_pos if net.after_skip(i) else None, relu=True)
        h = out
        if save is not None:                                 # (inference keeps only the current layer's input alive)
            hs.append(h)
    xr = _Act.empty(M, W, dev)                               # rgb_feature (no activation, :89); cat[., dirs] (:90) = the next layer's 2nd range
    _linear_nt(h, pk["rf"], xr.buf, xr.ld, W, xr.ld, M)
    g1 = _Act.empty(M, HW, dev)
    _linear_nt(xr, pk["rh"], g1.buf, g1.ld, HW, g1.ld, M, a1=x_dir, relu=True)
    q = _Act.empty(M, W, dev)                                # ins_feature(h.detach())  (:95-96)
    _linear_nt(h, pk["inf"], q.buf, q.ld, W, q.ld, M)
    g2 = _Act.empty(M, HW, dev)
    _linear_nt(q, pk["ih"], g2.buf, g2.ld, HW, g2.ld, M, relu=True)
    out = torch.empty(M, 4 + C, dtype=torch.float32, device=dev)     # cat[rgb, density, ins]  (:105)
    _linear_nt(g1, pk["ro"], out, 4 + C, 3, 3, M)
    _linear_nt(h, pk["den"], _col(out, 3), 4 + C, 1, 1, M)
    _linear_nt(g2, pk["io"], _col(out, 4), 4 + C, C, C, M)
    if save is not None:
        save.update(x_pos=x_pos, x_dir=x_dir, hs=hs, xr=xr, g1=g1, q=q, g2=g2)
    return out


def backward_layers(net, save, g_out):
    """Parameter gradients (list in ``named_parameters`` order) for the upstream gradient g_out [M, 4 + C]: data gradients on gemm_nt
    (W^T packed, the ReLU derivative as the epilogue's mask), weight and bias gradients of a layer as ONE split-K "TN" product on
    gemm_tn (csrc/gemm_tn.hip; a cat input's two column ranges of dW from their own sources)."""
    g_out = _lib.f32(g_out)
    M = g_out.shape[0]
    dev = g_out.device
    W, inp, inv, HW, C = net.W, net.inp, net.inv, net.HW, net.C
    ld = 4 + C
    x_pos, x_dir, hs, xr, g1, q, g2 = (save[k] for k in ("x_pos", "x_dir", "hs", "xr", "g1", "q", "g2"))
    h = hs[-1]
    pt = net.packed_backward()
    grads = {}
    # the three column slices of the upstream gradient (cat[rgb, density, ins], :105) as operands with 16-byte aligned rows
    g_rgb, g_den, g_ins = _slice_act(g_out, ld, 0, 3, M), _slice_act(g_out, ld, 3, 1, M), _slice_act(g_out, ld, 4, C, M)
    # ins branch (no gradient into h: h.detach(), :95)
    dg2 = _Act.empty(M, HW, dev)
    _linear_nt(g_ins, pt["io"], dg2.buf, dg2.ld, HW, dg2.ld, M, mask=g2.buf, ldm=g2.ld)
    grads["ins_linear"] = _wgrad_act(g_ins, C, g2, HW, M)
    dq = _Act.empty(M, W, dev)
    _linear_nt(dg2, pt["ih"], dq.buf, dq.ld, W, dq.ld, M)
    grads["ins_feature_linears.0"] = _wgrad_act(dg2, HW, q, W, M)
    grads["ins_feature_linear"] = _wgrad_act(dq, W, h, W, M)
    # rgb branch
    dg1 = _Act.empty(M, HW, dev)
    _linear_nt(g_rgb, pt["ro"], dg1.buf, dg1.ld, HW, dg1.ld, M, mask=g1.buf, ldm=g1.ld)
    grads["rgb_linear"] = _wgrad_act(g_rgb, 3, g1, HW, M)
    dxr = _Act.empty(M, W, dev)                              # gradient w.r.t. rgb_feature (the dirs columns of the cat need none)
    _linear_nt(dg1, pt["rh"], dxr.buf, dxr.ld, W, dxr.ld, M)
    dWrh = torch.empty(HW, W + inv, dtype=torch.float32, device=dev)     # cat[rgb_feature, dirs]: two column ranges, two sources
    _, db_rh = _wgrad_w(dg1, HW, xr, W, M, dW=dWrh, ldw=W + inv, want_bias=True)
    _wgrad_w(dg1, HW, x_dir, inv, M, dW=_col(dWrh, W), ldw=W + inv)
    grads["rgb_feature_linears.0"] = (dWrh, db_rh)
    grads["rgb_feature_linear"] = _wgrad_act(dxr, W, h, W, M)
    grads["density_linear"] = _wgrad_act(g_den, 1, h, W, M)
    # d h_D = df W_rf + g_sigma w_d, masked by relu'(h_D)
    dy = _Act.empty(M, W, dev)
    _linear_nt(dxr, pt["rf"], dy.buf, dy.ld, W, dy.ld, M)
    _linear_nt(g_den, pt["den"], dy.buf, dy.ld, W, dy.ld, M, mask=h.buf, ldm=h.ld, accumulate=True)
    for i in range(net.D - 1, -1, -1):
        Wi, _ = net.trunk[i]
        x = hs[i - 1] if i > 0 else x_pos
        n_in = Wi.shape[1]
        if net.after_skip(i):                                # cat[h, pts]: the two column ranges of dW from their own sources
            dW = torch.empty(W, n_in, dtype=torch.float32, device=dev)
            _, db = _wgrad_w(dy, W, x, W, M, dW=dW, ldw=n_in, want_bias=True)
            _wgrad_w(dy, W, x_pos, inp, M, dW=_col(dW, W), ldw=n_in)
            grads[f"mlps.{i}"] = (dW, db)
        else:
            grads[f"mlps.{i}"] = _wgrad_act(dy, W, x, n_in, M)
        if i > 0:                                            # gradient w.r.t. the relu output of the previous layer
            nxt = _Act.empty(M, W, dev)
            _linear_nt(dy, pt[f"mlps.{i}"], nxt.buf, nxt.ld, W, nxt.ld, M, mask=x.buf, ldm=x.ld)
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
    M = x2.shape[0]
    x_pos = _slice_act(x2, x2.shape[1], 0, model.input_ch_pts, M)
    x_dir = _slice_act(x2, x2.shape[1], model.input_ch_pts, model.input_ch_views, M)
    out = _run(model, x_pos, x_dir, train)
    return out.reshape(*x.shape[:-1], out.shape[-1])


def _run(model, x_pos, x_dir, train):
    params = [p for _, p in model.named_parameters()]
    if train:
        return GenericMLPFunction.apply(model, x_pos, x_dir, *params)
    # inference: the packed weights are kept with the model until a parameter changes (the key DM_NeRF.blob() uses)
    key = tuple((p.data_ptr(), p._version) for p in params)
    cached = getattr(model, "_generic_net", None)
    if cached is None or cached[0] != key:
        cached = (key, _Net(model, [p.detach() for p in params]))
        model._generic_net = cached
    return forward_layers(cached[1], x_pos, x_dir, None)


def run_network(model, rays_o, rays_d, z, train=False):
    """pts = o + d z -> embed(pts) | embed(d/|d|) -> model, for any network shape: [N,3], [N,3], [N,S] -> raw [N,S,4+C]."""
    lib = _lib.load()
    rays_o, rays_d, z = _lib.f32(rays_o.reshape(-1, 3)), _lib.f32(rays_d.reshape(-1, 3)), _lib.f32(z)
    _lib.require_gpu(rays_o, rays_d, z)
    N, S = z.shape
    M = N * S
    Lp, Lv = (model.input_ch_pts - 3) // 6, (model.input_ch_views - 3) // 6
    if 3 + 6 * Lp != model.input_ch_pts or 3 + 6 * Lv != model.input_ch_views:
        raise NotImplementedError("dm_nerf: the encoders must be get_embedder(multires, 0) outputs (3 + 6 L channels)")
    # points, view directions and both encodings in one launch, straight into the row-padded operands of the first layers
    x_pos, x_dir = _Act.empty(M, model.input_ch_pts, z.device), _Act.empty(M, model.input_ch_views, z.device)
    _lib.check(lib.dmnerf_ray_embed(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z), N, S, Lp, Lv, _lib.ptr(x_pos.buf), x_pos.ld,
                                    _lib.ptr(x_dir.buf), x_dir.ld, _lib.stream()), "dmnerf_ray_embed")
    out = _run(model, x_pos, x_dir, train)
    return out.reshape(N, S, out.shape[-1])
