"""GPU tests (-m gpu) of network shapes other than the shipped one (config.py:126-138 passes args.netdepth / netwidth /
multires / multires_views through): dm_nerf_amd/generic.py on csrc/gemm_nt.hip / gemm_tn.hip / gemm_chain.hip / generic.hip against the
oracle -- the strided GEMM in its three forms, gemm_nt's forward and data-gradient forms, gemm_tn's weight + bias gradient, the
chained trunk (bit-equal to the layer-by-layer one), dmnerf_ray_embed's rows, the entry points' error behaviour, DM_NeRF.forward,
the whole dm_nerf dict, and the parameter gradients."""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

SHAPES = [dict(D=6, W=128, multires=6, multires_views=2, ins_num=5),          # narrower / shallower, fewer octaves
          dict(D=8, W=192, multires=10, multires_views=4, ins_num=13),        # a width that is not a multiple of 128
          dict(D=10, W=320, multires=8, multires_views=4, ins_num=40),        # deeper / wider, two logit blocks
          dict(D=8, W=160, multires=10, multires_views=4, ins_num=13),        # five out-blocks: the chained trunk's widest, 6-block gradient tiles
          dict(D=6, W=64, multires=4, multires_views=1, ins_num=3)]           # two out-blocks, a 9-column direction encoding


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from dm_nerf_amd import _lib, config as Cfg, generic as G
    from dm_nerf_amd.networks import dm_nerf as M, helpers as H, render as R
    _lib.load()
    return types.SimpleNamespace(M=M, H=H, R=R, Cfg=Cfg, G=G, lib=_lib)


def cpu(t):
    torch.cuda.synchronize()
    return t.detach().cpu()


def maxrel(a, b):
    return float(((a - b).abs() / (1 + b.abs())).max())


def test_strided_gemm_three_forms(A):
    """forward  Y = relu(X W^T + b),  data gradient  (dY W) . [H > 0] (+ accumulate),  weight gradient  dY^T X  (split-K) and
    the column sums -- ragged sizes, operands inside wider buffers -- against float64."""
    g = torch.Generator().manual_seed(1)
    M_, K, N = 1000, 167, 130
    X = torch.randn(M_, K + 5, generator=g).cuda()                       # used through ld = K + 5
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    Y = torch.empty(M_, N + 3, device="cuda")
    A.G._linear(X, K + 5, Wt, b, A.G._col(Y, 3), N + 3, M_, relu=True)
    want = torch.relu(X[:, :K].double() @ Wt.double().t() + b.double())
    assert float((Y[:, 3:].double() - want).abs().max()) <= 2e-5
    dY = torch.randn(M_, N, generator=g).cuda()
    H = torch.randn(M_, K, generator=g).cuda()
    dX = torch.full((M_, K), 0.5, device="cuda")
    A.G._dgrad(dY, N, Wt, K, dX, K, M_, mask=H, ldm=K, accumulate=True)
    want = (dY.double() @ Wt.double() + 0.5) * (H > 0)
    assert float((dX.double() - want).abs().max()) <= 2e-5
    dW, db = A.G._wgrad(dY, N, N, X, K + 5, K, M_)
    want = dY.double().t() @ X[:, :K].double()
    assert float((dW.double() - want).abs().max()) <= 2e-4 * float(want.abs().max())
    assert float((db.double() - dY.double().sum(0)).abs().max()) <= 1e-4
    assert A.G._splits(N, K, 10 ** 6) > 1


@pytest.mark.parametrize("M_,K0,K1,N", [(1000, 167, 0, 130), (257, 63, 0, 128), (4097, 320, 63, 320), (300, 192, 27, 96), (129, 96, 0, 18),
                                         (1000, 1, 0, 320), (513, 384, 0, 400), (700, 64, 0, 3), (300, 160, 27, 352)])
def test_gemm_nt_forward_and_data_gradient_forms(A, M_, K0, K1, N):
    """csrc/gemm_nt.hip (the LDS-DMA / ds_read_b128 GEMM of the generic path) against float64: forward relu(X W^T + b) with a cat
    input as two K ranges, output into a column slice of a wider buffer with its pad columns zeroed; the data-gradient form
    (W^T packed transposed, ReLU-derivative mask, accumulate); ragged M, K not a multiple of 32, N of 1 .. 13 out-blocks (two tiles), both epilogue forms (16-byte stores when the row's
    pad allows, dword stores into a 4 + C wide output)."""
    G = A.G
    g = torch.Generator().manual_seed(M_ + N)
    x0 = G._Act.empty(M_, K0, "cuda"); x0.buf.copy_(torch.randn(M_, x0.ld, generator=g)); x0.buf[:, K0:] = 0
    x1 = None
    if K1:
        x1 = G._Act.empty(M_, K1, "cuda"); x1.buf.copy_(torch.randn(M_, x1.ld, generator=g)); x1.buf[:, K1:] = 0
    Wt = (torch.randn(N, K0 + K1, generator=g) / (K0 + K1) ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    pk = G._Packed(Wt, b, [(0, K0), (K0, K1)] if K1 else [(0, K0)])
    ldc = G._ld(N) + 8
    Y = torch.full((M_, ldc), 7.0, device="cuda")
    G._linear_nt(x0, pk, G._col(Y, 4), ldc, N, G._ld(N), M_, a1=x1, relu=True)
    X = torch.cat([x0.buf[:, :K0]] + ([x1.buf[:, :K1]] if K1 else []), 1).double()
    want = torch.relu(X @ Wt.double().t() + b.double())
    assert float((Y[:, 4:4 + N].double() - want).abs().max()) <= 2e-6 * max(10.0, float(want.abs().max()))
    assert bool((Y[:, 4 + N:4 + G._ld(N)] == 0).all()) and bool((Y[:, :4] == 7.0).all()) and bool((Y[:, 4 + G._ld(N):] == 7.0).all())
    # data gradient through the same layer: dX[:, :K0] = ((dY W)[:, :K0] + 0.5) . [H > 0]
    dY = G._Act.empty(M_, N, "cuda"); dY.buf.copy_(torch.randn(M_, dY.ld, generator=g)); dY.buf[:, N:] = 0
    H = torch.randn(M_, G._ld(K0), generator=g).cuda()
    pt = G._Packed(Wt, None, [(0, N)], n_rows=K0, transposed=True)
    dX = G._Act.empty(M_, K0, "cuda"); dX.buf.fill_(0.5)
    G._linear_nt(dY, pt, dX.buf, dX.ld, K0, dX.ld, M_, mask=H, ldm=H.shape[1], accumulate=True)
    want = (dY.buf[:, :N].double() @ Wt.double()[:, :K0] + 0.5) * (H[:, :K0] > 0)
    assert float((dX.buf[:, :K0].double() - want).abs().max()) <= 2e-6 * max(10.0, float(want.abs().max()))     # (sums of N terms: up to ~25)
    assert bool((dX.buf[:, K0:] == 0).all())
    # the dword epilogue (an output whose rows cannot take 16-byte stores: the heads' column slices of the [M, 4 + C] raw tensor):
    # exactly N columns written into a buffer of odd row length, everything around them untouched
    Z = torch.full((M_, N + 5), 7.0, device="cuda")
    G._linear_nt(x0, pk, G._col(Z, 3), N + 5, N, N, M_, a1=x1, relu=False)
    want = X @ Wt.double().t() + b.double()
    assert float((Z[:, 3:3 + N].double() - want).abs().max()) <= 2e-6 * max(10.0, float(want.abs().max()))
    assert bool((Z[:, :3] == 7.0).all()) and bool((Z[:, 3 + N:] == 7.0).all())


@pytest.mark.parametrize("cfg", SHAPES)
def test_model_forward_and_gradients_other_shapes(A, cfg):
    D, W, ins_num = cfg["D"], cfg["W"], cfg["ins_num"]
    inp, inv = 3 + 6 * cfg["multires"], 3 + 6 * cfg["multires_views"]
    sd = O.make_weights(7 + D, ins_num, W=W, gain=1.5, D=D, input_ch_pts=inp, input_ch_views=inv)
    m = A.M.DM_NeRF(D, W, inp, inv, [4], ins_num)
    m.load_state_dict(sd)
    m = m.cuda()
    assert not m._fused_ok()
    g = torch.Generator().manual_seed(D)
    Mr = 333
    pts = (torch.rand(Mr, 3, generator=g) * 2 - 1) * 5.0
    dirs = torch.nn.functional.normalize(torch.randn(Mr, 3, generator=g), dim=-1)
    x = torch.cat([O.embed(pts, cfg["multires"]), O.embed(dirs, cfg["multires_views"])], -1)
    cot = torch.randn(Mr, 4 + ins_num + 1, generator=g)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want = O.mlp_forward(sdg, x, input_ch_pts=inp, input_ch_views=inv, D=D)
    (want * cot).sum().backward()
    with torch.no_grad():
        y0 = m(x.cuda())
    assert y0.shape == want.shape and maxrel(cpu(y0), want.detach()) <= 1e-5
    m.train()
    y = m(x.cuda())
    assert y.requires_grad and torch.equal(y.detach(), y0)
    (y * cot.cuda()).sum().backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, k
        gw = sdg[k].grad.double()
        err = float((p.grad.cpu().double() - gw).abs().max())
        assert err <= 2e-4 * float(gw.abs().max()) + 1e-7, (k, err, float(gw.abs().max()))
    # h.detach() on the ins branch (dm_nerf.py:95): an ins-only loss reaches only the three ins layers
    m.zero_grad()
    m(x.cuda())[:, 4:].square().sum().backward()
    for k, p in m.named_parameters():
        assert (float(p.grad.abs().max()) > 0) == k.startswith(("ins_feature_linear", "ins_feature_linears.0", "ins_linear")), k


@pytest.mark.parametrize("cfg", SHAPES[:2])
def test_dm_nerf_dict_other_shapes(A, cfg):
    """create_nerf with non-default netdepth / netwidth / multires -> the 10-key dict against the oracle (inference and training)."""
    ins_num = cfg["ins_num"]
    args = types.SimpleNamespace(multires=cfg["multires"], multires_views=cfg["multires_views"], i_embed=0, netdepth=cfg["D"],
                                 netwidth=cfg["W"], ins_num=ins_num, device=torch.device("cuda:0"))
    with pytest.warns(RuntimeWarning, match="generic GEMM path"):          # honest about what it costs: said once, with the factor
        pe, ve, mc, mf, _ = A.Cfg.create_nerf(args)
    inp, inv = pe.out_dim, ve.out_dim
    kw = dict(W=cfg["W"], gain=1.7, sigma_bias=0.3, D=cfg["D"], input_ch_pts=inp, input_ch_views=inv)
    sd_c, sd_f = O.make_weights(31, ins_num, **kw), O.make_weights(32, ins_num, **kw)
    mc.load_state_dict(sd_c); mf.load_state_dict(sd_f)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(70.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(cfg["D"]).choice(480 * 640, 70, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    z = O.z_val_sample(70, 4.0, 15.0, 64).contiguous()
    eargs = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    mc.eval(); mf.eval()
    with torch.no_grad():
        want = O.dm_nerf(rays, sd_c, sd_f, z, perturb=0., multires=cfg["multires"], multires_views=cfg["multires_views"])
        got = {k: cpu(v) for k, v in A.R.dm_nerf(rays.cuda(), pe, ve, mc, mf, z.cuda(), eargs).items()}
        raw_f = cpu(A.R.run_network(mf, rays[0].cuda(), rays[1].cuda(), want['z_vals_fine'].cuda()))
    assert set(got) == set(want) and got['raw_fine'].shape == (70, 192, 4 + ins_num + 1)
    assert maxrel(got['raw_coarse'], want['raw_coarse']) <= 1e-5 and maxrel(raw_f, want['raw_fine']) <= 1e-5
    assert torch.allclose(got['rgb_coarse'], want['rgb_coarse'], rtol=2e-6, atol=2e-6)
    assert torch.allclose(got['ins_coarse'], want['ins_coarse'], rtol=2e-6, atol=2e-6)
    assert float(((got['z_vals_fine'] - want['z_vals_fine']).abs() <= 1e-4).float().mean()) >= 0.999
    assert torch.allclose(got['rgb_fine'], want['rgb_fine'], atol=2e-3)
    # training: gradients of an rgb + ins loss on both levels against autograd of the oracle (same fine depths)
    mc.train(); mf.train()
    targs = types.SimpleNamespace(perturb=0.0, N_importance=128, is_train=True, N_ins=None)
    out = A.R.dm_nerf(rays.cuda(), pe, ve, mc, mf, z.cuda(), targs)
    (out['rgb_fine'].sum() + out['rgb_coarse'].sum() + out['ins_fine'].sum() + out['ins_coarse'].sum()).backward()
    sdc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    sdf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    o = O.dm_nerf(rays, sdc, sdf, z, perturb=0., multires=cfg["multires"], multires_views=cfg["multires_views"],
                  z_fine_override=out['z_vals_fine'].detach().cpu())
    (o['rgb_fine'].sum() + o['rgb_coarse'].sum() + o['ins_fine'].sum() + o['ins_coarse'].sum()).backward()
    for m_, sd_ in ((mc, sdc), (mf, sdf)):
        for k, p in m_.named_parameters():
            gw = sd_[k].grad.double()
            rel_l2 = float((p.grad.cpu().double() - gw).norm() / (gw.norm() + 1e-30))
            assert rel_l2 <= 2e-3, (k, rel_l2)


def test_create_nerf_is_silent_for_the_shipped_shape(A):
    import warnings
    args = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=8, netwidth=256, ins_num=13, device=torch.device("cuda:0"))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        A.Cfg.create_nerf(args)


@pytest.mark.parametrize("D,W,multires,M_", [(6, 128, 6, 1000), (8, 160, 10, 4097), (6, 64, 4, 129), (8, 128, 10, 70000)])
def test_chained_trunk_equals_the_layer_by_layer_trunk_bit_for_bit(A, D, W, multires, M_, monkeypatch):
    """csrc/gemm_chain.hip (inference, widths up to 160: the trunk as ONE launch, activations LDS-resident) against the same network
    layer by layer on gemm_nt: the chunk order, the k order inside a chunk and the bias-first accumulation are the same, so the raw
    outputs are EQUAL -- ragged last tile, several tiles per workgroup, skip at layer 4 (its [h, pts] input from two LDS regions)."""
    ins_num, mv = 9, 4
    inp, inv = 3 + 6 * multires, 3 + 6 * mv
    sd = O.make_weights(40 + D, ins_num, W=W, gain=1.5, D=D, input_ch_pts=inp, input_ch_views=inv)
    m = A.M.DM_NeRF(D, W, inp, inv, [4], ins_num)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    assert not m._fused_ok() and bool(A.lib.load().dmnerf_mlp_chain_supported(W, inp))
    g = torch.Generator().manual_seed(M_)
    pts = (torch.rand(M_, 3, generator=g) * 2 - 1) * 5.0
    dirs = torch.nn.functional.normalize(torch.randn(M_, 3, generator=g), dim=-1)
    x = torch.cat([O.embed(pts, multires), O.embed(dirs, mv)], -1).cuda()
    with torch.no_grad():
        monkeypatch.setenv("DMNERF_GENERIC_CHAIN", "0")
        want = m(x)
        monkeypatch.setenv("DMNERF_GENERIC_CHAIN", "1")
        got = m(x)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (M_, 4 + ins_num + 1)
    assert torch.equal(got, want), float((got - want).abs().max())
    if M_ <= 1000:                                                       # ... and both agree with the oracle
        ref = O.mlp_forward(sd, x.cpu(), input_ch_pts=inp, input_ch_views=inv, D=D)
        assert maxrel(got.cpu(), ref) <= 1e-5


@pytest.mark.parametrize("N,S,Lp,Lv,ldp,ldv", [(7, 9, 10, 4, 64, 32),        # 63 samples: one ragged workgroup
                                                (33, 64, 10, 4, 64, 32),      # 2112 samples = 33 tiles of 64
                                                (5, 13, 6, 2, 39, 15),        # rows that are not 16-byte multiples: the dword copy
                                                (3, 50, 4, 0, 32, 4)])
def test_ray_embed_rows_against_the_reference_formulas(A, N, S, Lp, Lv, ldp, ldv):
    """dmnerf_ray_embed (csrc/gemm_nt.hip): pts = o + d z (render.py:49), viewdirs = d / |d| (render.py:37), both encodings
    (dm_nerf.py:22-38) as row-padded operands: [x, sin(2^k x), cos(2^k x) ...] per row, pad columns zero, nothing beyond row M - 1."""
    import ctypes
    lib = A.lib.load()
    g = torch.Generator().manual_seed(N * 100 + S)
    ro = (torch.rand(N, 3, generator=g) * 2 - 1) * 3.0
    rd = torch.randn(N, 3, generator=g)
    z = torch.rand(N, S, generator=g) * 6 + 0.5
    M_ = N * S
    xp = torch.full((M_ + 3, ldp), 7.0, device="cuda")                   # (three guard rows behind the last sample)
    xv = torch.full((M_ + 3, ldv), 7.0, device="cuda")
    d_ro, d_rd, d_z = ro.cuda(), rd.cuda(), z.cuda().contiguous()
    A.lib.check(lib.dmnerf_ray_embed(A.lib.ptr(d_ro), A.lib.ptr(d_rd), A.lib.ptr(d_z), N, S, Lp, Lv, A.lib.ptr(xp), ldp,
                                     A.lib.ptr(xv), ldv, A.lib.stream()), "dmnerf_ray_embed")
    xp, xv = cpu(xp), cpu(xv)
    pts = (ro[:, None, :] + rd[:, None, :] * z[..., None]).reshape(M_, 3)
    vd = (rd / rd.norm(dim=-1, keepdim=True))[:, None, :].expand(N, S, 3).reshape(M_, 3)
    assert torch.equal(xp[:M_, :3], pts)                                 # (one multiply, one add: bit-equal)
    assert float((xv[:M_, :3] - vd).abs().max()) <= 2e-7
    # the encodings against float64 sin / cos of the (float32) coordinate the kernel itself produced
    for x, L, ld in ((xp, Lp, ldp), (xv, Lv, ldv)):
        base = x[:M_, :3].double()
        for k in range(L):
            assert float((x[:M_, 3 + 6 * k:6 + 6 * k].double() - torch.sin(base * 2.0 ** k)).abs().max()) <= 1.2e-7
            assert float((x[:M_, 6 + 6 * k:9 + 6 * k].double() - torch.cos(base * 2.0 ** k)).abs().max()) <= 1.2e-7
        assert float(x[:M_, 3 + 6 * L:].abs().max() if ld > 3 + 6 * L else 0.0) == 0.0
        assert torch.all(x[M_:] == 7.0)


@pytest.mark.parametrize("M_,n_out,n_in", [(1000, 128, 128), (4097, 64, 160), (257, 3, 64), (70001, 13, 64), (300, 128, 27), (31, 1, 128),
                                           (5000, 320, 63), (2048, 192, 192), (1, 5, 7), (9000, 40, 288), (3000, 160, 128), (777, 96, 192)])
def test_gemm_tn_weight_and_bias_gradient(A, M_, n_out, n_in):
    """dmnerf_gemm_tn (csrc/gemm_tn.hip): dW = dy^T x and db = column sums of dy over M sample-major rows -- every tile shape class
    (1 x 4, 1 x 8, 4 x 1, 8 x 1, 2 x 2 .. 6 x 6 blocks, several output tiles), ragged last chunk, fewer chunks than slices, a column range of
    a wider dW (the cat inputs' halves), pad columns holding other data; against float64, and bit-reproducible from run to run."""
    G = A.G
    g = torch.Generator().manual_seed(M_ + n_out)
    dy, x = G._Act.empty(M_, n_out, "cuda"), G._Act.empty(M_, n_in, "cuda")
    dy.buf.copy_(torch.randn(dy.buf.shape, generator=g))               # (pad columns too: they must not reach dW / db)
    x.buf.copy_(torch.randn(x.buf.shape, generator=g))
    ldw = n_in + 5
    dW = torch.full((n_out, ldw), 7.0, device="cuda")
    db = G._wgrad_tn(dy, n_out, x, n_in, M_, dW, ldw, True)
    dW2 = torch.full((n_out, ldw), 7.0, device="cuda")
    db2 = G._wgrad_tn(dy, n_out, x, n_in, M_, dW2, ldw, True)
    dW3 = torch.empty(n_out, n_in, device="cuda")
    assert G._wgrad_tn(dy, n_out, x, n_in, M_, dW3, n_in, False) is None
    a64, b64 = cpu(dy.buf)[:, :n_out].double(), cpu(x.buf)[:, :n_in].double()
    want_W, want_b = a64.T @ b64, a64.sum(0)
    got_W, got_b = cpu(dW), cpu(db)
    assert torch.all(got_W[:, n_in:] == 7.0)
    tol = 4e-6 * max(1.0, M_ ** 0.5)
    assert float((got_W[:, :n_in].double() - want_W).abs().max()) <= tol
    assert float((got_b.double() - want_b).abs().max()) <= tol
    assert torch.equal(got_W, cpu(dW2)) and torch.equal(got_b, cpu(db2)) and torch.equal(cpu(dW3), got_W[:, :n_in])


def test_generic_entry_points_reject_what_they_cannot_do(A):
    """The C ABI's error behaviour on this path (include/dmnerf_hip.h): DMNERF_E_ARG with a message, nothing launched -- a workspace
    that is too small, operand rows that are not 16-byte aligned, a width the chained trunk has no room for, a layer whose packed
    weights do not match its inputs, layer 0 reading activations that do not exist yet; M = 0 is not an error (zero gradients)."""
    import ctypes
    lib, L = A.lib.load(), A.lib
    dev = "cuda"
    dy, x = torch.randn(64, 32, device=dev), torch.randn(64, 64, device=dev)
    dW, db = torch.full((32, 64), 7.0, device=dev), torch.full((32,), 7.0, device=dev)
    need = int(lib.dmnerf_gemm_tn_ws_floats(32, 64, 64))
    assert need > 0 and int(lib.dmnerf_gemm_tn_ws_floats(0, 64, 64)) == 0
    ws = torch.empty(need, device=dev)

    def tn(dy_=dy, ldy=32, x_=x, ldx=64, M_=64, ws_floats=need, ldw=64):
        return lib.dmnerf_gemm_tn(L.ptr(dy_), ldy, dy_.numel(), 32, L.ptr(x_), ldx, x_.numel(), 64, M_, L.ptr(dW), ldw, L.ptr(db), L.ptr(ws), ws_floats, L.stream())
    assert tn(ws_floats=need - 1) == -1 and "workspace" in L.last_error()
    assert tn(ldy=30) == -1 and "16-byte" in L.last_error()
    assert tn(x_=x.reshape(-1)[1:]) == -1 and "16-byte" in L.last_error()
    assert tn(ldw=63) == -1
    torch.cuda.synchronize()
    assert torch.all(dW == 7.0) and torch.all(db == 7.0)                   # nothing was launched
    assert tn(M_=0) == 0
    torch.cuda.synchronize()
    assert torch.all(dW == 0.0) and torch.all(db == 0.0)
    assert tn() == 0
    torch.cuda.synchronize()
    assert float((dW.double().cpu() - dy.double().cpu().T @ x.double().cpu()).abs().max()) <= 1e-4

    # the chained trunk
    assert lib.dmnerf_mlp_chain_supported(128, 63) == 1 and lib.dmnerf_mlp_chain_supported(160, 63) == 1
    assert lib.dmnerf_mlp_chain_supported(192, 63) == 0 and lib.dmnerf_mlp_chain_supported(100, 63) == 0 and lib.dmnerf_mlp_chain_supported(128, 0) == 0
    W_, inp = 128, 63
    G = A.G
    xp = G._Act.empty(256, inp, dev)
    p0 = G._Packed(torch.randn(W_, inp, device=dev), torch.zeros(W_, device=dev), [(0, inp)])
    p1 = G._Packed(torch.randn(W_, W_, device=dev), torch.zeros(W_, device=dev), [(0, W_)])
    out = G._Act.empty(256, W_, dev)

    def chain(layers, width=W_, ldo=None):
        arr = (L.ChainLayer * len(layers))(*layers)
        return lib.dmnerf_mlp_chain(L.ptr(xp.buf), xp.ld, xp.buf.numel(), inp, arr, len(layers), width, L.ptr(out.buf), out.ld if ldo is None else ldo, 256, L.stream())
    l0 = L.ChainLayer(p0.w.data_ptr(), p0.b.data_ptr(), p0.ldb, 0, 1, 1)
    l1 = L.ChainLayer(p1.w.data_ptr(), p1.b.data_ptr(), p1.ldb, 1, 0, 1)
    assert chain([l0, l1]) == 0
    assert chain([l1, l0]) == -1 and "layer 0" in L.last_error()
    assert chain([l0, L.ChainLayer(p1.w.data_ptr(), p1.b.data_ptr(), p1.ldb, 1, 1, 1)]) == -1 and "bad operands" in L.last_error()
    assert chain([l0, l1], width=192) == -1 and "not supported" in L.last_error()
    assert chain([l0, l1], ldo=64) == -1
    assert chain([l0] * 17) == -1
    torch.cuda.synchronize()
