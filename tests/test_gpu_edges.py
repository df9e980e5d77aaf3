"""GPU tests (-m gpu) of the shapes and call patterns the reference accepts beyond its shipped configs, and of the
boundary's sharp edges: N_importance = 0, DM_NeRF.forward on pre-embedded rows in training mode, a frozen fine model,
caller-supplied random draws of the wrong shape, in-place ``.data`` updates, a manipulation with the maximum number of
moved objects, per-thread error messages."""
import threading
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from dm_nerf_amd import _lib, autograd
    from dm_nerf_amd.networks import dm_nerf as M, helpers as H, manipulator as MA, render as R
    _lib.load()
    return types.SimpleNamespace(M=M, H=H, R=R, G=autograd, MA=MA, lib=_lib)


def cpu(t):
    torch.cuda.synchronize()
    return t.detach().cpu()


def maxrel(a, b):
    return float(((a - b).abs() / (1 + b.abs())).max())


def model_from(A, sd, ins_num, train=False):
    m = A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
    m.load_state_dict(sd)
    m = m.cuda()
    return m.train() if train else m.eval()


def _rays(n, seed, theta=33.0):
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(theta, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(seed).choice(480 * 640, n, replace=False))
    return torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)


@pytest.mark.parametrize("perturb", [0.0, 1.0])
def test_n_importance_zero(A, perturb):
    """config.py:43 allows N_importance = 0: sample_pdf returns [N, 0], the merged depths are the coarse ones and the
    fine network runs on them (render.py:66-83).  Whole dict against the oracle, inference and training mode."""
    ins_num, N = 13, 50
    sd_c, sd_f = O.make_weights(85, ins_num, gain=1.7, sigma_bias=0.3), O.make_weights(86, ins_num, gain=1.7, sigma_bias=0.3)
    rays = _rays(N, 85)
    z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous()
    t_rand = torch.rand(N, 64, generator=torch.Generator().manual_seed(1)) if perturb > 0 else None
    args = types.SimpleNamespace(perturb=perturb, N_importance=0, is_train=False, N_ins=None)
    with torch.no_grad():
        want = O.dm_nerf(rays, sd_c, sd_f, z, perturb=perturb, N_importance=0, t_rand=t_rand)
        mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
        got = A.R.dm_nerf(rays.cuda(), None, None, mc, mf, z.cuda(), args, t_rand=None if t_rand is None else t_rand.cuda())
    assert set(got) == set(want)
    assert got['raw_fine'].shape == (N, 64, 18) and got['z_vals_fine'].shape == (N, 64)
    assert torch.equal(cpu(got['z_vals_fine']), want['z_vals_fine']) and torch.equal(cpu(got['z_vals_coarse']), want['z_vals_coarse'])
    for k in ('raw_coarse', 'raw_fine'):
        assert maxrel(cpu(got[k]), want[k]) <= 1e-5, k
    for k in ('rgb_coarse', 'rgb_fine', 'ins_coarse', 'ins_fine', 'depth_coarse', 'depth_fine'):
        assert torch.allclose(cpu(got[k]), want[k], rtol=2e-6, atol=2e-6), k
    # training mode: same values, gradients reach both models
    mc, mf = model_from(A, sd_c, ins_num, True), model_from(A, sd_f, ins_num, True)
    targs = types.SimpleNamespace(perturb=perturb, N_importance=0, is_train=True, N_ins=None)
    out = A.R.dm_nerf(rays.cuda(), None, None, mc, mf, z.cuda(), targs, t_rand=None if t_rand is None else t_rand.cuda())
    assert torch.allclose(cpu(out['rgb_fine']), want['rgb_fine'], rtol=2e-6, atol=2e-6)
    (out['rgb_fine'].sum() + out['rgb_coarse'].sum()).backward()
    assert float(mc.mlps[0].weight.grad.abs().max()) > 0 and float(mf.mlps[0].weight.grad.abs().max()) > 0


def test_model_forward_on_embedded_rows_is_differentiable(A):
    """``DM_NeRF.forward`` called directly on [M, 90] rows in training mode (dm_nerf.py:80-106): values equal the
    inference kernel bit for bit, parameter gradients equal PyTorch autograd of the oracle; leading dims preserved."""
    for ins_num, seed, M_rows in ((13, 7, 200), (59, 8, 45)):
        sd = O.make_weights(seed, ins_num, gain=1.7)
        g = torch.Generator().manual_seed(seed)
        pts = (torch.rand(M_rows, 3, generator=g) * 2 - 1) * 6.0
        dirs = torch.nn.functional.normalize(torch.randn(M_rows, 3, generator=g), dim=-1)
        x = torch.cat([O.embed(pts, 10), O.embed(dirs, 4)], -1)
        cot = torch.randn(M_rows, 4 + ins_num + 1, generator=g)
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        want = O.mlp_forward(sdg, x)
        (want * cot).sum().backward()
        m = model_from(A, sd, ins_num, train=True)
        y = m(x.cuda())
        assert y.requires_grad and y.shape == want.shape
        assert maxrel(cpu(y), want.detach()) <= 1e-5
        (y * cot.cuda()).sum().backward()
        for k, p in m.named_parameters():
            assert p.grad is not None, k
            err = float((p.grad.cpu().double() - sdg[k].grad.double()).abs().max())
            assert err <= 2e-4 * float(sdg[k].grad.abs().max()) + 1e-7, (k, err)
        with torch.no_grad():
            assert torch.equal(m(x.cuda()), y.detach())                     # inference kernel == training forward
        assert m(x.cuda().reshape(5, M_rows // 5, 90)).shape == (5, M_rows // 5, 4 + ins_num + 1)
    with pytest.raises(NotImplementedError):                                 # no gradient for the embedded input itself
        m(x.cuda().requires_grad_(True))


def test_frozen_fine_model_still_trains_the_coarse_one(A):
    """Training is decided from the parameters of BOTH models: with model_fine frozen, a loss on the coarse level must
    still produce gradients for model_coarse (it used to take the non-differentiable path)."""
    ins_num, N = 13, 24
    mc = model_from(A, O.make_weights(87, ins_num, gain=1.7, sigma_bias=0.3), ins_num, True)
    mf = model_from(A, O.make_weights(88, ins_num, gain=1.7, sigma_bias=0.3), ins_num, True)
    for p in mf.parameters():
        p.requires_grad_(False)
    args = types.SimpleNamespace(perturb=0.0, N_importance=128, is_train=True, N_ins=None)
    out = A.R.dm_nerf(_rays(N, 87).cuda(), None, None, mc, mf, A.H.z_val_sample(N, 4.0, 15.0, 64), args)
    assert out['rgb_coarse'].requires_grad
    (out['rgb_coarse'].sum() + out['rgb_fine'].sum()).backward()
    assert float(mc.mlps[3].weight.grad.abs().max()) > 0
    assert all(p.grad is None for p in mf.parameters())


def test_wrong_shaped_random_draws_are_refused(A):
    """Caller-supplied t_rand / u reach the kernels as raw pointers: a wrong shape must raise, not read out of bounds."""
    ins_num, N = 13, 16
    mc, mf = model_from(A, O.make_weights(1, ins_num), ins_num), model_from(A, O.make_weights(2, ins_num), ins_num)
    rays, z = _rays(N, 3).cuda(), A.H.z_val_sample(N, 4.0, 15.0, 64)
    jit = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=False, N_ins=None)
    det = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    with torch.no_grad():
        with pytest.raises(ValueError, match="t_rand"):
            A.R.dm_nerf(rays, None, None, mc, mf, z, jit, t_rand=torch.rand(N, 32, device="cuda"))
        with pytest.raises(ValueError, match="u must"):
            A.R.dm_nerf(rays, None, None, mc, mf, z, jit, u=torch.rand(N, 64, device="cuda"))
        with pytest.raises(ValueError, match="u must"):
            A.R.dm_nerf(rays, None, None, mc, mf, z, det, u=torch.rand(N - 1, 128, device="cuda"))
        with pytest.raises(ValueError, match="u must"):
            A.H.importance_resample(z, torch.rand(N, 64, device="cuda"), 128, u=torch.rand(64, device="cuda"))
        with pytest.raises(ValueError):
            A.H.stratify(z, torch.rand(N, 63, device="cuda"))
        # the legal forms: a 1-D grid shared by all rays (det), a full [N, n_imp] draw (with jitter)
        a = A.R.dm_nerf(rays, None, None, mc, mf, z, det, u=torch.linspace(0., 1., 128).cuda())
        b = A.R.dm_nerf(rays, None, None, mc, mf, z, det)
        assert torch.equal(a['z_vals_fine'], b['z_vals_fine'])
        # a 1-D u together with jitter: accepted, stride 0 (every ray uses the same grid)
        c = A.R.dm_nerf(rays, None, None, mc, mf, z, jit, t_rand=torch.rand(N, 64, device="cuda"), u=torch.linspace(0., 1., 128).cuda())
        assert bool(torch.isfinite(c['rgb_fine']).all())
    mc.train(); mf.train()
    with pytest.raises(ValueError, match="u must"):                          # the training path validates too
        A.R.dm_nerf(rays, None, None, mc, mf, z, jit, u=torch.rand(N, 64, device="cuda"))


def test_invalidate_blobs_after_a_data_update(A):
    """An in-place op on the parameter bumps ``_version`` and refreshes the packed weights by itself; an update THROUGH
    ``.data`` does not (no version bump) -- ``invalidate_blobs()`` is the documented way to make it visible."""
    m = model_from(A, O.make_weights(4, 13, gain=1.7), 13)
    x = torch.randn(64, 90).cuda()
    with torch.no_grad():
        y0 = m(x)
        m.density_linear.bias.data.add_(1.0)
        y_stale = m(x)
        m.invalidate_blobs()
        y1 = m(x)
    assert torch.equal(y_stale, y0)                                           # the documented pitfall
    assert torch.allclose(y1[:, 3], y0[:, 3] + 1.0, atol=1e-5) and torch.equal(y1[:, :3], y0[:, :3])


def test_manipulation_with_the_maximum_number_of_moved_objects_composites(A):
    """T = 8 moved objects (the exchanger's limit): the final render composites 64 + 128 + 8 x 128 = 1216 samples per ray;
    the per-ray staging of the compositing kernels covers it (it was 1024: T = 7 was accepted by the exchanger and then
    refused by the render).  Compositing at S = 1216 against the oracle; S beyond the limit still fails loudly."""
    g = torch.Generator().manual_seed(12)
    N, S, C = 5, 1216, 8
    raw = torch.randn(N, S, 4 + C, generator=g)
    raw[..., 3] = raw[..., 3] * 2 + 0.2
    z = torch.sort(torch.rand(N, S, generator=g) * 11 + 4, -1)[0]
    d = torch.randn(N, 3, generator=g)
    want = O.manipulator_render(raw, z, d)
    got = A.MA.manipulator_render(raw.cuda(), z.cuda(), d.cuda())
    for a_, b_ in zip(got, want):
        assert torch.allclose(cpu(a_), b_, rtol=2e-6, atol=2e-6)
    want = O.render_train(raw, z, d)
    got = A.R.render_train(raw.cuda(), z.cuda(), d.cuda())
    for a_, b_ in zip(got, want):
        assert torch.allclose(cpu(a_), b_, rtol=2e-6, atol=2e-6)
    r1 = raw.cuda().requires_grad_(True)
    o = A.R.render_train(r1, z.cuda(), d.cuda())
    gr, = torch.autograd.grad(o[0].sum() + o[3].sum(), r1)
    r0 = raw.clone().requires_grad_(True)
    o = O.render_train(r0, z, d)
    gw, = torch.autograd.grad(o[0].sum() + o[3].sum(), r0)
    assert float((cpu(gr) - gw).abs().max()) <= 2e-4 * float(gw.abs().max()) + 1e-7
    with pytest.raises(RuntimeError):
        A.R.render_train(torch.zeros(2, 1300, 8, device="cuda"), torch.zeros(2, 1300, device="cuda"), torch.ones(2, 3, device="cuda"))
    # the exchanger itself with T = 8 target sets
    T = 8
    ori = torch.randn(N, 40, 4 + C, generator=g).cuda()
    tars = [torch.randn(N, 40, 4 + C, generator=g).cuda() for _ in range(T)]
    accs = [torch.rand(N, C, generator=g).cuda() for _ in range(T)]
    ori_acc = torch.rand(N, C, generator=g).cuda()
    want = O.exchanger(ori.cpu().clone(), [t.cpu().clone() for t in tars], ori_acc.cpu(), [t.cpu() for t in accs], list(range(T)))
    out = A.MA.exchanger(ori, tars, ori_acc, accs, list(range(T)))
    assert torch.equal(cpu(out[0]), want[0]) and torch.equal(cpu(out[2]), want[2])
    with pytest.raises(RuntimeError):
        A.MA.exchanger(ori, tars + tars[:1], ori_acc, accs + accs[:1], list(range(T + 1)))


def test_error_messages_are_per_thread(A):
    """``dmnerf_last_error()`` is thread-local: two host threads failing with different messages each read their own."""
    lib = A.lib.load()
    seen = {}

    def fail(tag, S):
        rc = lib.dmnerf_composite_fwd(None, None, None, 4, S, 14, None, None, None, None, None)
        for _ in range(200):                                                  # give the other thread every chance to interfere
            lib.dmnerf_composite_fwd(None, None, None, 4, S, 14, None, None, None, None, None)
        seen[tag] = (rc, A.lib.last_error())

    ts = [threading.Thread(target=fail, args=("a", 5000)), threading.Thread(target=fail, args=("b", 64))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert seen["a"][0] == -1 and "S=5000" in seen["a"][1]
    assert seen["b"][0] == -1 and "null" in seen["b"][1]


def test_device_head_fusion_equals_the_float64_formula(A):
    """``dmnerf_fuse_heads`` (float64 accumulation on the device, rounded once) against the readable torch statement of the
    same folding (weights.fuse_heads: float64 matmul): equal to the last bit except where the two f64 summation orders
    round the final f32 differently (<= 1 ulp); the 27 direction columns and every other parameter are copied verbatim."""
    from dm_nerf_amd import weights as Wt
    for ins_num, seed in ((13, 21), (93, 22)):
        sd = {k: v.cuda() for k, v in O.make_weights(seed, ins_num, gain=1.7).items()}
        flat = Wt.flat_params(sd)
        got = Wt.fused_flat(flat, ins_num)
        want = Wt.flat_params(O.fuse_heads(sd))
        assert got.shape == want.shape
        changed = (got != flat)
        assert int(changed.sum()) <= 2 * (128 * 256 + 128)                   # only the two hidden layers' fused blocks
        ulp = (got - want).abs() / (want.abs().clamp(min=1e-30) * 2 ** -23)
        assert float(ulp.max()) <= 1.0 and float((got != want).float().mean()) <= 1e-3
        # the dirs columns of rgb_feature_linears.0 are untouched
        off = sum(v.numel() for k, v in sd.items() if k.split(".")[0] in ("mlps", "rgb_feature_linear", "ins_feature_linear") and not k.startswith("rgb_feature_linears"))
        w_rh = got[off:off + 128 * 283].reshape(128, 283)
        assert torch.equal(w_rh[:, 256:], sd["rgb_feature_linears.0.weight"][:, 256:])


def test_training_batches_beyond_the_launch_limit_run_as_several_launches(monkeypatch):
    """A training launch addresses its workspace with 32-bit byte offsets (DMNERF_MAX_TRAIN_SAMPLES): larger batches are cut
    into several launches whose parameter gradients autograd adds up (autograd.run_network_train / mlp_forward_train).  With
    the limit lowered to 4096 samples a 100-ray x 64-sample batch runs as two launches: same raw (bit for bit: rays are
    independent), gradients equal to the single launch's up to the summation order; an over-long direct C-ABI call is refused."""
    import types
    from dm_nerf_amd import _lib, autograd as G
    from dm_nerf_amd.networks import dm_nerf as M
    lib = _lib.load()
    torch.manual_seed(3)
    ins_num, N, S = 13, 100, 64
    m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(4)
    ro, rd = torch.randn(N, 3, device="cuda", generator=g), torch.randn(N, 3, device="cuda", generator=g)
    z = torch.sort(torch.rand(N, S, device="cuda", generator=g) * 5 + 1, -1)[0]
    cot = torch.randn(N, S, 4 + ins_num + 1, device="cuda", generator=g)

    def run():
        for p in m.parameters():
            p.grad = None
        raw = G.run_network_train(m, ro, rd, z)
        (raw * cot).sum().backward()
        return raw.detach().clone(), [p.grad.clone() for p in m.parameters()]

    raw1, g1 = run()
    monkeypatch.setattr(G, "MAX_TRAIN_SAMPLES", 4096)
    raw2, g2 = run()
    assert torch.equal(raw1, raw2)
    for (k, _), a, b in zip(m.named_parameters(), g1, g2):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-9, k
    # pre-embedded rows: the same cut
    x = torch.randn(6000, 90, device="cuda", generator=g)
    for p in m.parameters():
        p.grad = None
    out2 = m(x); out2.square().sum().backward()
    gx2 = [p.grad.clone() for p in m.parameters()]
    monkeypatch.setattr(G, "MAX_TRAIN_SAMPLES", 1048576)
    for p in m.parameters():
        p.grad = None
    out1 = m(x); out1.square().sum().backward()
    assert torch.equal(out1.detach(), out2.detach())
    for (k, _), a, b in zip(m.named_parameters(), [p.grad for p in m.parameters()], gx2):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-9, k
    # the C ABI itself refuses a launch beyond the limit (loudly, before anything is launched)
    n_big = 1048576 // 64 + 1                                   # 16385 rays x 64 samples = the limit + 64
    rob, rdb = torch.randn(n_big, 3, device="cuda"), torch.randn(n_big, 3, device="cuda")
    zb = torch.sort(torch.rand(n_big, 64, device="cuda") * 5 + 1, -1)[0]
    rawb = torch.empty(n_big, 64, 4 + ins_num + 1, device="cuda")
    rc = lib.dmnerf_mlp_fwd_rays_train(_lib.ptr(m.blob()), ins_num, _lib.ptr(rob), _lib.ptr(rdb), _lib.ptr(zb), n_big, 64,
                                       _lib.ptr(rawb), _lib.ptr(rawb), _lib.stream())
    assert rc != 0 and b"exceed" in lib.dmnerf_last_error(), (rc, lib.dmnerf_last_error())
    torch.cuda.synchronize()


def test_more_objects_than_the_abi_supports_is_refused_loudly():
    """C = ins_num + 1 <= DMNERF_MAX_LOGITS = 128 (include/dmnerf_hip.h; Replica room_0 has 94): ins_num = 128 constructs (it is
    an ordinary nn.Module) but every path into a kernel raises instead of truncating the object-code head."""
    from dm_nerf_amd.networks import dm_nerf as M, render as R
    m = M.DM_NeRF(8, 256, 63, 27, [4], 128).cuda()
    with pytest.raises((ValueError, RuntimeError)):
        m.blob()
    ro, rd = torch.randn(4, 3, device="cuda"), torch.randn(4, 3, device="cuda")
    z = torch.sort(torch.rand(4, 8, device="cuda") + 1, -1)[0]
    with pytest.raises((ValueError, RuntimeError)), torch.no_grad():
        R.run_network(m, ro, rd, z)
    with pytest.raises((ValueError, RuntimeError)):
        m.train()(torch.randn(8, 90, device="cuda"))


def test_empty_training_batch_gives_zero_gradients(A):
    """An empty batch (a rank whose shard of a tiny batch is empty) trains without error: raw [0, S, 4 + C], zero gradients."""
    from dm_nerf_amd import autograd as G
    m = A.M.DM_NeRF(8, 256, 63, 27, [4], 13).cuda().train()
    ro, rd = torch.empty(0, 3, device="cuda"), torch.empty(0, 3, device="cuda")
    z = torch.empty(0, 64, device="cuda")
    raw = G.run_network_train(m, ro, rd, z)
    assert raw.shape == (0, 64, 18)
    raw.sum().backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and float(p.grad.abs().max()) == 0.0, k


def test_f16x2_range_of_the_two_plane_split(A, capsys):
    """The f16x2 mode splits every layer INPUT into two f16 planes: 22 significand bits as long as the activation is inside the f16
    range.  Activations in the thousands (weights 4.5x nn.Linear's scale: trunk to ~10^3, the heads' hidden layers to ~10^4; the
    reference's trained scenes sit below 100): still the f32-class tolerance against the oracle.  Beyond 65 504 (8x: the trunk alone
    reaches ~84 000) the planes SATURATE -- v_cvt_pkrtz_f16_f32
    rounds towards zero, never to infinity -- so the result stays finite but is no longer f32-class, which INTEGRATION.md states; the
    default kernels keep the f32 range and their tolerance on the same weights."""
    g = torch.Generator().manual_seed(1)
    N, S, ins_num = 8, 8, 13
    ro, rd = torch.randn(N, 3, generator=g) * 0.5, torch.randn(N, 3, generator=g)
    z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0]
    pts = ro[:, None, :] + rd[:, None, :] * z[:, :, None]
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    x = torch.cat([O.embed(pts.reshape(-1, 3), 10), O.embed(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
    rep = {}
    for gain in (4.5, 8.0):
        sd = O.make_weights(21, ins_num, gain=gain)
        want = O.mlp_forward(sd, x).reshape(N, S, -1)
        m = model_from(A, sd, ins_num)
        with torch.no_grad():
            got32 = cpu(A.R.run_network(m, ro.cuda(), rd.cuda(), z.cuda()))
            got16 = cpu(A.R.run_network(m, ro.cuda(), rd.cuda(), z.cuda(), split="f16x2"))
        scale = float(want.abs().max())
        rep[gain] = dict(max_raw=scale, f32=float((got32 - want).abs().max()) / scale, f16x2=float((got16 - want).abs().max()) / scale)
        assert bool(torch.isfinite(got16).all()) and bool(torch.isfinite(got32).all()), rep
        assert rep[gain]["f32"] <= 1e-5, rep
    with capsys.disabled():
        print("\n[f16x2 range] max |d raw| / max |raw| vs the oracle: " + str(rep))
    assert rep[4.5]["f16x2"] <= 1e-5, rep                  # inside the f16 range: f32 class
    assert rep[8.0]["f16x2"] > 1e-4, rep                   # beyond it: saturated planes (finite, documented, not f32 class)


def test_f16x2_saturation_is_reported_when_the_check_is_on(A, monkeypatch):
    """Run-time honesty of the opt-in mode: with DMNERF_CHECK_F16=1 (or args.check_f16) a render whose activations leave the f16
    range WARNS (sticky device flags word, read at the check), the same render inside the range does not, and a training step
    reports through ``autograd.check_f16x2()``.  With the check off nothing is launched and nothing is said (the default)."""
    import types
    import warnings
    from dm_nerf_amd import autograd as G
    from dm_nerf_amd.networks import render as R
    ins_num, N = 13, 48
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(20.0, -65.0, 7.0))
    rays = torch.stack([ro.reshape(-1, 3)[5000:5000 + N], rd.reshape(-1, 3)[5000:5000 + N]]).cuda()
    z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous().cuda()
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None, mfma_split="f16x2", check_f16=True)
    for gain, expect in ((1.7, False), (8.0, True)):
        mc, mf = model_from(A, O.make_weights(21, ins_num, gain=gain), ins_num), model_from(A, O.make_weights(22, ins_num, gain=gain), ins_num)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.no_grad():
                out = R.dm_nerf(rays, None, None, mc, mf, z, args)
        said = [x for x in w if issubclass(x.category, RuntimeWarning) and "65504" in str(x.message)]
        assert bool(said) == expect, (gain, [str(x.message) for x in w])
        assert bool(torch.isfinite(out['rgb_fine']).all())
        assert G.check_f16x2(warn=False) == 0                                  # the read cleared the word
    # check off (the default): silent, and no flag is ever set
    args.check_f16 = None
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            R.dm_nerf(rays, None, None, mc, mf, z, args)
    assert not [x for x in w if "65504" in str(x.message)] and G.check_f16x2(warn=False) == 0
    # training: the forward's saved activations (and the backward's scaled gradients) are scanned on the stream; the caller reads
    monkeypatch.setattr(G, "CHECK_F16", True)
    mc.train(); mf.train()
    targs = types.SimpleNamespace(perturb=0.0, N_importance=128, is_train=True, N_ins=None, mfma_split="f16x2")
    o = R.dm_nerf(rays, None, None, mc, mf, z, targs)
    (o['rgb_fine'].sum() + o['rgb_coarse'].sum()).backward()
    with pytest.warns(RuntimeWarning, match="65504"):
        flags = G.check_f16x2()
    assert flags & G.F16_ACT_SATURATED
    assert G.check_f16x2(warn=False) == 0
    small_c, small_f = model_from(A, O.make_weights(21, ins_num, gain=1.7), ins_num).train(), model_from(A, O.make_weights(22, ins_num, gain=1.7), ins_num).train()
    o = R.dm_nerf(rays, None, None, small_c, small_f, z, targs)
    (o['rgb_fine'].sum() + o['rgb_coarse'].sum()).backward()
    assert G.check_f16x2(warn=False) == 0                                      # inside the range (ragged M too): no false alarm


def test_overlapped_backward_with_three_launches_per_model_sends_only_the_first_to_the_side_stream(monkeypatch):
    """a model whose batch runs as THREE chunk launches inside ``overlapped_backward`` -- F3 on the side
    stream, F2 on the main stream (joins), and F1, which sees ``p.grad is None`` and nothing pending again, must NOT take the side
    stream (the engine would add its output to F3 + F2 on the main stream unsynchronised).  With the launch limit lowered to
    4096 samples a 150-ray x 64-sample batch is 3 launches per model; two models as in a training step.  The gradients equal the
    one-stream pass bit for bit, over several repetitions, and the side stream is taken exactly once per model-pass that may."""
    from dm_nerf_amd import autograd as G
    from dm_nerf_amd.networks import dm_nerf as M
    torch.manual_seed(5)
    ins_num, N, S = 13, 150, 64
    ms = [M.DM_NeRF(8, 256, 63, 27, [4], ins_num).cuda().train() for _ in range(2)]
    g = torch.Generator(device="cuda").manual_seed(6)
    ro, rd = torch.randn(N, 3, device="cuda", generator=g), torch.randn(N, 3, device="cuda", generator=g)
    z = torch.sort(torch.rand(N, S, device="cuda", generator=g) * 5 + 1, -1)[0]
    cot = torch.randn(N, S, 4 + ins_num + 1, device="cuda", generator=g)
    monkeypatch.setattr(G, "MAX_TRAIN_SAMPLES", 4096)
    side_calls = []
    real = G._mlp_backward_on_stream

    def spy(ctx, g_raw):
        side_calls.append(torch.cuda.current_stream() != torch.cuda.default_stream())
        return real(ctx, g_raw)
    monkeypatch.setattr(G, "_mlp_backward_on_stream", spy)

    def run(overlap):
        for m in ms:
            for p in m.parameters():
                p.grad = None
        loss = sum((G.run_network_train(m, ro, rd, z) * cot).sum() for m in ms)
        with G.overlapped_backward(overlap):
            loss.backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for m in ms for p in m.parameters()]

    want = run(False)
    assert side_calls == [False] * 6
    for _ in range(5):
        del side_calls[:]
        got = run(True)
        assert len(side_calls) == 6 and sum(side_calls) == 2, side_calls           # one side-stream launch per model, never its 2nd / 3rd
        assert side_calls[0] and not side_calls[1] and not side_calls[2]
        for a, b in zip(got, want):
            assert torch.equal(a, b)


def test_f16x2_check_via_args_reaches_training_and_the_probe_runs_in_pieces(A, monkeypatch):
    """(1) ``args.check_f16`` alone -- no env var, no module flag -- switches the TRAINING-side scans on (forward's
    saved activations, backward's scaled gradients); (2) the inference probe runs in pieces of at most PROBE_SAMPLES samples with one
    reused workspace, so an inference chunk larger than the training launch limit is probed instead of raising."""
    import types
    import warnings
    from dm_nerf_amd import autograd as G
    from dm_nerf_amd.networks import render as R
    ins_num, N = 13, 48
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(20.0, -65.0, 7.0))
    rays = torch.stack([ro.reshape(-1, 3)[5000:5000 + N], rd.reshape(-1, 3)[5000:5000 + N]]).cuda()
    z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous().cuda()
    assert G.CHECK_F16 is None and not G.f16_check_enabled()
    big_c, big_f = model_from(A, O.make_weights(21, ins_num, gain=8.0), ins_num).train(), model_from(A, O.make_weights(22, ins_num, gain=8.0), ins_num).train()
    for chk, expect in ((True, True), (None, False)):
        targs = types.SimpleNamespace(perturb=0.0, N_importance=128, is_train=True, N_ins=None, mfma_split="f16x2", check_f16=chk)
        o = R.dm_nerf(rays, None, None, big_c, big_f, z, targs)
        (o['rgb_fine'].sum() + o['rgb_coarse'].sum()).backward()
        assert bool(G.check_f16x2(warn=False) & G.F16_ACT_SATURATED) == expect
    # the probe in pieces: 48 rays x 192 samples with room for 1000 samples per piece -> 5 rays per launch, ragged last piece
    monkeypatch.setattr(G, "PROBE_SAMPLES", 1000)
    calls = []
    real = G._f16_range_scan
    monkeypatch.setattr(G, "_f16_range_scan", lambda ws, M, g: (calls.append(M), real(ws, M, g))[1])
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None, mfma_split="f16x2", check_f16=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            R.dm_nerf(rays, None, None, big_c.eval(), big_f.eval(), z, args)
    assert [x for x in w if "65504" in str(x.message)]
    assert calls == [15 * 64] * 3 + [3 * 64] + [5 * 192] * 9 + [3 * 192], calls
