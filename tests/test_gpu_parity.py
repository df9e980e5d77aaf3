"""GPU parity tests (-m gpu): every HIP stage, called through the C ABI via the Python mirror of
the reference API, against (a) the committed golden vectors produced by the reference itself and
(b) the CPU oracle on fresh seeded inputs.  Tolerances are the per-stage contract of SURVEY.md 8(a):
integer / permutation outputs exact; raygen / z grids <= 1 ulp-class; PE abs 2.4e-7 (x2 margin);
MLP |d raw| <= 1e-5 (1 + |raw|); compositing rel 2e-6 (+abs); argmax mismatch <= 1e-3 of rays.
"""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from dm_nerf_amd import _lib
    from dm_nerf_amd.networks import dm_nerf as M, helpers as H, render as R
    from dm_nerf_amd import config as Cfg
    _lib.load()
    return types.SimpleNamespace(M=M, H=H, R=R, Cfg=Cfg, lib=_lib)


def dev(t):
    return t.cuda()


def cpu(t):
    torch.cuda.synchronize()
    return t.detach().cpu()


def maxrel(a, b):
    return float(((a - b).abs() / (1 + b.abs())).max())


def model_from(A, sd, ins_num):
    m = A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
    m.load_state_dict(sd)
    return m.cuda().eval()


# ------------------------------------------------------------------------------------------
def test_embed_golden(A, golden):
    g = golden("embed")
    e10, d10 = A.M.get_embedder(10, 0)
    e4, d4 = A.M.get_embedder(4, 0)
    assert (d10, d4) == (63, 27)
    y10, y4 = cpu(e10.embed(dev(g["x"]))), cpu(e4.embed(dev(g["d"])))
    assert y10.shape == g["y10"].shape
    assert torch.equal(y10[:, :3], g["x"])
    assert float((y10 - g["y10"]).abs().max()) <= 4.8e-7      # args reach 7680 rad
    assert float((y4 - g["y4"]).abs().max()) <= 2.4e-7


def test_mlp_embedded_golden(A, golden):
    g = golden("mlp")
    for ins_num in (13, 59, 93):
        sd = O.make_weights(int(g[f"seed_{ins_num}"]), ins_num, gain=float(g["gain"]))
        m = model_from(A, sd, ins_num)
        with torch.no_grad():
            y = cpu(m(dev(g[f"x_{ins_num}"])))
        want = g[f"y_{ins_num}"]
        assert y.shape == want.shape
        assert maxrel(y, want) <= 1e-5, (ins_num, maxrel(y, want))


def test_mlp_embedded_vs_oracle_ragged_sizes(A):
    sd = O.make_weights(3, 13, gain=1.7)
    m = model_from(A, sd, 13)
    g = torch.Generator().manual_seed(3)
    for M_rows in (1, 31, 32, 33, 127, 1000):
        pts = (torch.rand(M_rows, 3, generator=g) * 2 - 1) * 8
        dirs = torch.nn.functional.normalize(torch.randn(M_rows, 3, generator=g), dim=-1)
        x = torch.cat([O.embed(pts, 10), O.embed(dirs, 4)], -1)
        with torch.no_grad():
            y = cpu(m(dev(x)))
        want = O.mlp_forward(sd, x)
        assert y.shape == want.shape and maxrel(y, want) <= 1e-5, (M_rows, maxrel(y, want))
    # leading dims are preserved and an empty batch is fine
    with torch.no_grad():
        assert m(dev(x).reshape(10, 100, 90)).shape == (10, 100, 18)
        assert m(torch.empty(0, 90, device="cuda")).shape == (0, 18)


def test_blob_refreshes_after_inplace_update(A):
    sd = O.make_weights(4, 13, gain=1.7)
    m = model_from(A, sd, 13)
    x = torch.randn(64, 90)
    with torch.no_grad():
        y0 = cpu(m(dev(x)))
        m.density_linear.bias.add_(1.0)             # what optimizer.step() does: in-place, bumps _version
        y1 = cpu(m(dev(x)))
    assert torch.allclose(y1[:, 3], y0[:, 3] + 1.0, atol=1e-5) and torch.equal(y1[:, :3], y0[:, :3])


def test_render_train_golden(A, golden):
    g = golden("render_train")
    for k in ("S64_C14", "S192_C14", "S320_C60", "S192_C94", "S5_C3", "kat"):
        raw, z, d = g[f"{k}_raw"], g[f"{k}_z"], g[f"{k}_d"]
        with torch.no_grad():
            rgb, w, dep, ins = [cpu(t) for t in A.R.render_train(dev(raw), dev(z), dev(d))]
        for got, name in ((rgb, "rgb"), (w, "w"), (dep, "depth"), (ins, "ins")):
            want = g[f"{k}_{name}"]
            assert got.shape == want.shape, (k, name)
            assert torch.allclose(got, want, rtol=2e-6, atol=2e-6 * max(1.0, float(want.abs().max()))), \
                (k, name, float((got - want).abs().max()))
        # object argmax: bit-exact.  (i) given the IDENTICAL ins_map (the golden one through the device argmax kernel);
        # (ii) of the composited map itself -- it agrees with the golden map to 2e-6 and the fixture has no closer ties
        want_ins = g[f"{k}_ins"]
        from dm_nerf_amd.networks import evaluator as E
        label, conf = E.ins_label_conf(dev(want_ins))
        assert torch.equal(cpu(label), want_ins.argmax(-1)) and torch.equal(cpu(conf), want_ins.max(-1).values)
        assert torch.equal(ins.argmax(-1), want_ins.argmax(-1)), k


def test_sample_pdf_golden(A, golden):
    g = golden("sample_pdf")
    bins, w = dev(g["bins"]), dev(g["w"])
    # stage-isolated: identical (cdf, u) -> identical indices (exact) and samples
    s, inds = A.H.sample_from_cdf(bins, dev(g["cdf"]), dev(g["u_rnd"]))
    assert torch.equal(cpu(inds), g["inds_rnd"])
    assert torch.allclose(cpu(s), g["s_rnd"], rtol=1e-6, atol=1e-6)
    s, inds = A.H.sample_from_cdf(bins, dev(g["cdf"]), dev(g["u_det"]))
    assert torch.equal(cpu(inds), g["inds_det"])
    assert torch.allclose(cpu(s), g["s_det"], rtol=1e-6, atol=1e-6)
    # full sample_pdf: cdf within 1-2 ulp, indices >= 99.9 % equal, samples close where indices agree
    for u_key, s_key, i_key in (("u_det", "s_det", "inds_det"), ("u_rnd", "s_rnd", "inds_rnd")):
        s, cdf, inds = A.H.sample_pdf(bins, w, 128, u=dev(g[u_key]), return_aux=True)
        s, cdf, inds = cpu(s), cpu(cdf), cpu(inds)
        assert float((cdf - g["cdf"]).abs().max()) <= 2.4e-7
        same = inds == g[i_key]
        assert same.float().mean() >= 0.999
        assert torch.allclose(s[same], g[s_key][same], rtol=1e-5, atol=1e-5)
    # det=True path builds its own linspace
    s = cpu(A.H.sample_pdf(bins, w, 128, det=True))
    assert torch.allclose(s, g["s_det"], rtol=1e-5, atol=2e-4)


def test_importance_resample_golden(A, golden):
    g = golden("sample_pdf")
    N = g["z"].shape[0]
    # weights_coarse[..., 1:-1] are the pdf weights (render.py:67)
    wc = torch.zeros(N, 64); wc[:, 1:-1] = g["w"]
    zf, zs = A.H.importance_resample(dev(g["z"]), dev(wc), 128, u=dev(g["u_det"]), return_samples=True)
    zf, zs = cpu(zf), cpu(zs)
    assert zf.shape == (N, 192)
    assert torch.allclose(zs, g["s_det"], rtol=1e-5, atol=2e-4)
    # exact permutation property: z_fine is the sorted multiset of coarse + the kernel's own samples
    assert torch.equal(zf, torch.sort(torch.cat([g["z"], zs], -1), -1)[0])
    assert torch.allclose(zf, g["zf_det"], rtol=1e-5, atol=2e-4)
    zf2, zs2 = A.H.importance_resample(dev(g["z"]), dev(wc), 128, u=dev(g["u_rnd"]), return_samples=True)
    assert torch.equal(cpu(zf2), torch.sort(torch.cat([g["z"], cpu(zs2)], -1), -1)[0])


def test_rays_and_zvals_golden(A, golden):
    g = golden("rays")
    H_, W_ = [int(v) for v in g["HW"]]
    c2w = g["c2w"]
    for name in ("dmsr", "replica", "scannet"):
        o, d = A.H.get_rays_k(H_, W_, g[f"K_{name}"].numpy(), dev(c2w))
        assert o.shape == (H_, W_, 3)
        assert torch.equal(cpu(o), g[f"o_{name}"])
        assert torch.allclose(cpu(d), g[f"d_{name}"], rtol=3e-7, atol=1e-7)
    o, d = A.H.get_rays_k(480, 640, g["K_full"].numpy(), dev(c2w))
    d = cpu(d)
    assert torch.allclose(d[0], g["d_full_row0"], rtol=3e-7, atol=1e-7)
    assert torch.allclose(d[479], g["d_full_row479"], rtol=3e-7, atol=1e-7)
    # a row band equals the same rows of the full frame (how ranks shard a frame)
    ob, db = A.H.get_rays_k(480, 640, g["K_full"].numpy(), dev(c2w), row0=120, nrows=60)
    assert torch.equal(cpu(db), d[120:180])
    z = cpu(A.H.z_val_sample(3, 4.0, 15.0, 64))
    assert torch.equal(z, g["z_4_15"])
    assert torch.equal(cpu(A.H.z_val_sample(2, 0.0, 4.7, 64)), g["z_0_47"])
    assert torch.equal(cpu(A.H.stratify(dev(g["z_4_15"]), dev(g["t_rand"]))), g["z_jit"])


def _check_dict(out, g, prefix):
    worst = {}
    for k in ('rgb_fine', 'ins_fine', 'z_vals_fine', 'raw_fine', 'raw_coarse', 'rgb_coarse', 'ins_coarse',
              'z_vals_coarse', 'depth_fine', 'depth_coarse'):
        got, want = cpu(out[k]), g[f"{prefix}_{k}"]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        worst[k] = maxrel(got, want)
    return worst


def _check_levels(A, out, g, prefix, mf, rays):
    """Coarse level strictly; the fine level stage by stage.  The inverse-CDF step divides by the
    local cdf slope (as small as 1e-5, helpers.py:151), so float noise in the coarse weights moves a
    few fine depths by ~1e-3, and 2^9-frequency encodings amplify that in raw_fine: end-to-end the
    fine tensors are compared loosely, and each fine stage is pinned on the GOLDEN inputs instead."""
    worst = _check_dict(out, g, prefix)
    assert worst['z_vals_coarse'] == 0.0
    assert worst['raw_coarse'] <= 1e-5 and worst['rgb_coarse'] <= 2e-6 and worst['ins_coarse'] <= 2e-6, worst
    assert worst['depth_coarse'] <= 2e-6, worst
    dz = (cpu(out['z_vals_fine']) - g[f"{prefix}_z_vals_fine"]).abs()
    assert float((dz <= 1e-4).float().mean()) >= 0.99 and float(dz.max()) <= 2e-2, (float(dz.max()),)
    assert worst['rgb_fine'] <= 1e-3 and worst['depth_fine'] <= 1e-3 and worst['ins_fine'] <= 1e-3, worst
    # fine network on the golden depths: f32-roundoff class
    zf = dev(g[f"{prefix}_z_vals_fine"])
    with torch.no_grad():
        raw_f = A.R.run_network(mf, rays[0], rays[1], zf)
    assert maxrel(cpu(raw_f), g[f"{prefix}_raw_fine"]) <= 1e-5
    # fine compositing on the golden raw
    with torch.no_grad():
        rgb, w, dep, ins = A.R.render_train(dev(g[f"{prefix}_raw_fine"]), zf, rays[1])
    assert torch.allclose(cpu(rgb), g[f"{prefix}_rgb_fine"], rtol=2e-6, atol=2e-6)
    assert torch.allclose(cpu(dep), g[f"{prefix}_depth_fine"], rtol=2e-6, atol=2e-5)
    n_ins = g[f"{prefix}_ins_fine"].shape[0]
    assert torch.allclose(cpu(ins)[-n_ins:], g[f"{prefix}_ins_fine"], rtol=2e-6, atol=2e-6)
    assert torch.equal(cpu(ins)[-n_ins:].argmax(-1), g[f"{prefix}_ins_fine"].argmax(-1))      # argmax exact given identical inputs
    return worst


def test_dm_nerf_dict_golden(A, golden):
    g = golden("dm_nerf")
    ins_num = int(g["ins_num"])
    kw = dict(gain=float(g["gain"]), sigma_bias=float(g["sigma_bias"]))
    mc = model_from(A, O.make_weights(int(g["seed_c"]), ins_num, **kw), ins_num)
    mf = model_from(A, O.make_weights(int(g["seed_f"]), ins_num, **kw), ins_num)
    pe, _ = A.M.get_embedder(10, 0); ve, _ = A.M.get_embedder(4, 0)
    rays = dev(g["rays"])
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    with torch.no_grad():
        out = A.R.dm_nerf(rays, pe, ve, mc, mf, dev(g["z_in"]), args)
    _check_levels(A, out, g, "det", mf, rays)
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=7)
    with torch.no_grad():
        out = A.R.dm_nerf(rays, pe, ve, mc, mf, dev(g["z_in"]), args, t_rand=dev(g["t_rand"]), u=dev(g["u"]))
    assert out['ins_fine'].shape == (7, 13) and out['ins_coarse'].shape == (7, 13)
    _check_levels(A, out, g, "prt", mf, rays)


def test_dm_nerf_vs_oracle_1024_rays(A):
    """BASELINE config 0 shape (1024 rays, 64+128) on the synthetic 640x480 DM-SR camera."""
    ins_num = 13
    sd_c = O.make_weights(11, ins_num, gain=1.7, sigma_bias=0.3)
    sd_f = O.make_weights(12, ins_num, gain=1.7, sigma_bias=0.3)
    mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
    K = O.dmsr_intrinsics(480, 640)
    c2w = O.pose_spherical(20.0, -65.0, 7.0)
    ro, rd = O.get_rays_k(480, 640, K, c2w)
    sel = torch.from_numpy(np.random.RandomState(0).choice(480 * 640, 1024, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    z = O.z_val_sample(1024, 4.0, 15.0, 64).contiguous()
    with torch.no_grad():
        want = O.dm_nerf(rays, sd_c, sd_f, z, perturb=0.)
        args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
        got = A.R.dm_nerf(dev(rays), None, None, mc, mf, dev(z), args)
    got = {k: cpu(v) for k, v in got.items()}
    assert maxrel(got['raw_coarse'], want['raw_coarse']) <= 1e-5
    assert torch.allclose(got['rgb_coarse'], want['rgb_coarse'], rtol=2e-6, atol=2e-6)
    # fine level: sampled depths agree to float noise on >= 99.9 % of the samples
    dz = (got['z_vals_fine'] - want['z_vals_fine']).abs()
    assert float((dz <= 1e-4).float().mean()) >= 0.999
    mse = float(((got['rgb_fine'] - want['rgb_fine']) ** 2).mean())
    psnr = -10 * np.log10(max(mse, 1e-20))
    assert psnr >= 80.0, psnr                                  # north_star: within 0.05 dB of the reference
    flips = float((got['ins_fine'].argmax(-1) != want['ins_fine'].argmax(-1)).float().mean())
    assert flips <= 1e-3 + 1.0 / 1024, flips


@pytest.mark.parametrize("ins_num", [59, 93])
def test_dm_nerf_vs_oracle_wide_object_heads(A, ins_num):
    """BASELINE config 2 (Replica office_0 / room_0: 59 / 93 object codes, two / three 32-row blocks in the ins_linear
    stage of the rays kernel): the whole dm_nerf dict against the oracle on 96 rays."""
    sd_c = O.make_weights(100 + ins_num, ins_num, gain=1.7, sigma_bias=0.3)
    sd_f = O.make_weights(200 + ins_num, ins_num, gain=1.7, sigma_bias=0.3)
    mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(75.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(ins_num).choice(480 * 640, 96, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    z = O.z_val_sample(96, 0.0, 4.7, 64).contiguous()                  # configs/replica/train/office_0.txt:13-14
    with torch.no_grad():
        want = O.dm_nerf(rays, sd_c, sd_f, z, perturb=0.)
        args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
        got = {k: cpu(v) for k, v in A.R.dm_nerf(dev(rays), None, None, mc, mf, dev(z), args).items()}
        # fine network on the oracle's own depths: the inverse-CDF step is compared on golden inputs elsewhere
        raw_f = cpu(A.R.run_network(mf, dev(rays[0]), dev(rays[1]), dev(want['z_vals_fine'])))
    assert got['raw_fine'].shape == (96, 192, 4 + ins_num + 1) and got['ins_fine'].shape == (96, ins_num)
    assert maxrel(got['raw_coarse'], want['raw_coarse']) <= 1e-5
    assert maxrel(raw_f, want['raw_fine']) <= 1e-5
    assert torch.allclose(got['rgb_coarse'], want['rgb_coarse'], rtol=2e-6, atol=2e-6)
    assert torch.allclose(got['ins_coarse'], want['ins_coarse'], rtol=2e-6, atol=2e-6)
    assert torch.equal(got['ins_coarse'].argmax(-1), want['ins_coarse'].argmax(-1))
    mse = float(((got['rgb_fine'] - want['rgb_fine']) ** 2).mean())
    assert -10 * np.log10(max(mse, 1e-20)) >= 80.0


@pytest.mark.parametrize("ins_num,near,far", [(13, 4.0, 15.0), (59, 0.0, 4.7), (93, 0.0, 4.7)])
def test_end_to_end_labels_4096_rays(A, ins_num, near, far, capsys):
    """north_star: "bit-exact for sampled indices / object argmax" given identical inputs, and end to end a label mismatch
    of <= 1e-3 of the rays on the f32 path (SURVEY 8a tolerances).  One full 4096-ray chunk (BASELINE configs 1 / 2) per
    object-code width against the oracle: labels of ins_fine AND ins_coarse, PSNR of rgb_fine, and the measured fraction
    of fine depths that the ill-conditioned inverse-CDF step moved (published in DESIGN.md section 2)."""
    import json
    sd_c = O.make_weights(300 + ins_num, ins_num, gain=1.7, sigma_bias=0.3)
    sd_f = O.make_weights(400 + ins_num, ins_num, gain=1.7, sigma_bias=0.3)
    mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(140.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(1000 + ins_num).choice(480 * 640, 4096, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    z = O.z_val_sample(4096, near, far, 64).contiguous()
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    with torch.no_grad():
        got = {k: cpu(v) for k, v in A.R.dm_nerf(dev(rays), None, None, mc, mf, dev(z), args).items()}
        want = O.dm_nerf(rays, sd_c, sd_f, z, perturb=0.)
    n = 4096
    flips_c = int((got['ins_coarse'].argmax(-1) != want['ins_coarse'].argmax(-1)).sum())
    flips_f = int((got['ins_fine'].argmax(-1) != want['ins_fine'].argmax(-1)).sum())
    dz = (got['z_vals_fine'] - want['z_vals_fine']).abs()
    mse = float(((got['rgb_fine'] - want['rgb_fine']).double() ** 2).mean())
    rep = dict(ins_num=ins_num, label_flips_coarse=flips_c, label_flips_fine=flips_f,
               frac_fine_depths_gt_1e5=float((dz > 1e-5).float().mean()), frac_fine_depths_gt_1e4=float((dz > 1e-4).float().mean()),
               frac_rays_with_a_moved_depth=float((dz > 1e-5).any(-1).float().mean()), max_dz=float(dz.max()),
               psnr_rgb_fine_db=-10 * np.log10(max(mse, 1e-30)), max_abs_rgb_fine=float((got['rgb_fine'] - want['rgb_fine']).abs().max()),
               rel_raw_coarse=maxrel(got['raw_coarse'], want['raw_coarse']), labels_present=int(len(torch.unique(want['ins_fine'].argmax(-1)))))
    with capsys.disabled():
        print("\n[end-to-end 4096 rays] " + json.dumps(rep))
    assert rep["rel_raw_coarse"] <= 1e-5
    assert torch.allclose(got['ins_coarse'], want['ins_coarse'], rtol=2e-6, atol=2e-6)
    assert flips_c <= 4 and flips_f <= 4, rep                       # <= 1e-3 of 4096 rays
    assert rep["frac_fine_depths_gt_1e4"] <= 1e-3 and rep["psnr_rgb_fine_db"] >= 80.0, rep


def test_full_size_properties(A):
    """BASELINE config 1 size (4096 rays x 64+128) without the oracle: structural invariants."""
    ins_num = 13
    mc = model_from(A, O.make_weights(21, ins_num, gain=1.7, sigma_bias=0.3), ins_num)
    mf = model_from(A, O.make_weights(22, ins_num, gain=1.7, sigma_bias=0.3), ins_num)
    K = O.dmsr_intrinsics(480, 640)
    c2w = O.pose_spherical(20.0, -65.0, 7.0)
    ro, rd = A.H.get_rays_k(480, 640, K, dev(c2w))
    ro, rd = ro.reshape(-1, 3)[:4096], rd.reshape(-1, 3)[:4096]
    z = A.H.z_val_sample(4096, 4.0, 15.0, 64)
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    with torch.no_grad():
        out = A.R.dm_nerf(torch.stack([ro, rd]), None, None, mc, mf, z, args)
        out2 = A.R.dm_nerf(torch.stack([ro, rd]), None, None, mc, mf, z, args)
        # chunk independence: rays 1000..1999 alone give the same answer (rays are independent units)
        sub = A.R.dm_nerf(torch.stack([ro[1000:2000], rd[1000:2000]]), None, None, mc, mf, z[1000:2000].contiguous(), args)
    zf = out['z_vals_fine']
    assert bool((zf[:, 1:] >= zf[:, :-1]).all())                               # sortedness
    assert bool(torch.isfinite(out['raw_fine']).all())
    for k in out:
        assert torch.equal(out[k], out2[k]), k                                 # run-to-run determinism
        assert torch.equal(out[k][1000:2000], sub[k]), k
    w_sum = A.R.render_train(out['raw_fine'], zf, rd)[1].sum(-1)
    assert float(w_sum.max()) <= 1.0 + 1e-5 and float(w_sum.min()) >= 0.0      # weights are a sub-probability
    assert float(out['rgb_fine'].min()) >= 0.0 and float(out['rgb_fine'].max()) <= 1.0 + 1e-6
    # compositing is linear in the logits' weights: render_train(raw) reproduces the dict entries
    rgb, w, dep, ins = A.R.render_train(out['raw_fine'], zf, rd)
    assert torch.equal(rgb, out['rgb_fine']) and torch.equal(dep, out['depth_fine']) and torch.equal(ins, out['ins_fine'])


def test_one_call_beyond_2_31_elements_equals_chunked(A):
    """Maximum sizes: 700 001 rays in ONE dm_nerf call -- raw_fine has 2.42e9 elements (> 2^31) and 9.7 GB (> 2^32 bytes
    behind one base pointer), N is not a multiple of the 32-sample tile -- equals the 4096-ray chunks the drivers use,
    bit for bit: at the start, across the 2^31-element boundary (ray 621 378) and in the ragged tail."""
    ins_num, N = 13, 700001
    mc = model_from(A, O.make_weights(23, ins_num, gain=1.7, sigma_bias=0.3), ins_num)
    mf = model_from(A, O.make_weights(24, ins_num, gain=1.7, sigma_bias=0.3), ins_num)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = A.H.get_rays_k(480, 640, K, dev(O.pose_spherical(20.0, -65.0, 7.0)))
    idx = torch.arange(N, device="cuda") % (480 * 640)
    ro, rd = ro.reshape(-1, 3)[idx].contiguous(), rd.reshape(-1, 3)[idx].contiguous()
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    with torch.no_grad():
        big = A.R.dm_nerf(torch.stack([ro, rd]), None, None, mc, mf, A.H.z_val_sample(N, 4.0, 15.0, 64), args)
        assert big['raw_fine'].numel() > 2 ** 31
        for s0, n in ((0, 4096), (619520, 4096), (N - 2913, 2913)):
            part = A.R.dm_nerf(torch.stack([ro[s0:s0 + n], rd[s0:s0 + n]]), None, None, mc, mf, A.H.z_val_sample(n, 4.0, 15.0, 64), args)
            for k in ('raw_coarse', 'raw_fine', 'z_vals_fine', 'rgb_fine', 'ins_fine', 'depth_fine', 'rgb_coarse'):
                assert torch.equal(big[k][s0:s0 + n], part[k]), (k, s0)
        assert bool(torch.isfinite(big['rgb_fine']).all())
        # the frame repeats: ray i and ray i + 307200 are the same ray
        assert torch.equal(big['rgb_fine'][:85601], big['rgb_fine'][614400:])


def test_create_nerf_and_state_dict_roundtrip(A):
    args = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=8, netwidth=256,
                                 ins_num=13, device=torch.device("cuda:0"))
    pe, ve, mc, mf, args2 = A.Cfg.create_nerf(args)
    assert args2 is args and pe.out_dim == 63 and ve.out_dim == 27
    keys = list(mc.state_dict())
    assert keys[:2] == ["mlps.0.weight", "mlps.0.bias"] and len(keys) == 30
    assert "rgb_feature_linears.0.weight" in keys and "ins_linear.bias" in keys
    assert sum(p.numel() for p in mc.parameters()) == 696338
    sd = O.make_weights(5, 13)
    mc.load_state_dict(sd)                                  # reference checkpoints load unchanged
    x = torch.randn(40, 90)
    with torch.no_grad():
        assert maxrel(cpu(mc(dev(x))), O.mlp_forward(sd, x)) <= 1e-5


def test_errors_are_loud(A):
    m = model_from(A, O.make_weights(1, 13), 13)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            m(torch.randn(4, 90))                           # CPU tensor: no fallback
    with pytest.raises(NotImplementedError):
        A.M.DM_NeRF(4, 128, 63, 27, [2], 13).cuda().blob()
    with pytest.raises(RuntimeError):
        A.R.render_train(torch.zeros(2, 2000, 8, device="cuda"), torch.zeros(2, 2000, device="cuda"),
                         torch.ones(2, 3, device="cuda"))    # S beyond the kernel's per-ray staging


@pytest.mark.parametrize("S,n_imp,N", [(40, 24, 50), (16, 200, 7), (100, 1, 33)])
def test_dm_nerf_with_unusual_sample_counts(A, S, n_imp, N):
    """The shipped configs use 64 + 128 samples; the path itself takes any S >= 3 and N_importance >= 1
    (N_samples / N_importance are config values, config.py:41-43).  Whole dict against the oracle."""
    ins_num = 13
    sd_c, sd_f = O.make_weights(83, ins_num, gain=1.7, sigma_bias=0.3), O.make_weights(84, ins_num, gain=1.7, sigma_bias=0.3)
    mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(15.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(S).choice(480 * 640, N, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    z = O.z_val_sample(N, 4.0, 15.0, S).contiguous()
    args = types.SimpleNamespace(perturb=False, N_importance=n_imp, is_train=False, N_ins=None)
    with torch.no_grad():
        want = O.dm_nerf(rays, sd_c, sd_f, z, perturb=0., N_importance=n_imp)
        got = {k: cpu(v) for k, v in A.R.dm_nerf(dev(rays), None, None, mc, mf, dev(z), args).items()}
        raw_f = cpu(A.R.run_network(mf, dev(rays[0]), dev(rays[1]), dev(want['z_vals_fine'])))
    assert got['raw_fine'].shape == (N, S + n_imp, 18) and got['z_vals_fine'].shape == (N, S + n_imp)
    assert maxrel(got['raw_coarse'], want['raw_coarse']) <= 1e-5
    assert maxrel(raw_f, want['raw_fine']) <= 1e-5
    assert torch.allclose(got['rgb_coarse'], want['rgb_coarse'], rtol=2e-6, atol=2e-6)
    assert torch.allclose(got['depth_coarse'], want['depth_coarse'], rtol=2e-6, atol=2e-6)
    zf = got['z_vals_fine']
    assert bool((zf[:, 1:] >= zf[:, :-1]).all())
    assert float(((zf - want['z_vals_fine']).abs() <= 1e-4).float().mean()) >= 0.999
    assert torch.allclose(got['rgb_fine'], want['rgb_fine'], atol=2e-3)


def test_empty_and_single_ray_batches(A):
    """N = 0 (an empty chunk: every entry point returns immediately) and N = 1 (one ray fills 1/32 of one wave's
    sample tile; the other lanes compute on clamped duplicates and store nothing)."""
    ins_num = 13
    sd_c, sd_f = O.make_weights(81, ins_num, gain=1.7, sigma_bias=0.3), O.make_weights(82, ins_num, gain=1.7, sigma_bias=0.3)
    mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    with torch.no_grad():
        out = A.R.dm_nerf(torch.zeros(2, 0, 3, device="cuda"), None, None, mc, mf, torch.zeros(0, 64, device="cuda"), args)
        torch.cuda.synchronize()
        assert out['rgb_fine'].shape == (0, 3) and out['raw_fine'].shape == (0, 192, 18) and out['ins_fine'].shape == (0, 13)
        rgb, w, depth, ins = A.R.render_train(torch.zeros(0, 64, 18, device="cuda"), torch.zeros(0, 64, device="cuda"), torch.zeros(0, 3, device="cuda"))
        assert rgb.shape == (0, 3) and w.shape == (0, 64) and depth.shape == (0,) and ins.shape == (0, 13)
        assert A.H.sample_pdf(torch.zeros(0, 63, device="cuda"), torch.zeros(0, 62, device="cuda"), 128, det=True).shape == (0, 128)
        K = O.dmsr_intrinsics(480, 640)
        ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(10.0, -65.0, 7.0))
        rays = torch.stack([ro.reshape(-1, 3)[123456:123457], rd.reshape(-1, 3)[123456:123457]])
        z = O.z_val_sample(1, 4.0, 15.0, 64).contiguous()
        want = O.dm_nerf(rays, sd_c, sd_f, z, perturb=0.)
        got = {k: cpu(v) for k, v in A.R.dm_nerf(dev(rays), None, None, mc, mf, dev(z), args).items()}
    assert maxrel(got['raw_coarse'], want['raw_coarse']) <= 1e-5
    assert torch.allclose(got['rgb_coarse'], want['rgb_coarse'], rtol=2e-6, atol=2e-6)
    assert torch.allclose(got['rgb_fine'], want['rgb_fine'], atol=1e-4)


def test_get_select_full_same_rng_stream_as_reference(A, golden):
    """identical numpy stream => identical pixels, targets and rays."""
    g = golden("select")
    H_, W_, N = [int(v) for v in g["HWN"]]
    np.random.seed(0)
    tc, ti, rays = A.H.get_select_full(dev(g["rgb"]), dev(g["c2w"])[:3, :4], g["K"].numpy(), dev(g["lab"]), N)
    assert rays.shape == (2, N, 3)
    assert torch.equal(cpu(tc), g["target_c"]) and torch.equal(cpu(ti), g["target_i"])
    assert torch.equal(cpu(rays[0]), g["rays"][0])
    assert torch.allclose(cpu(rays[1]), g["rays"][1], rtol=3e-7, atol=1e-7)
    # and the host RNG advanced exactly as in the reference: the next draw matches
    np.random.seed(0)
    np.random.choice(H_ * W_, size=[N], replace=False)
    want_next = np.random.rand()
    np.random.seed(0)
    A.H.get_select_full(dev(g["rgb"]), dev(g["c2w"])[:3, :4], g["K"].numpy(), dev(g["lab"]), N)
    assert np.random.rand() == want_next


def test_get_select_crop_same_rng_stream_as_reference(A, golden):
    """ScanNet batch (helpers.py:64-95): identical numpy stream => identical pixels, targets, rays and N_ins;
    both the 30 % quota case and the clamp to the number of labelled pixels."""
    g = golden("select_crop")
    crop = g["crop"].numpy()
    for name in ("many", "few"):
        N, n_ins_want = [int(v) for v in g[f"{name}_N"]]
        np.random.seed(3)
        tc, ti, rays, n_ins = A.H.get_select_crop(dev(g["rgb"]), dev(g["c2w"])[:3, :4], g["K"].numpy(), dev(g["lab"]),
                                                  g[f"{name}_ins_index"].numpy(), crop, N)
        assert np.random.rand() == float(g[f"{name}_next_rand"])         # host RNG advanced exactly as in the reference
        assert n_ins == n_ins_want and rays.shape == (2, N, 3) and ti.shape == (n_ins,)
        assert torch.equal(cpu(tc), g[f"{name}_target_c"]) and torch.equal(cpu(ti), g[f"{name}_target_i"])
        assert torch.equal(cpu(rays[0]), g[f"{name}_rays"][0])
        assert torch.allclose(cpu(rays[1]), g[f"{name}_rays"][1], rtol=3e-7, atol=1e-7)


def test_fused_heads_inference_matches_layerwise(A):
    """rgb_feature_linear / ins_feature_linear folded into the hidden layers.  Same function
    up to f32 re-association: the MLP tolerance of the contract, |d raw| <= 1e-5 (1 + |raw|), against the oracle AND
    against the layer-by-layer kernel; the 10-key dict of dm_nerf stays within the end-to-end tolerances."""
    for ins_num, seed in ((13, 3), (59, 4)):
        sd = O.make_weights(seed, ins_num, gain=1.7, sigma_bias=0.3)
        m = A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
        m.load_state_dict(sd); m = m.cuda()
        g = torch.Generator().manual_seed(seed)
        N, S = 37, 64
        ro, rd = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
        z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0]
        lib = A.lib.load()
        raw_f = torch.empty(N, S, 4 + ins_num + 1, device="cuda")
        ro_d, rd_d, z_d = dev(ro), dev(rd), dev(z)                    # (kept alive across the raw-pointer call)
        A.lib.check(lib.dmnerf_mlp_fwd_rays_fused(A.lib.ptr(m.blob_fused()), ins_num, A.lib.ptr(ro_d), A.lib.ptr(rd_d), A.lib.ptr(z_d),
                                                N, S, A.lib.ptr(raw_f), A.lib.stream()), "fused")
        raw_l = A.R.run_network(m, ro_d, rd_d, z_d)
        pts = ro[:, None, :] + rd[:, None, :] * z[:, :, None]
        vd = rd / torch.norm(rd, dim=-1, keepdim=True)
        x = torch.cat([O.embed(pts.reshape(-1, 3), 10), O.embed(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
        want = O.mlp_forward(sd, x).reshape(N, S, -1)
        for got, what in ((cpu(raw_f), "fused vs oracle"), (cpu(raw_l), "layerwise vs oracle")):
            assert bool(((got - want).abs() <= 1e-5 * (1 + want.abs())).all()), (what, ins_num, float((got - want).abs().max()))
        assert bool(((cpu(raw_f) - cpu(raw_l)).abs() <= 1e-5 * (1 + cpu(raw_l).abs())).all())
    # through dm_nerf: args.fuse_heads
    ins_num = 13
    sd_c, sd_f = O.make_weights(11, ins_num, gain=1.7, sigma_bias=0.3), O.make_weights(12, ins_num, gain=1.7, sigma_bias=0.3)
    mc, mf = A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num), A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
    mc.load_state_dict(sd_c); mf.load_state_dict(sd_f); mc, mf = mc.cuda(), mf.cuda()
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(30.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(1).choice(480 * 640, 256, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0).cuda()
    z = A.H.z_val_sample(256, 4.0, 15.0, 64)
    base = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    fuse = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None, fuse_heads=True)
    with torch.no_grad():
        a, b = A.R.dm_nerf(rays, None, None, mc, mf, z, base), A.R.dm_nerf(rays, None, None, mc, mf, z, fuse)
    assert float((a['raw_coarse'] - b['raw_coarse']).abs().max()) <= 1e-5 * (1 + float(a['raw_coarse'].abs().max()))
    assert float((a['rgb_coarse'] - b['rgb_coarse']).abs().max()) <= 5e-6
    assert float((a['rgb_fine'] - b['rgb_fine']).abs().max()) <= 2e-3       # through the ill-conditioned inverse-CDF step
    assert float(((a['ins_fine'].argmax(-1) != b['ins_fine'].argmax(-1)).float().mean())) <= 1e-2


@pytest.mark.parametrize("split", ["bf16x3", "f16x2"])
def test_split_bf16_inference_is_f32_class(A, split):
    """Opt-in split-operand MFMA paths (bf16x3: three bf16 planes per operand, six products; f16x2: two f16 planes, three
    products; f32 accumulation): same tolerance as the f32 kernels, |d raw| <= 1e-5 (1 + |raw|) against the oracle, for 1, 2
    and 3 logit blocks and ragged batches (incl. one that ends inside a 128-sample workgroup and one of a single sample)."""
    for ins_num, seed, N, S in ((13, 3, 37, 64), (59, 4, 9, 21), (93, 5, 6, 33), (13, 6, 1, 1), (120, 7, 3, 50)):
        sd = O.make_weights(seed, ins_num, gain=1.7, sigma_bias=0.3)
        m = A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
        m.load_state_dict(sd); m = m.cuda()
        g = torch.Generator().manual_seed(seed)
        ro, rd = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
        z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0]
        lib = A.lib.load()
        raw_s = torch.empty(N, S, 4 + ins_num + 1, device="cuda")
        ro_d, rd_d, z_d, blob = dev(ro), dev(rd), dev(z), (m.blob_split() if split == "bf16x3" else m.blob_f16())
        fn = lib.dmnerf_mlp_fwd_rays_split if split == "bf16x3" else lib.dmnerf_mlp_fwd_rays_f16
        A.lib.check(fn(A.lib.ptr(blob), ins_num, A.lib.ptr(ro_d), A.lib.ptr(rd_d), A.lib.ptr(z_d),
                                                N, S, A.lib.ptr(raw_s), A.lib.stream()), "split")
        pts = ro[:, None, :] + rd[:, None, :] * z[:, :, None]
        vd = rd / torch.norm(rd, dim=-1, keepdim=True)
        x = torch.cat([O.embed(pts.reshape(-1, 3), 10), O.embed(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
        want = O.mlp_forward(sd, x).reshape(N, S, -1)
        got = cpu(raw_s)
        assert bool(((got - want).abs() <= 1e-5 * (1 + want.abs())).all()), (split, ins_num, float(((got - want).abs() / (1 + want.abs())).max()))
    # through dm_nerf
    ins_num = 13
    sd_c, sd_f = O.make_weights(11, ins_num, gain=1.7, sigma_bias=0.3), O.make_weights(12, ins_num, gain=1.7, sigma_bias=0.3)
    mc, mf = A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num), A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
    mc.load_state_dict(sd_c); mf.load_state_dict(sd_f); mc, mf = mc.cuda(), mf.cuda()
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(30.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(1).choice(480 * 640, 256, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0).cuda()
    z = A.H.z_val_sample(256, 4.0, 15.0, 64)
    base = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    sargs = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None, mfma_split=split)
    with torch.no_grad():
        a, b = A.R.dm_nerf(rays, None, None, mc, mf, z, base), A.R.dm_nerf(rays, None, None, mc, mf, z, sargs)
    assert float((a['raw_coarse'] - b['raw_coarse']).abs().max()) <= 1e-5 * (1 + float(a['raw_coarse'].abs().max()))
    assert float((a['rgb_coarse'] - b['rgb_coarse']).abs().max()) <= 5e-6
    assert float((a['rgb_fine'] - b['rgb_fine']).abs().max()) <= 2e-3       # through the ill-conditioned inverse-CDF step


_opt_in_oracle = {}         # (ins_num, near, far) -> the oracle's dict: the three modes are checked against the same CPU render (4 s each)


@pytest.mark.parametrize("mode", ["fuse_heads", "mfma_split", "mfma_split=f16x2"])
@pytest.mark.parametrize("ins_num,near,far", [(13, 4.0, 15.0), (59, 0.0, 4.7), (93, 0.0, 4.7)])
def test_opt_in_inference_modes_full_dict_vs_oracle(A, mode, ins_num, near, far, capsys):
    """The two opt-in inference modes (never the default, never the headline metric) held to the DEFAULT path's contract on
    the whole 10-key dict, for the three object-code widths: raw_coarse within 1e-5 (1 + |raw|), coarse maps 5e-6, and end
    to end PSNR / label flips against the oracle on 1024 rays (printed: the figures DESIGN.md quotes for the modes)."""
    import json
    sd_c = O.make_weights(500 + ins_num, ins_num, gain=1.7, sigma_bias=0.3)
    sd_f = O.make_weights(600 + ins_num, ins_num, gain=1.7, sigma_bias=0.3)
    mc, mf = model_from(A, sd_c, ins_num), model_from(A, sd_f, ins_num)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(160.0, -65.0, 7.0))
    sel = torch.from_numpy(np.random.RandomState(2000 + ins_num).choice(480 * 640, 1024, replace=False))
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    z = O.z_val_sample(1024, near, far, 64).contiguous()
    key, _, val = mode.partition("=")
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None, **{key: (val or True)})
    with torch.no_grad():
        got = {k: cpu(v) for k, v in A.R.dm_nerf(dev(rays), None, None, mc, mf, dev(z), args).items()}
        if (ins_num, near, far) not in _opt_in_oracle:
            _opt_in_oracle[(ins_num, near, far)] = O.dm_nerf(rays, sd_c, sd_f, z, perturb=0.)
        want = _opt_in_oracle[(ins_num, near, far)]
    assert set(got) == set(want)
    for k in want:
        assert got[k].shape == want[k].shape, k
    rel = maxrel(got['raw_coarse'], want['raw_coarse'])
    mse = float(((got['rgb_fine'] - want['rgb_fine']).double() ** 2).mean())
    rep = dict(mode=mode, ins_num=ins_num, rel_raw_coarse=rel, max_abs_rgb_coarse=float((got['rgb_coarse'] - want['rgb_coarse']).abs().max()),
               label_flips_coarse=int((got['ins_coarse'].argmax(-1) != want['ins_coarse'].argmax(-1)).sum()),
               label_flips_fine=int((got['ins_fine'].argmax(-1) != want['ins_fine'].argmax(-1)).sum()),
               psnr_rgb_fine_db=-10 * np.log10(max(mse, 1e-30)), frac_fine_depths_gt_1e4=float(((got['z_vals_fine'] - want['z_vals_fine']).abs() > 1e-4).float().mean()))
    with capsys.disabled():
        print("\n[opt-in mode, 1024 rays] " + json.dumps(rep))
    assert rel <= 1e-5 and rep["max_abs_rgb_coarse"] <= 5e-6, rep
    assert torch.allclose(got['ins_coarse'], want['ins_coarse'], rtol=5e-6, atol=5e-6)
    assert rep["label_flips_coarse"] <= 1 and rep["label_flips_fine"] <= 2, rep          # <= 1e-3 .. 2e-3 of 1024 rays
    assert rep["psnr_rgb_fine_db"] >= 75.0 and rep["frac_fine_depths_gt_1e4"] <= 2e-3, rep
