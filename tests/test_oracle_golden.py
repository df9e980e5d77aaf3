"""The CPU oracle (oracle/ref_cpu.py) against the committed golden vectors.

The vectors were produced by running the reference itself (tests/golden/make_golden.py,
bit-exact there).  On another host the CPU's BLAS / vector-math code paths can differ in the
last ulp, so GEMM / transcendental stages are compared with a tight tolerance and the
integer / permutation stages exactly.
"""
import numpy as np
import torch

from oracle import ref_cpu as O

torch.set_num_threads(1)


def close(a, b, rtol=2e-6, atol=2e-6):
    assert a.shape == b.shape
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


def test_embed(golden):
    g = golden("embed")
    close(O.embed(g["x"], 10), g["y10"], atol=5e-7, rtol=0)
    close(O.embed(g["d"], 4), g["y4"], atol=5e-7, rtol=0)
    assert O.embed_out_dim(10) == 63 and O.embed_out_dim(4) == 27
    # layout: [x | sin(x) | cos(x) | sin(2x) ...] in blocks of 3 (networks/dm_nerf.py:37)
    y = O.embed(g["x"], 10)
    assert torch.equal(y[:, :3], g["x"])
    assert torch.allclose(y[:, 9:12], torch.sin(g["x"] * 2.0))


def test_mlp(golden):
    g = golden("mlp")
    for ins_num in (13, 59, 93):
        sd = O.make_weights(int(g[f"seed_{ins_num}"]), ins_num, gain=float(g["gain"]))
        y = O.mlp_forward(sd, g[f"x_{ins_num}"])
        assert y.shape == (80, 4 + ins_num + 1)
        close(y, g[f"y_{ins_num}"], rtol=1e-5, atol=1e-5)


def test_param_inventory():
    # 696 338 parameters at ins_num=13; key names = reference state_dict
    sd = O.make_weights(0, 13)
    assert sum(v.numel() for v in sd.values()) == 696338
    assert list(sd)[:2] == ["mlps.0.weight", "mlps.0.bias"]
    assert sd["mlps.5.weight"].shape == (256, 319)
    assert sd["rgb_feature_linears.0.weight"].shape == (128, 283)
    assert sd["ins_linear.weight"].shape == (14, 128)


def test_render_train(golden):
    g = golden("render_train")
    for k in ("S64_C14", "S192_C14", "S320_C60", "S192_C94", "S5_C3", ):
        rgb, w, dep, ins = O.render_train(g[f"{k}_raw"], g[f"{k}_z"], g[f"{k}_d"])
        close(rgb, g[f"{k}_rgb"]); close(w, g[f"{k}_w"]); close(dep, g[f"{k}_depth"], rtol=1e-5); close(ins, g[f"{k}_ins"])
    rgb, w, dep, ins = O.render_train(g["kat_raw"], g["kat_z"], g["kat_d"])
    # SURVEY 8(a-8) known answers
    assert np.allclose(w[0, :3].numpy(), [0.16020966, 0.13454282, 0.11298747], rtol=1e-6)
    assert abs(float(dep[0]) - 4.915222645) < 1e-5 and abs(float(ins[0, 2]) - 0.95257413) < 1e-6
    assert np.allclose(rgb.numpy(), 0.5, atol=1e-6)


def test_sample_pdf(golden):
    g = golden("sample_pdf")
    s, cdf, inds = O.sample_pdf(g["bins"], g["w"], 128, det=True, return_aux=True)
    close(cdf, g["cdf"], rtol=0, atol=2e-7)
    close(s, g["s_det"], rtol=1e-5)
    # stage-isolated: identical (cdf, u) -> identical indices and samples
    s2, i2 = O.sample_from_cdf(g["bins"], g["cdf"], g["u_rnd"])
    assert torch.equal(i2, g["inds_rnd"])
    close(s2, g["s_rnd"], rtol=1e-6, atol=1e-6)
    # SURVEY 8(a-9) KAT: uniform weights on the (4,15) grid
    kat = g["s_det"][0, [0, 1, 64, 126, 127]].numpy()
    assert np.allclose(kat, [4.087301254, 4.172540665, 9.542618752, 14.827458382, 14.912697792], rtol=1e-6)
    # merged fine depths are a sorted permutation of coarse + samples
    zf = torch.sort(torch.cat([g["z"], g["s_det"]], -1), -1)[0]
    assert torch.equal(zf, g["zf_det"])


def test_rays_and_zvals(golden):
    g = golden("rays")
    H, W = [int(v) for v in g["HW"]]
    for name in ("dmsr", "replica", "scannet"):
        o, d = O.get_rays_k(H, W, g[f"K_{name}"].numpy(), g["c2w"])
        close(d, g[f"d_{name}"], rtol=1e-6, atol=1e-7)
        assert torch.equal(o.contiguous(), g[f"o_{name}"])
    z = O.z_val_sample(3, 4.0, 15.0, 64)
    assert torch.equal(z.contiguous(), g["z_4_15"])
    assert [float(z[0, i]) for i in (0, 1, 31, 63)] == [4.0, 4.17460298538208, 9.412698745727539, 15.0]
    assert torch.equal(O.stratify(g["z_4_15"], g["t_rand"]), g["z_jit"])


def test_dm_nerf_dict(golden):
    g = golden("dm_nerf")
    ins_num = int(g["ins_num"])
    sd_c = O.make_weights(int(g["seed_c"]), ins_num, gain=float(g["gain"]), sigma_bias=float(g["sigma_bias"]))
    sd_f = O.make_weights(int(g["seed_f"]), ins_num, gain=float(g["gain"]), sigma_bias=float(g["sigma_bias"]))
    with torch.no_grad():
        out = O.dm_nerf(g["rays"], sd_c, sd_f, g["z_in"], perturb=0.)
    assert set(out) == {'rgb_fine', 'ins_fine', 'z_vals_fine', 'raw_fine', 'raw_coarse', 'rgb_coarse',
                        'ins_coarse', 'z_vals_coarse', 'depth_fine', 'depth_coarse'}
    assert out['raw_fine'].shape == (24, 192, 18) and out['ins_fine'].shape == (24, 13)
    for k, v in out.items():
        close(v, g[f"det_{k}"], rtol=2e-4, atol=2e-4)
    with torch.no_grad():
        out = O.dm_nerf(g["rays"], sd_c, sd_f, g["z_in"], perturb=1.0, is_train=True, N_ins=7,
                        t_rand=g["t_rand"], u=g["u"])
    assert out['ins_fine'].shape == (7, 13) and out['ins_coarse'].shape == (7, 13)
    for k, v in out.items():
        close(v, g[f"prt_{k}"], rtol=2e-4, atol=2e-4)


def test_ins_criterion_golden(golden):
    """Object-code loss (networks/evaluator.py:19-74): oracle == the reference's outputs and gradient, bit for bit."""
    g = golden("ins_criterion")
    for name in ("all", "some", "wide"):
        ins_num = int(g[f"{name}_ins_num"])
        pred = g[f"{name}_pred"].clone().requires_grad_(True)
        out = O.ins_criterion(pred, g[f"{name}_lab"].long(), ins_num)
        out[0].sum().backward()
        got = torch.stack([t.detach().float().reshape(()) for t in out])
        assert torch.equal(got, g[f"{name}_out"]), name
        assert torch.equal(pred.grad, g[f"{name}_grad"]), name
        cc, cs, valid = O.ins_cost_matrices(g[f"{name}_pred"], g[f"{name}_lab"].long(), ins_num)
        assert torch.equal(cc[:valid], g[f"{name}_cost_ce"]) and torch.equal(cs[:valid], g[f"{name}_cost_siou"])
