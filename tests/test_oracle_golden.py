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


def test_frame_pixels(golden):
    """640 x 480 frame fixture (tester.py:58-77 on 3319 + 1024 pixels of one pose): the oracle on a slice of both weight sets."""
    g = golden("frame")
    H, W = [int(v) for v in g["HW"]]
    ins_num = int(g["ins_num"])
    ro, rd = O.get_rays_k(H, W, g["K"].numpy(), g["c2w"])
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    for name, kw, tol in (("plain", dict(gain=1.7, sigma_bias=0.3), 2e-4), ("peaky", O.PEAKY, 5e-2)):
        s_c, s_f = [int(v) for v in g[f"{name}_seeds"]]
        pix = g[f"{name}_pix"][:96]
        rays = torch.stack([ro[pix], rd[pix]], 0)
        z = O.z_val_sample(len(pix), 4.0, 15.0, 64)
        with torch.no_grad():
            out = O.dm_nerf(rays, O.make_weights(s_c, ins_num, **kw), O.make_weights(s_f, ins_num, **kw), z, perturb=0.)
        close(out['rgb_coarse'], g[f"{name}_rgb_coarse"][:96], rtol=1e-5, atol=1e-5)
        assert float((out['rgb_fine'] - g[f"{name}_rgb"][:96]).abs().max()) <= tol
        assert float((out['ins_fine'].argmax(-1) != g[f"{name}_label"][:96]).float().mean()) <= 0.03
    assert len(np.unique(g["peaky_label"].numpy())) >= 6                  # the peaky frame has a varied label map


def test_scannet_step_forward_slice(golden):
    """ScanNet-form step fixture (train_scannet.py:24-64, 3072 rays): batch selection re-derived from the seeds, and the oracle's
    forward on the first 64 and the labelled last 64 rays with the recorded jitter."""
    g = golden("scannet_step")
    H, W = [int(v) for v in g["HW"]]
    ins_num = int(g["ins_num"])
    N, N_ins = [int(v) for v in g["N"]]
    rgb, lab, crop, ins_index = O.scannet_scene(H, W, ins_num)
    assert len(ins_index) == int(g["ins_index_len"]) and N_ins == int(N * 0.3)
    # get_select_crop's draws (helpers.py:64-95) on the recorded numpy stream
    np.random.seed(0)
    labeled = ins_index[np.random.choice(ins_index.shape[0], size=[N_ins], replace=False)]
    crop_idx = np.where(crop.reshape(-1) == 1)[0]
    unl = crop_idx[np.random.choice(len(set(crop_idx.tolist()) - set(labeled.tolist())), size=[N - N_ins], replace=False)]
    assert np.random.rand() == float(g["next_rand"])
    flat = np.concatenate([unl, labeled])
    assert torch.equal(rgb.reshape(-1, 3)[flat], g["target_c"])
    assert torch.equal(torch.from_numpy(lab.reshape(-1)[labeled].astype(np.int32)), g["target_i"])
    ro, rd = O.get_rays_k(H, W, g["K"].numpy(), g["c2w"][:3, :4])
    close(rd.reshape(-1, 3)[flat], g["rays"][1], rtol=1e-6, atol=1e-7)
    s_c, s_f, s_jit, _ = [int(v) for v in g["seeds"]]
    torch.manual_seed(s_jit)
    t_rand, u = torch.rand(N, 64), torch.rand(N, 128)
    sel = torch.cat([torch.arange(64), torch.arange(N - 64, N)])
    z = O.z_val_sample(128, 0.0, 9.5, 64)
    with torch.no_grad():
        out = O.dm_nerf(g["rays"][:, sel], O.make_weights(s_c, ins_num, gain=1.7, sigma_bias=0.3), O.make_weights(s_f, ins_num, gain=1.7, sigma_bias=0.3),
                        z, perturb=1.0, is_train=True, N_ins=64, t_rand=t_rand[sel], u=u[sel])
    close(out['rgb_coarse'], g["rgb_coarse"][sel], rtol=1e-5, atol=1e-5)
    close(out['ins_coarse'], g["ins_coarse"][-64:], rtol=1e-5, atol=1e-5)
    close(out['z_vals_fine'], g["z_vals_fine"][sel], rtol=0, atol=5e-2)
    assert float(((out['z_vals_fine'] - g["z_vals_fine"][sel]).abs() > 1e-4).float().mean()) <= 2e-3
    close(out['rgb_fine'], g["rgb_fine"][sel], rtol=1e-3, atol=1e-3)


def test_manipulator_stages(golden):
    """Recorded intermediates of one run of the reference's manipulator(): the oracle's stages on the golden inputs."""
    g = golden("manipulator_stages")
    labels = [int(v) for v in g["labels"]]
    # first exchanger call: exact
    out = O.exchanger(g["ex1_ori_raw_in"].clone(), [g["ex1_tar_raw0"].clone(), g["ex1_tar_raw1"].clone()], g["ex1_ori_acc"],
                      [g["ex1_tar_acc0"], g["ex1_tar_acc1"]], labels)
    assert torch.equal(out[0], g["ex1_out_raw"])
    # step 2: weights of the edited coarse field, the fourth resampling, the merged depths, the second exchanger, the final render
    _, w, _, _ = O.manipulator_render(g["ex1_out_raw"], g["s2_z"], g["ori_rays"][1])
    close(w, g["s2_w"])
    assert torch.equal(g["s2_w"][..., 1:-1], g["pdf3_w"])
    mid = .5 * (g["s2_z"][..., 1:] + g["s2_z"][..., :-1])
    close(O.sample_pdf(mid, g["pdf3_w"], 128, u=g["pdf3_u"]), g["pdf3_out"], rtol=1e-5, atol=1e-5)
    merged = torch.sort(torch.cat([g["s2_z"], g["pdf3_out"], g["pdf1_out"], g["pdf2_out"]], -1), -1)[0]
    assert torch.equal(merged, g["s2_ori_z_merged"])
    out = O.exchanger(g["s2_ori_raw"].clone(), [g["s2_tar_raw0"].clone(), g["s2_tar_raw1"].clone()], g["ex1_ori_acc"],
                      [g["ex1_tar_acc0"], g["ex1_tar_acc1"]], labels)
    assert torch.equal(out[0], g["ex2_out_raw"]) and torch.equal(out[2], g["ex2_out_ori_label"]) and torch.equal(out[3], g["ex2_out_tar_label"])
    rgb, _, _, ins = O.manipulator_render(g["ex2_out_raw"], g["s2_ori_z_merged"], g["ori_rays"][1])
    close(rgb, g["final_rgb"]); close(ins, g["final_ins"])


def test_manipulator_frame(golden):
    """One pose through the reference's own ``manipulator_eval`` chunk loop (tests/golden/make_golden.py::gen_manipulator_frame):
    the oracle's restatement of the loop reproduces rays, target pose and the four accumulated outputs from the recorded draws."""
    g = golden("manipulator_frame")
    H, W, N_test = [int(v) for v in g["HWN"]]
    sd_c = O.make_weights(int(g["seeds"][0]), int(g["ins_num"]), **O.PEAKY)
    sd_f = O.make_weights(int(g["seeds"][1]), int(g["ins_num"]), **O.PEAKY)
    n_chunks = -(-H * W // N_test)
    us = [[g[f"u{c}_{i}"] for i in range(3)] for c in range(n_chunks)]
    with torch.no_grad():
        out = O.manipulate_frame(sd_c, sd_f, H, W, g["K"].numpy(), g["ori_pose"], g["trans"], N_test, 64, 128, 4.0, 15.0, [int(g["label"])], us=us)
    assert torch.equal(out[4], g["tar_pose"])
    ro, rd = O.get_rays_k(H, W, g["K"].numpy(), out[4])
    assert torch.equal(ro.reshape(-1, 3), g["tar_rays"][0]) and torch.equal(rd.reshape(-1, 3), g["tar_rays"][1])
    for got, name in zip(out[:4], ("full_rgb", "full_ins", "full_tar_rgb", "full_tar_ins")):
        want = g[name]
        assert got.shape == (H, W, want.shape[-1])
        close(got.reshape(want.shape), want, rtol=1e-5, atol=1e-5)
    assert int((out[1].reshape(-1, out[1].shape[-1]).argmax(-1) != g["full_ins"].argmax(-1)).sum()) == 0


def test_checkpoint_format_matches_the_module():
    """The checkpoint structure the reference writes (train_dmsr.py:78-86; checkpoint_format.json was produced from the
    reference's own DM_NeRF + Adam) against the drop-in module's state_dict -- no GPU needed for key names and shapes."""
    import json
    import os
    from dm_nerf_amd.networks.dm_nerf import DM_NeRF
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint_format.json")) as f:
        fmt = json.load(f)
    m = DM_NeRF(8, 256, 63, 27, [4], 13)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == fmt["model_keys"]
    assert fmt["top_keys"] == ['iteration', 'network_coarse_state_dict', 'network_fine_state_dict', 'optimizer_state_dict']
    assert fmt["state_shapes"][:2] == [[256, 63], [256]] and fmt["n_params"] == 60
