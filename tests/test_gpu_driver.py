"""GPU tests of what sits around the hot path: the full-frame driver (reference: the per-pose body of render_test,
networks/tester.py:58-85) and the boundary's promise that the library never allocates, frees or synchronises
(SURVEY 8(b) "Ownership"), which makes a render step capturable in a HIP graph."""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from dm_nerf_amd import _lib, distributed as D
    from dm_nerf_amd.networks import dm_nerf as M, helpers as H, render as R
    _lib.load()
    return types.SimpleNamespace(M=M, H=H, R=R, D=D)


def models(A, ins_num=13):
    out = []
    for seed in (61, 62):
        m = A.M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
        m.load_state_dict(O.make_weights(seed, ins_num, gain=1.7, sigma_bias=0.3))
        out.append(m.cuda().eval())
    return out


def test_render_frame_ragged_chunks_equal_one_shot_render(A):
    """A 9 x 13 frame in chunks of 50 rays (two full chunks + a ragged one of 17, tester.py:65-67) assembles to exactly
    what one dm_nerf call over all 117 rays returns: rays are independent, so chunking must not change a bit."""
    H, W = 9, 13
    mc, mf = models(A)
    K = np.array([[20.0, 0, W / 2], [0, -20.0, H / 2], [0, 0, -1]])
    c2w = O.pose_spherical(30.0, -65.0, 7.0).cuda()
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    with torch.no_grad():
        rgb, ins, depth = A.D.render_frame(H, W, K, c2w, (mc, mf), 4.0, 15.0, args, chunk=50, n_samples=64)
        ro, rd = A.H.get_rays_k(H, W, K, c2w)
        z = A.H.z_val_sample(H * W, 4.0, 15.0, 64, device=ro.device)
        one = A.R.dm_nerf(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)]), None, None, mc, mf, z, args)
    assert rgb.shape == (H, W, 3) and ins.shape == (H, W, 13) and depth.shape == (H, W)
    assert torch.equal(rgb.reshape(-1, 3), one['rgb_fine'])
    assert torch.equal(ins.reshape(-1, 13), one['ins_fine'])
    assert torch.equal(depth.reshape(-1), one['depth_fine'])
    assert torch.equal(ins.argmax(-1).reshape(-1), one['ins_fine'].argmax(-1))
    with torch.no_grad():
        rgb2, label, conf, depth2 = A.D.render_frame(H, W, K, c2w, (mc, mf), 4.0, 15.0, args, chunk=50, n_samples=64, labels_only=True)
    assert torch.equal(rgb2, rgb) and torch.equal(depth2, depth)
    assert label.dtype == torch.int64 and torch.equal(label.cpu(), ins.cpu().argmax(-1))        # ins_eval runs on ins.cpu()
    assert torch.equal(conf.cpu(), ins.cpu().max(-1).values)


def test_render_path_on_the_device(A):
    """The pose loop through the HIP path: two poses, ScanNet-style crop window, ground truth for the device PSNR."""
    H, W = 9, 13
    mc, mf = models(A)
    K = np.array([[20.0, 0, W / 2], [0, -20.0, H / 2], [0, 0, -1]])
    poses = torch.stack([O.pose_spherical(30.0, -65.0, 7.0), O.pose_spherical(80.0, -65.0, 7.0)]).cuda()
    mask = torch.zeros(H, W, dtype=torch.int64)
    mask[1:7, 2:12] = 1
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None, N_test=50, N_samples=64,
                                 near=4.0, far=15.0, crop_height=6, crop_width=10)
    gt = torch.rand(2, 6, 10, 3, generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        out = A.D.render_path(poses, (H, W, K), (mc, mf), args, gt_imgs=gt, crop_mask=mask, labels_only=True)
        rgb, ins, depth = A.D.render_frame(H, W, K, poses[1], (mc, mf), 4.0, 15.0, args, chunk=50, n_samples=64)
    assert out["rgb"].shape == (2, 6, 10, 3) and out["label"].shape == (2, 6, 10) and out["psnr"].shape == (2,)
    assert torch.equal(out["rgb"][1], rgb[1:7, 2:12]) and torch.equal(out["depth"][1], depth[1:7, 2:12])
    assert torch.equal(out["label"][1].cpu(), ins[1:7, 2:12].cpu().argmax(-1))
    want = -10 * torch.log10(((rgb[1:7, 2:12] - gt[1]) ** 2).mean())
    assert torch.allclose(out["psnr"][1], want)


def test_ins_label_conf_ties_and_wide_rows(A):
    """First maximum wins (torch.argmax on CPU, what ins_eval sees), for the three object-code widths of the configs."""
    from dm_nerf_amd.networks import evaluator as E
    g = torch.Generator().manual_seed(3)
    for C in (13, 59, 93):
        x = torch.rand(1000, C, generator=g)
        x[::7] = torch.round(x[::7] * 4) / 4                     # many exact ties
        x[5] = 0.25
        label, conf = E.ins_label_conf(x.cuda().reshape(10, 100, C))
        assert label.shape == (10, 100) and torch.equal(label.cpu().reshape(-1), x.argmax(-1))
        assert torch.equal(conf.cpu().reshape(-1), x.max(-1).values)
    with pytest.raises(RuntimeError):
        E.ins_label_conf(torch.rand(4, 13))                      # CPU tensors are refused: there is no CPU path


def test_render_step_is_graph_capturable(A):
    """No allocation, free or synchronisation inside the library: a dm_nerf render step records into a HIP graph and
    the replay on new ray data reproduces the eager result bit for bit."""
    N = 256
    mc, mf = models(A)
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(50.0, -65.0, 7.0))
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    pick = lambda s: torch.stack([ro[s:s + N], rd[s:s + N]]).cuda()
    z = A.H.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    args = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    rays = pick(1000)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on a side stream (weight blobs packed, workspaces sized)
            A.R.dm_nerf(rays, None, None, mc, mf, z, args)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = A.R.dm_nerf(rays, None, None, mc, mf, z, args)
        rays.copy_(pick(150000))
        graph.replay()
        torch.cuda.synchronize()
        got = {k: v.clone() for k, v in out.items()}
        want = A.R.dm_nerf(pick(150000), None, None, mc, mf, z, args)
    for k in ("rgb_fine", "ins_fine", "depth_fine", "z_vals_fine", "raw_fine", "raw_coarse"):
        assert torch.equal(got[k], want[k]), k
    assert float(got["rgb_fine"].std()) > 0


@pytest.mark.parametrize("mode", [None, "f16x2"])
def test_training_step_is_graph_capturable(A, mode):
    """The whole optimisation step -- dm_nerf with saved activations, img2mse + Hungarian-matched ins_criterion + emptiness
    penalizer on both levels, composite / dgrad / wgrad backward, Adam, weight re-packing -- records into ONE HIP graph
    (dm_nerf_amd.graphed.GraphedTrainStep) and replays on new batches: parameters after three replayed steps are bit-equal to
    three eager steps from the same start on the same batches and the same jitter stream."""
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.graphed import GraphedTrainStep
    N, ins_num = 96, 13
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(50.0, -65.0, 7.0))
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    g = torch.Generator().manual_seed(11)
    batches = []
    for s in (1000, 90000, 200000, 5000):
        batches.append((torch.stack([ro[s:s + N], rd[s:s + N]]).cuda(), torch.rand(N, 3, generator=g).cuda(),
                        torch.randint(0, 5, (N,), generator=g).cuda()))
    z = A.H.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05,
                                 mfma_split=mode or False)

    def fresh():
        mc, mf = models(A)
        mc.train(); mf.train()
        opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=torch.tensor(5e-4, device="cuda"), capturable=True)
        return mc, mf, opt

    # eager
    mc, mf, opt = fresh()
    torch.cuda.manual_seed(123)
    eager_losses = []
    for rays, tgt, lab in batches[1:]:
        eager_losses.append(D.sharded_train_step(rays, z, tgt, lab, (mc, mf), args, opt, ins_num)[0].clone())
    want = [p.detach().clone() for m in (mc, mf) for p in m.parameters()]
    # graphed: constructed (and warmed up) on another batch, then the same three batches
    mc2, mf2, opt2 = fresh()
    gs = GraphedTrainStep((mc2, mf2), opt2, args, ins_num, batches[0][0], z, batches[0][1], batches[0][2])
    start = [p.detach().clone() for m in models(A) for p in m.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(start, [p for m in (mc2, mf2) for p in m.parameters()])), "warm-up was not undone"
    torch.cuda.manual_seed(123)
    graph_losses = []
    for rays, tgt, lab in batches[1:]:
        graph_losses.append(gs.step(rays, z, tgt, lab).clone())
    torch.cuda.synchronize()
    got = [p.detach() for m in (mc2, mf2) for p in m.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(eager_losses, graph_losses)), (eager_losses, graph_losses)
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    assert not torch.equal(want[0], start[0])                                   # the steps really moved the weights


def test_eager_render_between_graph_replays_sees_the_current_weights(A):
    """train_dmsr.py:88-100 renders test views every i_test iterations: an eager dm_nerf call between two replays of the graphed
    step must use the parameters as the LAST replay left them.  (Replays do not bump the parameters' ``_version``, the key of the
    models' packed-weight caches, so the second evaluation once reused the first one's weights.)  Replay, eval, replay, eval:
    both evaluations equal the render of an eagerly trained twin at the same point, bit for bit."""
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.graphed import GraphedTrainStep
    N, ins_num = 64, 13
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(50.0, -65.0, 7.0))
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    g = torch.Generator().manual_seed(12)
    batches = [(torch.stack([ro[s:s + N], rd[s:s + N]]).cuda(), torch.rand(N, 3, generator=g).cuda(),
                torch.randint(0, 5, (N,), generator=g).cuda()) for s in (1000, 90000, 200000)]
    z = A.H.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05)
    eargs = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    eval_rays = torch.stack([ro[150000:150000 + N], rd[150000:150000 + N]]).cuda()

    def fresh():
        mc, mf = models(A)
        mc.train(); mf.train()
        opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=torch.tensor(5e-4, device="cuda"), capturable=True)
        return mc, mf, opt

    def evaluate(mc, mf):
        with torch.no_grad():
            out = A.R.dm_nerf(eval_rays, None, None, mc, mf, z, eargs)
        return {k: out[k].clone() for k in ("rgb_fine", "ins_fine", "depth_fine", "raw_fine")}

    mc, mf, opt = fresh()
    torch.cuda.manual_seed(5)
    want = []
    for rays, tgt, lab in batches[1:]:
        D.sharded_train_step(rays, z, tgt, lab, (mc, mf), args, opt, ins_num)
        want.append(evaluate(mc, mf))
    mc2, mf2, opt2 = fresh()
    gs = GraphedTrainStep((mc2, mf2), opt2, args, ins_num, *batches[0][:1], z, *batches[0][1:])
    torch.cuda.manual_seed(5)
    got = []
    for rays, tgt, lab in batches[1:]:
        gs.step(rays, z, tgt, lab)
        got.append(evaluate(mc2, mf2))
    for w, g_ in zip(want, got):
        for k in w:
            assert torch.equal(w[k], g_[k]), k
    assert not torch.equal(got[0]["raw_fine"], got[1]["raw_fine"])              # the second step did change what the networks return
    assert float(got[1]["rgb_fine"].std()) > 0
