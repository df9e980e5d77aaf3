"""GPU tests (-m gpu) of dm_nerf_amd.optim.FlatAdam (extension; csrc/optim.hip): the reference's optimizer is
``torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr, betas=(0.9, 0.999))`` (train_dmsr.py:124-125) with
``param_group['lr']`` rewritten after every step (:68-72) -- that object is the yardstick here, not the oracle."""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
INS = 13


def models(seeds=(61, 62), ins_num=INS):
    from dm_nerf_amd.networks import dm_nerf as M
    out = []
    for seed in seeds:
        m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num)
        m.load_state_dict(O.make_weights(seed, ins_num, gain=1.7, sigma_bias=0.3))
        out.append(m.cuda().train())
    return out


def test_flat_adam_follows_torch_adam_on_identical_gradients(capsys):
    """20 steps on the same synthetic gradient sequence (magnitudes spread over six decades, some exact zeros), the learning rate
    decayed after every step the way train_dmsr.py:68-72 does it: every parameter within 2 ulp of torch.optim.Adam's, the moments
    too; the parameters keep their identity (same nn.Parameter objects, state_dict keys, values) when the optimizer adopts them."""
    from dm_nerf_amd.optim import FlatAdam
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    ref_models, own_models = models(), models()
    before = [p.detach().clone() for m in own_models for p in m.parameters()]
    ids = [id(p) for m in own_models for p in m.parameters()]
    ref = torch.optim.Adam([p for m in ref_models for p in m.parameters()], lr=5e-4, betas=(0.9, 0.999))
    own = FlatAdam(own_models, lr=5e-4, betas=(0.9, 0.999))
    assert [id(p) for m in own_models for p in m.parameters()] == ids
    assert all(torch.equal(a, p) for a, p in zip(before, [p for m in own_models for p in m.parameters()]))
    assert all(p.data_ptr() == own.flat.data_ptr() + 4 * o for p, o in zip(own.params, np.cumsum([0] + [p.numel() for p in own.params[:-1]])))
    g = torch.Generator(device="cuda").manual_seed(3)
    gmax = [0.0] * len(own.params)
    for it in range(1, 21):
        ref.zero_grad(); own.zero_grad()
        for k, (pr, po) in enumerate(zip([p for m in ref_models for p in m.parameters()], own.params)):
            gr = torch.randn(pr.shape, device="cuda", generator=g) * 10.0 ** torch.randint(-6, 1, (1,), device="cuda", generator=g).float()
            gr[torch.rand(pr.shape, device="cuda", generator=g) < 0.05] = 0.0
            gmax[k] = max(gmax[k], float(gr.abs().max()))
            pr.grad, po.grad = gr, gr.clone()                            # (foreign gradient tensors: gathered into the arena)
        ref.step(); own.step()
        lr = 5e-4 * (0.1 ** (it / 500000.0))
        for grp in ref.param_groups:
            grp['lr'] = lr
        for grp in own.param_groups:
            grp['lr'] = lr
    torch.cuda.synchronize()
    # Every operation of the update is the same f32 operation in both, in the same order, with the same three fused multiply-adds
    # ATen's device kernels contract to (csrc/optim.hip) -- observed on MI355X: bit-equal.  The bounds allow for a build of torch
    # that contracts differently (half an ulp of the accumulated quantity per step, amplified where a moment passes through
    # zero): parameters 2 ulp + 2e-8, first moment 2 ulp + 2 ulp of the largest gradient the tensor saw, second moment 8 ulp.
    exact = {"p": True, "m": True, "v": True}
    worst = {"p": 0.0, "m": 0.0, "v": 0.0}
    o = 0
    for pr, po, gm in zip([p for m in ref_models for p in m.parameters()], own.params, gmax):
        n = pr.numel()
        st = ref.state[pr]
        dp = (pr.detach() - po.detach()).abs()
        assert bool((dp <= 2.4e-7 * pr.detach().abs() + 2e-8).all()), float(dp.max())
        dm = (st["exp_avg"].reshape(-1) - own.exp_avg[o:o + n]).abs()
        assert bool((dm <= 2.4e-7 * st["exp_avg"].reshape(-1).abs() + 2.4e-7 * gm).all()), (float(dm.max()), gm)
        dv = (st["exp_avg_sq"].reshape(-1) - own.exp_avg_sq[o:o + n]).abs()
        assert bool((dv <= 1e-6 * st["exp_avg_sq"].reshape(-1).abs() + 1e-37).all()), float(dv.max())
        exact = {"p": exact["p"] and float(dp.max()) == 0.0, "m": exact["m"] and float(dm.max()) == 0.0, "v": exact["v"] and float(dv.max()) == 0.0}
        worst["p"] = max(worst["p"], float((dp / (pr.detach().abs() + 1e-3)).max()))
        worst["m"] = max(worst["m"], float(dm.max()) / gm)
        worst["v"] = max(worst["v"], float((dv / (st["exp_avg_sq"].reshape(-1).abs() + 1e-30)).max()))
        o += n
    moved = max(float((a - p.detach()).abs().max()) for a, p in zip(before, own.params))
    with capsys.disabled():
        print(f"\n[FlatAdam vs torch.optim.Adam, 20 steps] worst |dp| / (|p| + 1e-3) {worst['p']:.2e}, |d exp_avg| / max|g| {worst['m']:.2e}, "
              f"|d exp_avg_sq| / exp_avg_sq {worst['v']:.2e}; bit-equal: {exact}; parameters moved by up to {moved:.2e}; step count {int(own.state2[0])}")
    assert int(own.state2[0]) == 20 and int(own.state2[1]) == 0
    assert moved > 5e-3


def test_repack_equals_the_separate_pack_kernels_bit_for_bit():
    """What FlatAdam installs after a step -- flat copy, forward blob, W^T blob incl. the head product -- equals what the models
    pack for themselves from the same parameters (dmnerf_head_product + dmnerf_pack_weights), and they are NEW tensors per step."""
    from dm_nerf_amd import weights
    from dm_nerf_amd.optim import FlatAdam
    ms = models(ins_num=13) + models(seeds=(63,), ins_num=59)               # two widths in one launch
    own = FlatAdam(ms[:2])
    wide = FlatAdam(ms[2:])
    for opt, group in ((own, ms[:2]), (wide, ms[2:])):
        own_flat0 = group[0].flat()
        opt.zero_grad()
        for p in opt.params:
            p.grad = torch.randn_like(p) * 1e-2
        opt.step()
        for m in group:
            state = dict(m.named_parameters())
            flat = weights.flat_params(state)
            assert torch.equal(m.flat(), flat) and m.flat().data_ptr() != flat.data_ptr()
            assert torch.equal(m.blob(), weights.pack_blob(state, m.ins_num, flat=flat))
            assert torch.equal(m.blob_t(), weights.pack_blob(state, m.ins_num, transposed=True, flat=flat))
        assert group[0].flat().data_ptr() != own_flat0.data_ptr() and not torch.equal(group[0].flat(), own_flat0)


def test_state_dict_round_trips_with_torch_adam():
    """Checkpoints (train_dmsr.py:78-86 saves optimizer.state_dict()): FlatAdam -> torch.optim.Adam -> continue, and back."""
    from dm_nerf_amd.optim import FlatAdam
    a_models, b_models = models(), models()
    own = FlatAdam(a_models, lr=5e-4)
    g = torch.Generator(device="cuda").manual_seed(8)
    grads = [[torch.randn(p.shape, device="cuda", generator=g) * 1e-2 for p in own.params] for _ in range(6)]

    def run(opt, params, gs):
        for gset in gs:
            opt.zero_grad()
            for p, gr in zip(params, gset):
                p.grad = gr.clone()
            opt.step()
    run(own, own.params, grads[:3])
    sd = own.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == 60 and float(sd["state"][0]["step"]) == 3.0
    b_params = [p for m in b_models for p in m.parameters()]
    with torch.no_grad():
        for pb, pa in zip(b_params, own.params):
            pb.copy_(pa)
    ref = torch.optim.Adam(b_params, lr=5e-4)
    ref.load_state_dict(sd)
    run(ref, b_params, grads[3:])
    run(own, own.params, grads[3:])
    assert all(bool(((pb.detach() - pa.detach()).abs() <= 2.4e-7 * pb.detach().abs() + 1e-9).all()) for pb, pa in zip(b_params, own.params))
    # and back: torch's state into a fresh FlatAdam
    c_models = models()
    with torch.no_grad():
        for pc, pb in zip([p for m in c_models for p in m.parameters()], b_params):
            pc.copy_(pb)
    own2 = FlatAdam(c_models, lr=1e-3)
    own2.load_state_dict(ref.state_dict())
    assert int(own2.state2[0]) == 6 and own2.param_groups[0]["lr"] == 5e-4
    extra = [torch.randn(p.shape, device="cuda", generator=g) * 1e-2 for p in b_params]
    run(ref, b_params, [extra]); run(own2, own2.params, [extra])
    assert all(bool(((pb.detach() - pc.detach()).abs() <= 2.4e-7 * pb.detach().abs() + 1e-9).all()) for pb, pc in zip(b_params, own2.params))


def _batches(N):
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(50.0, -65.0, 7.0))
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    g = torch.Generator().manual_seed(11)
    return [(torch.stack([ro[s:s + N], rd[s:s + N]]).cuda(), torch.rand(N, 3, generator=g).cuda(), torch.randint(0, 5, (N,), generator=g).cuda())
            for s in (1000, 90000, 200000, 5000)]


ARGS = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05)


def test_training_steps_with_flat_adam_match_torch_adam():
    """Three sharded_train_steps: the first step's gradients are BIT-equal to the torch.optim.Adam run's (same kernels, the
    arena is only where they land), losses and parameters follow to float noise."""
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.networks import helpers as Hh
    from dm_nerf_amd.optim import FlatAdam
    N = 96
    batches = _batches(N)
    z = Hh.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    r_models, o_models = models(), models()
    ref = torch.optim.Adam([p for m in r_models for p in m.parameters()], lr=5e-4)
    own = FlatAdam(o_models, lr=5e-4)
    losses = [[], []]
    for k, (ms, opt) in enumerate(((r_models, ref), (o_models, own))):
        torch.cuda.manual_seed(123)
        for i, (rays, tgt, lab) in enumerate(batches[1:]):
            losses[k].append(float(D.sharded_train_step(rays, z, tgt, lab, ms, ARGS, opt, INS)[0]))
            if i == 0:
                first = [p.grad.clone() for m in ms for p in m.parameters()]
                if k == 0:
                    ref_first = first
                else:
                    assert all(torch.equal(a, b) for a, b in zip(ref_first, first))
                    assert own.arena.resident()
    assert abs(losses[0][0] - losses[1][0]) == 0.0
    assert np.allclose(losses[0], losses[1], rtol=2e-6)
    worst = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip([p for m in r_models for p in m.parameters()], own.params))
    assert worst <= 2e-6, worst


def test_graphed_step_with_flat_adam_equals_eager_flat_adam():
    """GraphedTrainStep with FlatAdam(capturable=True): update + re-pack are captured, the re-pack writes IN PLACE into buffers the
    next replay's forward reads; parameters after three replays on new batches are bit-equal to three eager FlatAdam steps, an
    eager render between replays sees the current weights, and set_lr reaches the captured update."""
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.graphed import GraphedTrainStep
    from dm_nerf_amd.networks import helpers as Hh, render as R
    from dm_nerf_amd.optim import FlatAdam
    N = 96
    batches = _batches(N)
    z = Hh.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    eargs = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)

    def evaluate(ms):
        with torch.no_grad():
            return R.dm_nerf(batches[0][0], None, None, ms[0], ms[1], z, eargs)["rgb_fine"].clone()
    e_models = models()
    eager = FlatAdam(e_models, lr=5e-4, capturable=True)
    torch.cuda.manual_seed(123)
    e_losses, e_evals = [], []
    for i, (rays, tgt, lab) in enumerate(batches[1:]):
        if i == 2:
            eager.set_lr(1e-4)
        e_losses.append(D.sharded_train_step(rays, z, tgt, lab, e_models, ARGS, eager, INS)[0].clone())
        e_evals.append(evaluate(e_models))
    g_models = models()
    start = [p.detach().clone() for m in g_models for p in m.parameters()]
    opt = FlatAdam(g_models, lr=5e-4, capturable=True)
    gs = GraphedTrainStep(g_models, opt, ARGS, INS, batches[0][0], z, batches[0][1], batches[0][2])
    assert all(torch.equal(a, p) for a, p in zip(start, opt.params)), "warm-up was not undone"
    assert int(opt.state2[0]) == 0 and float(opt.exp_avg.abs().max()) == 0.0
    torch.cuda.manual_seed(123)
    g_losses, g_evals = [], []
    for i, (rays, tgt, lab) in enumerate(batches[1:]):
        if i == 2:
            gs.set_lr(1e-4)
        g_losses.append(gs.step(rays, z, tgt, lab).clone())
        g_evals.append(evaluate(g_models))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(e_losses, g_losses)), (e_losses, g_losses)
    assert all(torch.equal(a, b) for a, b in zip(e_evals, g_evals))
    assert all(torch.equal(a.detach(), b.detach()) for a, b in zip(eager.params, opt.params))
    assert int(opt.state2[0]) == 3 and not torch.equal(e_evals[0], e_evals[1])


def test_flat_adam_refuses_a_step_with_missing_gradients():
    """torch.optim.Adam skips parameters whose ``.grad`` is None; one pass over flat vectors cannot -- FlatAdam raises instead of
    updating them with a zero or stale gradient (e.g. a step that differentiated only the coarse model)."""
    from dm_nerf_amd.optim import FlatAdam
    ms = models()
    own = FlatAdam(ms)
    own.zero_grad()
    for p in ms[0].parameters():
        p.grad = torch.zeros_like(p)
    with pytest.raises(RuntimeError, match=r"models\[1\]\.mlps\.0\.weight has no gradient"):
        own.step()
    assert int(own.state2[0]) == 0


def _plain_loss(ms, rays, z, tgt, lab):
    """The loss of train_dmsr.py:35-49 through the drop-in functions (every parameter of both models gets a gradient)."""
    from dm_nerf_amd.networks import evaluator as E, render as R
    out = R.dm_nerf(rays, None, None, ms[0], ms[1], z, ARGS)
    return E.img2mse(out['rgb_fine'], tgt) + E.img2mse(out['rgb_coarse'], tgt) \
        + E.ins_criterion(out['ins_fine'], lab, INS)[0] + E.ins_criterion(out['ins_coarse'], lab, INS)[0]


def test_zero_grad_keeping_the_tensors_accumulates_once_not_twice():
    """``zero_grad(set_to_none=False)`` keeps ``p.grad`` = views of the gradient arena and zeroes them.  Had the
    arena then handed its slots out again, the backward would have written the new gradient INTO those views and AccumulateGrad
    would have added the memory onto itself (2 g).  Three steps of the plain loop ``zero_grad(set_to_none=False); backward; step``
    against torch.optim.Adam on twin models: gradients bit-equal at every step, parameters to float noise -- and a second backward
    without zero_grad in between accumulates (g1 + g2), as it does for torch tensors."""
    from dm_nerf_amd.networks import helpers as Hh
    from dm_nerf_amd.optim import FlatAdam
    N = 96
    batches = _batches(N)
    z = Hh.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    r_models, o_models = models(), models()
    r_params = [p for m in r_models for p in m.parameters()]
    ref = torch.optim.Adam(r_params, lr=5e-4)
    own = FlatAdam(o_models, lr=5e-4)
    for step, (rays, tgt, lab) in enumerate(batches[:3]):
        grads = []
        for ms, opt in ((r_models, ref), (o_models, own)):
            torch.cuda.manual_seed(40 + step)
            opt.zero_grad(set_to_none=False)
            _plain_loss(ms, rays, z, tgt, lab).backward()
            grads.append([p.grad.clone() for m in ms for p in m.parameters()])
            opt.step()
        # step 0: no tensors to keep yet (the arena slots are written in place); steps 1, 2: the kept views are zeroed and ADDED into
        worst = max(float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in zip(*grads))
        assert worst <= (0.0 if step == 0 else 1e-4), (step, worst)     # (later steps: the runs' parameters differ by float noise; 2 g would read 1.0)
        assert min(float(g.abs().max()) for g in grads[1]) > 0.0
        assert own.arena.resident()                                    # the kept tensors ARE the arena views
    worst = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(r_params, own.params))
    assert worst <= 2e-6, worst                                        # (2 g instead of g would not show here -- Adam is scale-invariant -- the
    #                                                                    gradient comparison above is the check; this one says the runs stayed twins)
    # accumulation across two backward passes: g(batch 0) + g(batch 1), in both worlds
    sums = []
    for ms, opt in ((r_models, ref), (o_models, own)):
        opt.zero_grad(set_to_none=False)
        for k in (0, 1):
            torch.cuda.manual_seed(50 + k)
            _plain_loss(ms, batches[k][0], z, batches[k][1], batches[k][2]).backward()
        sums.append([p.grad.clone() for m in ms for p in m.parameters()])
    assert all(bool(((a - b).abs() <= 1e-4 * a.abs().max() + 1e-12).all()) for a, b in zip(*sums))


def test_a_replaced_middle_gradient_is_not_mistaken_for_the_arena():
    """the arena check looks at ALL views.  A caller that replaces the gradient of one tensor in the MIDDLE of a model
    out of place (clipping / scaling one layer) gets that tensor's new values in the update, not the arena's stale ones."""
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.networks import helpers as Hh
    from dm_nerf_amd.optim import FlatAdam
    N = 96
    rays, tgt, lab = _batches(N)[0]
    z = Hh.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    ms = models()
    own = FlatAdam(ms, lr=5e-4)

    class NoStep:                                                    # run the step's forward / backward, keep the update for below
        wants_arena = True
        zero_grad = staticmethod(lambda set_to_none=True: own.zero_grad(set_to_none))
        step = staticmethod(lambda: None)
    torch.cuda.manual_seed(5)
    D.sharded_train_step(rays, z, tgt, lab, ms, ARGS, NoStep, INS)
    assert own.arena.resident()
    mid = ms[0].mlps[3].weight                                       # neither first nor last of its model
    before = mid.detach().clone()
    mid.grad = mid.grad * 0.0                                        # out of place: a NEW tensor, the arena still holds the old values
    assert not own.arena.resident()
    own.step()
    torch.cuda.synchronize()
    assert torch.equal(mid.detach(), before)                         # Adam with g = 0 on fresh moments: no movement at all
    assert not torch.equal(ms[0].mlps[2].weight.detach(), O.make_weights(61, INS, gain=1.7, sigma_bias=0.3)["mlps.2.weight"].cuda())
