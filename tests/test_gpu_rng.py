"""The IMPLICIT random draws of the path (SURVEY 8(b) "Device semantics"): what a caller who passes no ``t_rand`` / ``u`` gets
must be what the reference's own calls would consume from the device generator, in the reference's order --

* ``dm_nerf`` with ``args.perturb > 0``: ``torch.rand([N, 64])`` (networks/render.py:46), then ``torch.rand([N, 128])``
  (networks/helpers.py:135), inference and training alike;
* ``manipulator``: ``2 + T`` draws of ``torch.rand([N, N_importance])`` -- original, each target, original again
  (networks/manipulator.py:148,170,187: ``sample_pdf(det=False)`` even at evaluation);
* ``sharded_train_step``: the two tensors drawn FULL-size and sliced, so ray i's jitter does not depend on the world size.

Each test seeds the device generator, runs the call without draws, reseeds, makes the reference's draws by hand and runs the call
again with them passed in: every output is ``torch.equal`` -- no tolerance, the draws either are these tensors or they are not."""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
N, INS = 160, 13


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from dm_nerf_amd import _lib, distributed as D
    from dm_nerf_amd.networks import dm_nerf as M, helpers as H, manipulator as MA, render as R
    _lib.load()
    return types.SimpleNamespace(M=M, H=H, R=R, D=D, MA=MA)


def _models(A, train=False):
    out = []
    for seed in (81, 82):
        m = A.M.DM_NeRF(8, 256, 63, 27, [4], INS)
        m.load_state_dict(O.make_weights(seed, INS, gain=1.7, sigma_bias=0.3))
        m = m.cuda()
        out.append(m.train() if train else m.eval())
    return out


def _rays(start, n=N, theta=50.0):
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(theta, -65.0, 7.0))
    return torch.stack([ro.reshape(-1, 3)[start:start + n], rd.reshape(-1, 3)[start:start + n]]).cuda()


KEYS = ('rgb_fine', 'ins_fine', 'z_vals_fine', 'raw_fine', 'raw_coarse', 'rgb_coarse', 'ins_coarse', 'z_vals_coarse', 'depth_fine',
        'depth_coarse')


@pytest.mark.parametrize("training", [False, True])
def test_dm_nerf_draws_t_rand_then_u_from_the_device_generator(A, training):
    mc, mf = _models(A, training)
    rays = _rays(100000)
    z = A.H.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=training, N_ins=None)
    ctx = torch.enable_grad() if training else torch.no_grad()
    with ctx:
        torch.cuda.manual_seed(1234)
        got = A.R.dm_nerf(rays, None, None, mc, mf, z, args)
        after_implicit = torch.rand(4, device="cuda")                 # the generator advanced by exactly the two draws
        torch.cuda.manual_seed(1234)
        t_rand = torch.rand([N, 64], device="cuda")                   # render.py:46
        u = torch.rand([N, 128], device="cuda")                       # helpers.py:135
        after_explicit = torch.rand(4, device="cuda")
        want = A.R.dm_nerf(rays, None, None, mc, mf, z, args, t_rand=t_rand, u=u)
    assert training == bool(got['rgb_fine'].requires_grad)
    for k in KEYS:
        assert torch.equal(got[k], want[k]), k
    assert torch.equal(after_implicit, after_explicit)
    # the jitter really is in play (not the deterministic grid)
    assert not torch.equal(got['z_vals_coarse'], z)
    zc = O.dm_nerf(rays.cpu(), {k: v.detach().cpu() for k, v in mc.state_dict().items()}, {k: v.detach().cpu() for k, v in mf.state_dict().items()},
                   z.cpu(), perturb=1.0, t_rand=t_rand.cpu(), u=u.cpu())['z_vals_coarse']
    assert torch.equal(got['z_vals_coarse'].detach().cpu(), zc)         # == the oracle's stratification of the same t_rand


def test_manipulator_draws_two_plus_T_tensors_in_order(A):
    mc, mf = _models(A)
    n = 96
    ori = _rays(120000, n)
    tars = [_rays(120000, n, theta=54.0), _rays(120000, n, theta=58.0)]
    for T in (1, 2):
        args = types.SimpleNamespace(N_samples=64, N_importance=128, near=4.0, far=15.0, target_labels=list(range(1, T + 1)))
        with torch.no_grad():
            torch.cuda.manual_seed(77)
            got = A.MA.manipulator(None, None, mc, mf, ori, tars[:T], args)
            after_implicit = torch.rand(4, device="cuda")
            torch.cuda.manual_seed(77)
            us = [torch.rand([n, 128], device="cuda") for _ in range(2 + T)]     # original, each target, original again
            after_explicit = torch.rand(4, device="cuda")
            want = A.MA.manipulator(None, None, mc, mf, ori, tars[:T], args, us=us)
            wrong = A.MA.manipulator(None, None, mc, mf, ori, tars[:T], args, us=us[::-1])
        for g, w in zip(got, want):
            assert torch.equal(g, w)
        assert torch.equal(after_implicit, after_explicit)
        assert not torch.equal(got[0], wrong[0])                        # (the order of the draws matters)


def test_sharded_step_draws_full_size_then_slices(A):
    """One process: the step's implicit draws are ``torch.rand(z.shape)`` then ``torch.rand([N, N_importance])`` over the WHOLE
    batch (whatever the world size: tests/test_distributed_gloo.py checks that each rank then renders rows ``ray_slice`` of exactly
    these tensors); passing them in reproduces the update bit for bit."""
    rays = _rays(30000)
    z = A.H.z_val_sample(N, 4.0, 15.0, 64, device="cuda")
    g = torch.Generator().manual_seed(5)
    target, labels = torch.rand(N, 3, generator=g).cuda(), torch.randint(0, 6, (N,), generator=g).cuda()
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05)
    res = []
    for explicit in (False, True):
        mc, mf = _models(A, True)
        opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
        torch.cuda.manual_seed(99)
        kw = {}
        if explicit:
            kw = dict(t_rand=torch.rand(z.shape, device="cuda"), u=torch.rand([N, 128], device="cuda"))
        loss, _ = A.D.sharded_train_step(rays, z, target, labels, (mc, mf), args, opt, INS, **kw)
        res.append((loss.clone(), [p.detach().clone() for m in (mc, mf) for p in m.parameters()], torch.rand(4, device="cuda")))
    assert torch.equal(res[0][0], res[1][0])
    assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
    assert torch.equal(res[0][2], res[1][2])
