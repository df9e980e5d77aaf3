"""Object-code loss ``ins_criterion`` (networks/evaluator.py:19-74) on the device vs the reference.

The fixtures hold the reference's own outputs (scipy 1.15 assignment).  Tolerances: the loss terms are means of
f32 logs over N rays; the reference sums them in f32 (pairwise), the kernels in f32 per 64 rays and f64 across --
rel 2e-6 on the values, 2e-5 of the gradient's largest entry on the gradient; the assignment itself must be
identical (the cost gaps of the fixtures are far above rounding).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from dm_nerf_amd.networks import evaluator
    return evaluator


def dev(t):
    return t.cuda()


@pytest.mark.parametrize("name", ["all", "some", "wide"])
def test_ins_criterion_golden(E, golden, name):
    g = golden("ins_criterion")
    ins_num = int(g[f"{name}_ins_num"])
    pred = dev(g[f"{name}_pred"]).requires_grad_(True)
    out = E.ins_criterion(pred, dev(g[f"{name}_lab"]), ins_num)
    want = g[f"{name}_out"]
    for got, w, what in zip(out, want, ("loss", "valid_ce", "invalid_ce", "valid_siou")):
        assert got.dim() == 0
        assert abs(float(got.detach()) - float(w)) <= 2e-6 * max(1.0, abs(float(w))), (name, what, float(got.detach()), float(w))
    out[0].backward()
    gw = g[f"{name}_grad"]
    err = float((pred.grad.cpu() - gw).abs().max())
    assert err <= 2e-5 * float(gw.abs().max()), (name, err, float(gw.abs().max()))
    # unmatched channels all receive the same constant, matched ones do not
    cols = g[f"{name}_cols"].numpy()
    n_valid = g[f"{name}_cost_ce"].shape[0]
    if len(cols) > n_valid:
        un = pred.grad[:, torch.from_numpy(cols[n_valid:]).cuda()]
        assert float(un.max() - un.min()) == 0.0


def test_ins_criterion_each_output_differentiable(E, golden):
    """valid_ce / invalid_ce / valid_siou are outputs of their own: their gradients add up to the loss gradient."""
    g = golden("ins_criterion")
    ins_num = int(g["some_ins_num"])
    grads = []
    for k in range(4):
        pred = dev(g["some_pred"]).requires_grad_(True)
        E.ins_criterion(pred, dev(g["some_lab"]), ins_num)[k].backward()
        grads.append(pred.grad.clone())
    assert torch.allclose(grads[0], grads[1] + grads[2] + grads[3], rtol=1e-6, atol=1e-9)


def test_ins_criterion_vs_oracle_bench_size(E):
    """4096 rays, ins_num 13 (the bench's training batch): device loss vs the oracle (scipy assignment)."""
    from oracle import ref_cpu as O
    gen = torch.Generator().manual_seed(77)
    N, ins_num = 4096, 13
    lab = torch.randint(0, 9, (N,), generator=gen)
    pred = torch.sigmoid(torch.randn(N, ins_num, generator=gen) + 2.0 * torch.nn.functional.one_hot((lab * 5 + 3) % ins_num, ins_num))
    po = pred.clone().requires_grad_(True)
    want = O.ins_criterion(po, lab, ins_num)
    want[0].sum().backward()
    pg = pred.cuda().requires_grad_(True)
    got = E.ins_criterion(pg, lab.cuda(), ins_num)
    got[0].backward()
    for a, b in zip(got, want):
        assert abs(float(a.detach()) - float(b.detach().reshape(-1)[0])) <= 3e-6 * max(1.0, abs(float(b.detach().reshape(-1)[0])))
    assert float((pg.grad.cpu() - po.grad).abs().max()) <= 2e-5 * float(po.grad.abs().max())


def test_ins_criterion_errors_are_loud(E):
    with pytest.raises(RuntimeError):
        E.ins_criterion(torch.rand(8, 13), torch.zeros(8, dtype=torch.int64), 13)           # CPU tensors
    with pytest.raises(ValueError):
        E.ins_criterion(torch.rand(8, 12, device="cuda"), torch.zeros(8, dtype=torch.int64, device="cuda"), 13)


def test_label_conditions_the_reference_raises_on_are_flagged(E):
    """Where the reference raises (more distinct labels than channels: the one-hot column assignment of evaluator.py:24
    fails; a label outside [0, ins_num]: F.one_hot / the indexing of :23 fails) the stream-resident loss cannot -- it
    records the condition, and ``check=True`` turns it into the reference's behaviour at the price of one sync."""
    g = torch.Generator().manual_seed(5)
    ins_num, N = 5, 300
    pred = torch.sigmoid(torch.randn(N, ins_num, generator=g)).cuda()
    ok = torch.randint(0, ins_num, (N,), generator=g).cuda()
    out = E.ins_criterion(pred, ok, ins_num, check=True)                       # clean batch: no flag, no raise
    assert bool(torch.isfinite(out[0]))
    many = torch.arange(N).cuda() % (ins_num + 1)                              # 6 distinct labels 0..5 for 5 channels
    E.ins_criterion(pred, many, ins_num)                                       # default: keeps the first ins_num, no sync
    with pytest.raises(ValueError, match="distinct labels"):
        E.ins_criterion(pred, many, ins_num, check=True)
    bad = ok.clone(); bad[17] = ins_num + 3; bad[40] = -1
    with pytest.raises(ValueError, match="outside"):
        E.ins_criterion(pred, bad, ins_num, check=True)
    # the flagged rays joined no row: same result as dropping... their label rows only (they still count in the sums over all rays)
    assert bool(torch.isfinite(E.ins_criterion(pred, bad, ins_num)[0]))
