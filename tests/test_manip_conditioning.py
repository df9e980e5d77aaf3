"""CPU test of the acceptance rule the GPU suite holds the manipulation frame to (oracle/manip_margins.py::check_frame;
tests/test_gpu_manipulator_frame.py::test_against_the_reference_manipulator_eval_run): is the rule SOUND -- does it accept
independent f32-class evaluations of the reference's chain -- and does it have POWER -- does it reject a frame whose pixels were
mis-routed?  Everything here is the oracle (bit-equal to the reference's run, tests/golden/make_golden.py) and its re-evaluations."""
import pytest
import torch

from oracle import manip_margins as MM


@pytest.fixture(scope="module")
def frame(golden):
    g = golden("manipulator_frame")
    outs = {name: MM.run_variant(g, net) for name, net in MM.variant_nets().items()}
    critical, dist = MM.slope_critical(g)
    return g, outs, critical


def _sens(g, outs, names):
    ref = [g[k] for k in MM.OUTPUTS]
    return torch.stack([torch.stack([(outs[n][k] - ref[k]).abs().amax(-1) for n in names], 0).amax(0) for k in range(4)], 1)


def test_the_rule_accepts_every_variant_left_out_of_its_own_tolerance(frame):
    """Leave-one-out: each f32-class evaluation (network in float64; K summed in 2 .. 4 pieces) judged against a tolerance measured
    WITHOUT it must pass -- it stands in for the HIP path, which the tolerance is never measured with."""
    g, outs, critical = frame
    names = list(outs)
    assert 0.02 <= float(critical.float().mean()) <= 0.12          # ~10 % of the rays have one of their 384 draws on the slope threshold
    worst = {}
    for leave in names:
        rep = MM.check_frame(outs[leave], g, _sens(g, outs, [n for n in names if n != leave]), critical)
        assert rep["ok"], (leave, rep)
        worst[leave] = rep["worst_ratio_noncritical"]
    assert max(worst.values()) <= 2.0, worst                       # (measured: 0.8; the odd pixel at 1.2 with only three variants)
    full = _sens(g, outs, names)
    floor_frac = [float((1e-4 + 4.0 * full[:, k] <= 2e-4).float().mean()) for k in range(4)]
    assert min(floor_frac) >= 0.8 and floor_frac[2] == 1.0, floor_frac     # the tolerance IS 1e-4 .. 2e-4 on >= 80 % of the pixels


@pytest.mark.parametrize("frac", [0.2, 0.02])
def test_the_rule_rejects_mis_routed_pixels(frame, frac):
    """A frame driver that sends some of the pixels to the wrong place: the recorded frame with ``frac`` of its pixels replaced by the
    pixel ``shift`` rows further on (all four outputs).  20 % (the old 75 % / 10 % bounds passed that) and 2 %."""
    g, outs, critical = frame
    sens = _sens(g, outs, list(outs))
    ref = [g[k] for k in MM.OUTPUTS]
    n = ref[0].shape[0]
    assert MM.check_frame(ref, g, sens, critical)["ok"]
    gen = torch.Generator().manual_seed(3)
    pick = torch.randperm(n, generator=gen)[:max(4, int(frac * n))]
    bad = [t.clone() for t in ref]
    for t, r in zip(bad, ref):
        t[pick] = r[(pick + 3 * 20 + 7) % n]
    rep = MM.check_frame(bad, g, sens, critical)
    assert not rep["ok"] and rep["n_exceed"] >= 0.5 * len(pick), rep
