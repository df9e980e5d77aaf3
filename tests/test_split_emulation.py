"""CPU test: the ARITHMETIC of the opt-in split-operand modes, emulated through the whole MLP (scripts/split_emulate.py: fused-heads
form, products of 16-bit planes exact in f32, f32 accumulation -- what the MFMA does) against a float64 evaluation.

Pins the two facts the f16x2 kernels are built on:
  * x ~ hi + lo with hi = f16(x) (round-toward-zero, v_cvt_pkrtz_f16_f32), lo = f16(x - hi), and only THREE products hi.hi + hi.lo +
    lo.hi is f32-class -- within a small factor of a plain f32 evaluation of the same network, on default-init and PEAKY weights;
  * it needs the f16 SUBNORMAL lo planes (weights ~ U(-1/16, 1/16): every lo is below 2^-14): an MFMA that flushed them would be two
    to three decades worse (the GPU side of this is scripts/micro/mfma_f16.hip: v_mfma_f32_32x32x16_f16 does not flush)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import split_emulate as E                                   # noqa: E402
from dm_nerf_amd import weights                             # noqa: E402
from oracle import ref_cpu as O                             # noqa: E402


@pytest.mark.parametrize("name,kw,near,far", [("plain ins59", dict(seed=6, ins_num=59, gain=1.7, sigma_bias=0.3), 0.0, 4.7),
                                               ("PEAKY ins13", dict(seed=5, ins_num=13, **O.PEAKY), 4.0, 15.0)])
def test_three_products_of_two_f16_planes_are_f32_class(name, kw, near, far):
    pts, vd = E.inputs(512, near, far)
    sdf = dict(O.fuse_heads(O.make_weights(**kw)))
    xp, xv = O.embed(pts, 10), O.embed(vd, 4)
    ref = E.network({k: v.double() for k, v in sdf.items()}, xp.double(), xv.double(), E.Scheme("f64"))
    err = {}
    for key, S in (("f32", E.Scheme("f32")), ("bf16x3", E.Scheme("bf16x3")), ("f16x2", E.Scheme("f16x2", "rtz")),
                   ("f16x2 flushed", E.Scheme("f16x2", "rtz", flush=True))):
        got = E.network(sdf, xp, xv, S).double()
        err[key] = float(((got - ref).abs() / (1 + ref.abs())).max())
        if key != "f16x2 flushed":
            assert int((got[:, 4:].argmax(-1) != ref[:, 4:].argmax(-1)).sum()) == 0, (name, key)
    assert err["f16x2"] <= 4 * err["f32"] + 1e-7 and err["bf16x3"] <= 2 * err["f32"] + 1e-7, (name, err)
    assert err["f16x2"] <= 1e-5 * (1 if "plain" in name else 10), (name, err)          # the parity bar itself on plain weights
    assert err["f16x2 flushed"] >= 50 * err["f16x2"], (name, err)
