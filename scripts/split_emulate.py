"""Build container (CPU, no GPU): accuracy of operand-splitting schemes for the opt-in MFMA modes, emulated through the
whole DM-NeRF MLP (fused-heads form, the function the split kernels evaluate) against a float64 evaluation.

    python scripts/split_emulate.py [--rows 4096] [--json out.json]

Schemes (every product of two 16-bit operands is exact in f32; accumulation in f32, as the MFMA does):
  f32        plain f32 GEMMs (the default kernels' class)
  bf16x3     x = hi + mid + lo by truncation, six products           (mlp_split_impl.h, round 1/2)
  f16x2      x ~ hi + lo, hi = f16(x), lo = f16(x - hi); products hi.hi + hi.lo + lo.hi (THREE MFMAs per f32 product)
             variants: rounding of the activation split (rtz = v_cvt_pkrtz_f16_f32 | rtn), weights always rtn (pre-split
             offline), f16 subnormals kept or flushed (what an MFMA that flushes f16 denormals would see), and a
             power-of-two scale 2^a carried by every activation (mlps.0 weights and all biases scaled by 2^a, exact;
             ReLU is positively homogeneous; outputs unscaled at the end) to lift the lo planes out of the subnormal range.
Weights: default-init class (oracle.make_weights gain 1.7 / sigma_bias 0.3), PEAKY ("trained-like"), ins_num 13 / 59 / 93.
Reports max / mean of |raw - raw64| / (1 + |raw64|), and the largest |activation| seen (f16 range check: 65 504 / 2^a)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_cpu as O  # noqa: E402  (scripts/ are diagnostics, not the product path)


def rtz_f16(x):
    """f32 -> f16 round toward zero (v_cvt_pkrtz_f16_f32), returned as f32; overflow saturates at 65504."""
    h = x.to(torch.float16)                                   # rtn
    hf = h.float()
    over = hf.abs() > x.abs()                                 # rounded away from zero: step one ulp back
    hb = h.view(torch.int16)
    hb = torch.where(over, hb - 1, hb)                        # sign-magnitude: magnitude - 1 (never crosses zero when over)
    out = hb.view(torch.float16).float()
    out = torch.where(torch.isinf(out), torch.sign(x) * 65504.0, out)
    return out


def rtn_f16(x):
    return x.to(torch.float16).float()


def flush_sub(h):
    """f16 values (held in f32) with subnormals flushed to zero."""
    return torch.where(h.abs() < 2.0 ** -14, torch.zeros_like(h), h)


def split_f16(x, rnd, flush):
    cv = rtz_f16 if rnd == "rtz" else rtn_f16
    hi = cv(x)
    lo = cv(x - hi)
    if flush:
        hi, lo = flush_sub(hi), flush_sub(lo)
    return hi, lo


def split_bf16(x):
    def trunc(v):
        return (v.view(torch.int32) & -65536).view(torch.float32)
    hi = trunc(x); r = x - hi
    mid = trunc(r); lo = r - mid
    return hi, mid, trunc(lo)


def mm(a, b):
    """f32 GEMM a [M, K] x b [N, K]^T with f32 accumulation."""
    return a @ b.t()


class Scheme:
    def __init__(self, kind, rnd="rtz", flush=False, act_scale=0):
        self.kind, self.rnd, self.flush, self.a = kind, rnd, flush, act_scale
        self.maxact = 0.0

    def name(self):
        if self.kind != "f16x2":
            return self.kind
        return f"f16x2 act-{self.rnd}" + (" flush-subnormals" if self.flush else "") + (f" act-scale 2^{self.a}" if self.a else "")

    def linear(self, x, W, b, first=False, last_scale=False):
        """y = x W^T + b with this scheme's GEMM; x carries the activation scale (f16x2), so does y."""
        if self.kind == "f64":
            return x @ W.double().t() + b.double()
        if self.kind == "f32":
            return mm(x, W) + b
        if self.kind == "bf16x3":
            xh, xm, xl = split_bf16(x); wh, wm, wl = split_bf16(W)
            y = mm(xh, wh) + mm(xm, wh) + mm(xh, wm) + mm(xl, wh) + mm(xm, wm) + mm(xh, wl)
            return y + b
        sc = np.float32(2.0 ** self.a)
        Ws = W * sc if first else W
        self.maxact = max(self.maxact, float(x.abs().max()))
        xh, xl = split_f16(x, self.rnd, self.flush)
        wh, wl = split_f16(Ws, "rtn", self.flush)
        return mm(xh, wh) + (mm(xl, wh) + mm(xh, wl)) + b * sc


def network(sd, xp, xv, S):
    """Fused-heads form of DM_NeRF.forward (dm_nerf.py:80-106; weights.py::fuse_heads), heads on the VALU in f32."""
    relu = torch.relu
    h = relu(S.linear(xp, sd["mlps.0.weight"], sd["mlps.0.bias"], first=True))
    sc = (2.0 ** S.a) if S.kind == "f16x2" else 1.0
    for i in range(1, 8):
        W = sd[f"mlps.{i}.weight"]
        if i == 5:
            # [h, pts]: the pts columns see the UNSCALED encoding; their weights carry the activation scale
            y = S.linear(h, W[:, :256], sd[f"mlps.{i}.bias"]) + (S.linear(xp, W[:, 256:], torch.zeros(256, dtype=W.dtype), first=True))
        else:
            y = S.linear(h, W, sd[f"mlps.{i}.bias"])
        h = relu(y)
    f64 = S.kind == "f64"
    cast = (lambda t: t.double()) if f64 else (lambda t: t)
    inv = 1.0 / sc
    den = (h @ cast(sd["density_linear.weight"]).t()) * inv + cast(sd["density_linear.bias"])
    Wr = sd["rgb_feature_linears.0.weight"]
    g1 = relu(S.linear(h, Wr[:, :256], sd["rgb_feature_linears.0.bias"]) + S.linear(xv, Wr[:, 256:], torch.zeros(128, dtype=Wr.dtype), first=True))
    rgb = (g1 @ cast(sd["rgb_linear.weight"]).t()) * inv + cast(sd["rgb_linear.bias"])
    g2 = relu(S.linear(h, sd["ins_feature_linears.0.weight"], sd["ins_feature_linears.0.bias"]))
    ins = S.linear(g2, sd["ins_linear.weight"], sd["ins_linear.bias"]) * inv
    return torch.cat([rgb, den, ins], -1)


def inputs(rows, near, far, seed=0):
    K = O.dmsr_intrinsics(480, 640)
    ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(30.0, -65.0, 7.0))
    g = torch.Generator().manual_seed(seed)
    sel = torch.randperm(480 * 640, generator=g)[:rows // 64]
    ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
    z = O.z_val_sample(ro.shape[0], near, far, 64)
    pts = (ro[:, None] + rd[:, None] * z[..., None]).reshape(-1, 3)
    vd = (rd / rd.norm(dim=-1, keepdim=True))[:, None].expand(-1, 64, -1).reshape(-1, 3)
    return pts, vd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    schemes = [lambda: Scheme("f32"), lambda: Scheme("bf16x3"),
               lambda: Scheme("f16x2", "rtz"), lambda: Scheme("f16x2", "rtn"), lambda: Scheme("f16x2", "rtz", flush=True),
               lambda: Scheme("f16x2", "rtz", act_scale=3), lambda: Scheme("f16x2", "rtz", act_scale=5),
               lambda: Scheme("f16x2", "rtz", flush=True, act_scale=5)]
    cases = [("default-init ins13", dict(seed=5, ins_num=13, gain=1.7, sigma_bias=0.3), 4.0, 15.0),
             ("PEAKY ins13", dict(seed=5, ins_num=13, **O.PEAKY), 4.0, 15.0),
             ("default-init ins59 (Replica near/far)", dict(seed=6, ins_num=59, gain=1.7, sigma_bias=0.3), 0.0, 4.7),
             ("PEAKY ins93", dict(seed=7, ins_num=93, **O.PEAKY), 0.0, 4.7)]
    out = {}
    for cname, kw, near, far in cases:
        pts, vd = inputs(a.rows, near, far)
        sd = O.make_weights(**kw)
        sdf = {k: v for k, v in __import__("oracle.ref_cpu", fromlist=["fuse_heads"]).fuse_heads(sd).items()}
        xp, xv = O.embed(pts, 10), O.embed(vd, 4)
        ref = network({k: v.double() for k, v in sdf.items()}, xp.double(), xv.double(), Scheme("f64"))
        print(f"== {cname}: {xp.shape[0]} samples, |raw64| max {float(ref.abs().max()):.1f}")
        out[cname] = {}
        for mk in schemes:
            S = mk()
            got = network(sdf, xp, xv, S).double()
            e = (got - ref).abs() / (1 + ref.abs())
            flips = int((got[:, 4:].argmax(-1) != ref[:, 4:].argmax(-1)).sum())
            print(f"  {S.name():44s} max {float(e.max()):.2e}  mean {float(e.mean()):.2e}  per-sample logit-argmax flips {flips}"
                  + (f"  max|act| {S.maxact:.1f} (x2^{S.a} carried)" if S.kind == 'f16x2' else ""))
            out[cname][S.name()] = {"max": float(e.max()), "mean": float(e.mean()), "argmax_flips": flips, "max_act": S.maxact}
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
