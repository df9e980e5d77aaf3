"""GPU box: accuracy of the inference kernels against a float64 evaluation of the same network, and their launch times.

layerwise f32 MFMA (default) | fused heads f32 MFMA | fused heads + split-bf16 MFMA (bf16x3) | fused heads + split-f16 MFMA
(f16x2) -- max and mean of |raw - raw64| / (1 + |raw64|) over 4096 x 64 samples of the bench camera, ins_num = 13 (or argv[1]);
then HIP-event times of each kernel on the fine-network launch (4096 x 192 samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dm_nerf_amd import _lib
from dm_nerf_amd.networks import dm_nerf as M, helpers as H, render as R
from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical

dev = torch.device("cuda:0")
torch.manual_seed(0)
ins_num = int(sys.argv[1]) if len(sys.argv) > 1 else 13
m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(dev)
with torch.no_grad():
    m.density_linear.bias.add_(0.3)
K = dmsr_intrinsics(480, 640)
ro, rd = H.get_rays_k(480, 640, K, pose_spherical(30.0, -65.0, 7.0).to(dev))
ro, rd = ro.reshape(-1, 3)[:4096].contiguous(), rd.reshape(-1, 3)[:4096].contiguous()
z = H.z_val_sample(4096, 4.0, 15.0, 64, device=dev)
lib = _lib.load()
N, S = z.shape


def run(fn, blob):
    raw = torch.empty(N, S, 4 + ins_num + 1, device=dev)
    _lib.check(fn(_lib.ptr(blob), ins_num, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), N, S, _lib.ptr(raw), _lib.stream()), "mlp")
    return raw.double().cpu()


# float64 reference on the CPU (reference op order, networks/dm_nerf.py:80-106)
sd = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
pts = (ro[:, None, :] + rd[:, None, :] * z[:, :, None]).double().cpu().reshape(-1, 3)
vd = (rd / torch.norm(rd, dim=-1, keepdim=True)).double().cpu()[:, None, :].expand(N, S, 3).reshape(-1, 3)
emb = lambda x, L: torch.cat([x] + [f(x * 2.0 ** k) for k in range(L) for f in (torch.sin, torch.cos)], -1)
xp, xv = emb(pts, 10), emb(vd, 4)
h = xp
for i in range(8):
    h = torch.relu(h @ sd[f"mlps.{i}.weight"].t() + sd[f"mlps.{i}.bias"])
    if i == 4:
        h = torch.cat([h, xp], -1)
den = h @ sd["density_linear.weight"].t() + sd["density_linear.bias"]
f = h @ sd["rgb_feature_linear.weight"].t() + sd["rgb_feature_linear.bias"]
hr = torch.relu(torch.cat([f, xv], -1) @ sd["rgb_feature_linears.0.weight"].t() + sd["rgb_feature_linears.0.bias"])
rgb = hr @ sd["rgb_linear.weight"].t() + sd["rgb_linear.bias"]
q = h @ sd["ins_feature_linear.weight"].t() + sd["ins_feature_linear.bias"]
hi = torch.relu(q @ sd["ins_feature_linears.0.weight"].t() + sd["ins_feature_linears.0.bias"])
ins = hi @ sd["ins_linear.weight"].t() + sd["ins_linear.bias"]
ref = torch.cat([rgb, den, ins], -1).reshape(N, S, -1)
# (the f32 inputs of the reference are pts / viewdirs rounded to f32 first; use the same rounding)
kernels = (("layerwise f32 MFMA (default)", lib.dmnerf_mlp_fwd_rays, m.blob()),
           ("fused heads, f32 MFMA", lib.dmnerf_mlp_fwd_rays_fused, m.blob_fused()),
           ("fused heads, split-bf16 MFMA (bf16x3)", lib.dmnerf_mlp_fwd_rays_split, m.blob_split()),
           ("fused heads, split-f16 MFMA (f16x2)", lib.dmnerf_mlp_fwd_rays_f16, m.blob_f16()))
for name, fn, blob in kernels:
    got = run(fn, blob)
    e = ((got - ref).abs() / (1 + ref.abs()))
    flips = int((got[..., 4:].argmax(-1) != ref[..., 4:].argmax(-1)).sum())
    print(f"{name:40s} max {float(e.max()):.2e}   mean {float(e.mean()):.2e}   per-sample logit argmax flips {flips}   "
          f"worst channel {int(e.reshape(-1, e.shape[-1]).max(0).values.argmax())}", flush=True)
# launch times on the fine-network shape
zf = H.z_val_sample(4096, 4.0, 15.0, 192, device=dev)
rawf = torch.empty(N, 192, 4 + ins_num + 1, device=dev)
for name, fn, blob in kernels:
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for b, e in ev:
        b.record()
        _lib.check(fn(_lib.ptr(blob), ins_num, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(zf), N, 192, _lib.ptr(rawf), _lib.stream()), "mlp")
        e.record()
    torch.cuda.synchronize()
    ts = sorted(b.elapsed_time(e) for b, e in ev[2:])
    print(f"{name:40s} fine launch (4096 x 192): median {ts[len(ts) // 2]:.3f} ms   min {ts[0]:.3f} ms", flush=True)
