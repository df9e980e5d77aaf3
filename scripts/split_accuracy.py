"""GPU box: accuracy of the three inference kernels against a float64 evaluation of the same network.

layerwise f32 MFMA (default) | fused heads f32 MFMA | fused heads + split-bf16 MFMA -- max and mean of
|raw - raw64| / (1 + |raw64|) over 4096 x 64 samples of the bench camera, ins_num = 13."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dm_nerf_amd import _lib
from dm_nerf_amd.networks import dm_nerf as M, helpers as H, render as R
from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical

dev = torch.device("cuda:0")
torch.manual_seed(0)
ins_num = 13
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
for name, fn, blob in (("layerwise f32 MFMA (default)", lib.dmnerf_mlp_fwd_rays, m.blob()),
                       ("fused heads, f32 MFMA", lib.dmnerf_mlp_fwd_rays_fused, m.blob_fused()),
                       ("fused heads, split-bf16 MFMA", lib.dmnerf_mlp_fwd_rays_split, m.blob_split())):
    e = ((run(fn, blob) - ref).abs() / (1 + ref.abs()))
    print(f"{name:32s} max {float(e.max()):.2e}   mean {float(e.mean()):.2e}")
