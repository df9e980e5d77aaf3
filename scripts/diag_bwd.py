"""Diagnostic: per-parameter gradient errors of the HIP MLP backward vs oracle autograd (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as O
from dm_nerf_amd import autograd as G, _lib
from dm_nerf_amd.networks import dm_nerf as M

torch.manual_seed(0)
import os
N, S, ins_num, seed = int(os.environ.get("DN", 8)), int(os.environ.get("DS", 64)), 13, 5
sd = O.make_weights(seed, ins_num, gain=1.7)
g = torch.Generator().manual_seed(seed)
rays_o = torch.randn(N, 3, generator=g); rays_d = torch.randn(N, 3, generator=g)
z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0]
cot = torch.randn(N, S, 4 + ins_num + 1, generator=g)
sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
vd = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
x = torch.cat([O.embed(pts.reshape(-1, 3), 10), O.embed(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
raw, acts = O.mlp_forward(sdg, x, return_acts=True)
for a in acts: a.retain_grad()
(raw.reshape(N, S, -1) * cot).sum().backward()
m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num); m.load_state_dict(sd); m = m.cuda().train()
rawg = G.run_network_train(m, rays_o.cuda(), rays_d.cuda(), z.cuda())
fn = rawg.grad_fn
save = fn.save.clone()
Mtot = N * S
xs = G._views(save, Mtot)
print("fwd raw err", float((rawg.detach().cpu() - raw.detach().reshape(N, S, -1)).abs().max()))
for l in range(8):
    want = acts[l].detach()[:, :256].t()
    print(f"saved h{l} err", float((xs['h'][l].cpu()[:, :Mtot] - want).abs().max()))
print("saved pe err", float((xs['pe'].cpu()[:, :Mtot] - x[:, :63].t()).abs().max()), "de err", float((xs['de'].cpu()[:, :Mtot] - x[:, 63:].t()).abs().max()))
(rawg * cot.cuda()).sum().backward()
for k, p in m.named_parameters():
    w = sdg[k].grad
    e = float((p.grad.cpu() - w).abs().max()); s = float(w.abs().max())
    print(f"{k:34s} err {e:.3e} scale {s:.3e} rel {e / (s + 1e-30):.2e}")
