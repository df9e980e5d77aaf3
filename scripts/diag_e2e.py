"""Diagnostic: end-to-end training grads -- HIP vs oracle f32 vs oracle f64 (truth)."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ref_cpu as O
from dm_nerf_amd.networks import dm_nerf as M, render as R

ins_num, N = 13, int(sys.argv[1]) if len(sys.argv) > 1 else 48
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sd_c = O.make_weights(31, ins_num, gain=1.7, sigma_bias=0.3)
sd_f = O.make_weights(32, ins_num, gain=1.7, sigma_bias=0.3)
K = O.dmsr_intrinsics(480, 640)
ro, rd = O.get_rays_k(480, 640, K, O.pose_spherical(25.0, -65.0, 7.0))
sel = torch.from_numpy(np.random.RandomState(SEED).choice(480 * 640, N, replace=False))
rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
z = O.z_val_sample(N, 4.0, 15.0, 64).contiguous()
g = torch.Generator().manual_seed(30 + SEED)
t_rand, u = torch.rand(N, 64, generator=g), torch.rand(N, 128, generator=g)
C = ins_num + 1
cts = [torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, ins_num, generator=g),
       torch.randn(N, ins_num, generator=g), 0.01 * torch.randn(N, 192, C, generator=g), 0.01 * torch.randn(N, 64, C, generator=g)]
def loss_from(out, cts):
    return ((out['rgb_fine'] * cts[0]).sum() + (out['rgb_coarse'] * cts[1]).sum() + (out['ins_fine'] * cts[2]).sum()
            + (out['ins_coarse'] * cts[3]).sum() + (out['raw_fine'][..., 4:] * cts[4]).sum() + (out['raw_coarse'][..., 4:] * cts[5]).sum())
def mk(sd):
    m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num); m.load_state_dict(sd); return m.cuda().train()
mc, mf = mk(sd_c), mk(sd_f)
args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None)
out = R.dm_nerf(rays.cuda(), None, None, mc, mf, z.cuda(), args, t_rand=t_rand.cuda(), u=u.cuda())
loss_from(out, [c.cuda() for c in cts]).backward()
zf = out['z_vals_fine'].detach().cpu()
def oracle(dt):
    sdc = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd_c.items()}
    sdf = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd_f.items()}
    w = O.dm_nerf(rays.to(dt), sdc, sdf, z.to(dt), perturb=1.0, t_rand=t_rand.to(dt), u=u.to(dt), z_fine_override=zf.to(dt))
    loss_from(w, [c.to(dt) for c in cts]).backward()
    return sdc, sdf
c32, f32_ = oracle(torch.float32)
c64, f64_ = oracle(torch.float64)
ratios, eh, eo = [], [], []
for name, m, o32, o64 in (("coarse", mc, c32, c64), ("fine", mf, f32_, f64_)):
    for k, p in m.named_parameters():
        t = o64[k].grad
        s = float(t.abs().max())
        e_hip = float((p.grad.cpu().double() - t).abs().max())
        e_o32 = float((o32[k].grad.double() - t).abs().max())
        print(f"{name:6s} {k:32s} scale {s:.2e}  hip-vs-f64 {e_hip / s:.2e}  oracle32-vs-f64 {e_o32 / s:.2e}")
        l_hip = float((p.grad.cpu().double() - t).norm() / (t.norm() + 1e-300))
        l_o32 = float((o32[k].grad.double() - t).norm() / (t.norm() + 1e-300))
        if s > 0:
            ratios.append((e_hip / max(e_o32, 1e-300), l_hip / max(l_o32, 1e-300), name, k)); eh.append((e_hip / s, l_hip)); eo.append((e_o32 / s, l_o32))
r = np.array([[a, b] for a, b, _, _ in ratios])
eh, eo = np.array(eh), np.array(eo)
print(f"N={N} seed={SEED}: tensors {len(r)}; max-err ratio hip/o32: median {np.median(r[:,0]):.2f} max {r[:,0].max():.2f} (at {ratios[int(r[:,0].argmax())][2:]}); "
      f"l2-err ratio: median {np.median(r[:,1]):.2f} max {r[:,1].max():.2f}; sum max-err hip {eh[:,0].sum():.3e} o32 {eo[:,0].sum():.3e}; "
      f"sum l2 hip {eh[:,1].sum():.3e} o32 {eo[:,1].sum():.3e}; worst hip max-err {eh[:,0].max():.2e}, worst o32 {eo[:,0].max():.2e}")
