import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from dm_nerf_amd import _lib, autograd as G
from dm_nerf_amd.networks import dm_nerf as M
from oracle import ref_cpu as O
for ins_num, N, S, seed in ((13, 8, 64, 15), (13, 37, 64, 71)):
    sd = O.make_weights(seed, ins_num, gain=1.7)
    g = torch.Generator().manual_seed(seed)
    ro, rd = torch.randn(N, 3, generator=g).cuda(), torch.randn(N, 3, generator=g).cuda()
    z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0].cuda()
    cot = torch.randn(N, S, 4 + ins_num + 1, generator=g).cuda()
    grads = {}
    for mode in (None, "bf16x3", "f16x2"):
        m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num); m.load_state_dict(sd); m = m.cuda().train()
        raw = G.run_network_train(m, ro, rd, z, split=mode)
        (raw * cot).sum().backward()
        torch.cuda.synchronize()
        grads[mode] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    print(f"M={N*S}")
    for k in grads[None]:
        a = grads[None][k]
        e1 = float((grads["bf16x3"][k] - a).abs().max()) / (float(a.abs().max()) + 1e-30)
        e2 = float((grads["f16x2"][k] - a).abs().max()) / (float(a.abs().max()) + 1e-30)
        flag = "  <<<<" if e2 > 1e-4 else ""
        print(f"   {k:32s} scale {float(a.abs().max()):.3e}  bf16x3 {e1:.2e}  f16x2 {e2:.2e}{flag}")
