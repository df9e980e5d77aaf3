"""GPU box: the training forward of the split-f16 mode against the fused-heads f32 training forward on the same inputs -- every
saved tensor of the SaveLayout workspace (pe, de, h_0..h_7, g1, g2) and the 1-bit ReLU masks, per layer: how many mask bits
differ (a pre-activation within rounding distance of zero gets a different ReLU bit from two f32-class forwards; ReLU's
derivative is discontinuous there, so such a bit moves single gradient entries by O(dy) -- see
tests/test_gpu_train.py::test_split_backward_kernels_vs_oracle_autograd)."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from dm_nerf_amd import _lib
from dm_nerf_amd.networks import dm_nerf as M
from oracle import ref_cpu as O
lib = _lib.load()
for ins_num, N, S, seed in ((13, 6, 64, 71), (13, 6, 64, 74), (13, 37, 64, 71), (13, 64, 192, 3), (59, 64, 192, 4)):
    sd = O.make_weights(seed, ins_num, gain=1.7)
    m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num); m.load_state_dict(sd); m = m.cuda()
    g = torch.Generator().manual_seed(seed)
    ro, rd = torch.randn(N, 3, generator=g).cuda(), torch.randn(N, 3, generator=g).cuda()
    z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0].cuda()
    M_ = N * S
    Mp = (M_ + 31) // 32 * 32
    n = lib.dmnerf_train_save_floats(M_)
    res = []
    for fn, blob in ((lib.dmnerf_mlp_fwd_rays_train_fused, m.blob_fused()), (lib.dmnerf_mlp_fwd_rays_train_f16, m.blob_f16())):
        raw = torch.empty(N, S, 4 + ins_num + 1, device="cuda")
        save = torch.full((n,), float("nan"), device="cuda")
        _lib.check(fn(_lib.ptr(blob), ins_num, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), N, S, _lib.ptr(raw), _lib.ptr(save), _lib.stream()), "fwd")
        torch.cuda.synchronize()
        res.append((raw.cpu(), save.cpu()))
    (r0, s0), (r1, s1) = res
    print(f"M={M_}: raw maxdiff {float((r0-r1).abs().max()):.2e}; nan pattern equal {bool(torch.equal(torch.isnan(s0), torch.isnan(s1)))}  nans {int(torch.isnan(s0).sum())} vs {int(torch.isnan(s1).sum())}")
    off = 0
    names = [("pe", 63), ("de", 27)] + [(f"h{l}", 256) for l in range(8)] + [("g1", 128), ("g2", 128), ("bits", 72)]
    for name, rows in names:
        a, b = s0[off:off + rows * Mp], s1[off:off + rows * Mp]
        if name == "bits":
            ai, bi = a.view(torch.int32), b.view(torch.int32)
            x = (ai ^ bi)
            nb = sum(int(((x >> k) & 1).sum()) for k in range(32))
            print(f"   bits: words differing {int((ai != bi).sum())} of {ai.numel()}, bits differing {nb}")
            # per layer
            W = ai.reshape(-1, 2304); V = bi.reshape(-1, 2304)
            for l in range(8):
                d = (W[:, l*256:(l+1)*256] != V[:, l*256:(l+1)*256]).sum()
                print(f"      layer {l}: {int(d)} words differ", end="")
            print(f"   g1 {int((W[:,2048:2176]!=V[:,2048:2176]).sum())} g2 {int((W[:,2176:2304]!=V[:,2176:2304]).sum())}")
        else:
            an, bn = torch.nan_to_num(a), torch.nan_to_num(b)
            print(f"   {name}: maxdiff {float((an-bn).abs().max()):.2e} scale {float(an.abs().max()):.2e} nan-mismatch {int((torch.isnan(a)!=torch.isnan(b)).sum())}")
        off += rows * Mp
