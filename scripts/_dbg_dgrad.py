import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from dm_nerf_amd import _lib, autograd as G
from dm_nerf_amd.networks import dm_nerf as M
from oracle import ref_cpu as O
lib = _lib.load()
for ins_num, N, S, seed in ((13, 9, 64, 31), (13, 37, 64, 71), (13, 64, 192, 3)):
    sd = O.make_weights(seed, ins_num, gain=1.7)
    m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num); m.load_state_dict(sd); m = m.cuda()
    g = torch.Generator().manual_seed(seed)
    ro, rd = torch.randn(N, 3, generator=g).cuda(), torch.randn(N, 3, generator=g).cuda()
    z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0].cuda()
    M_, C = N * S, ins_num + 1
    graw = torch.randn(M_, 4 + C, generator=g).cuda()
    raw = torch.empty(N, S, 4 + C, device="cuda")
    save = torch.empty(lib.dmnerf_train_save_floats(M_), device="cuda")
    fwdfn, fblob = (lib.dmnerf_mlp_fwd_rays_train_f16, m.blob_f16()) if os.environ.get("F16FWD") else (lib.dmnerf_mlp_fwd_rays_train, m.blob())
    _lib.check(fwdfn(_lib.ptr(fblob), ins_num, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), N, S, _lib.ptr(raw), _lib.ptr(save), _lib.stream()), "fwd")
    Mp = G._row_len(M_)
    outs = []
    for k in range(3):
        dsave = torch.full_like(save, float("nan"))
        gt = torch.full((Mp // 32, 4 + C, 32), float("nan"), device="cuda")
        if k == 0:
            _lib.check(lib.dmnerf_mlp_bwd_data(_lib.ptr(m.blob()), _lib.ptr(m.blob_t()), ins_num, _lib.ptr(save), _lib.ptr(graw), M_, _lib.ptr(dsave), _lib.ptr(gt), _lib.stream()), "bwd")
        else:
            _lib.check(lib.dmnerf_mlp_bwd_data_f16(_lib.ptr(m.blob_t_f16()), ins_num, _lib.ptr(save), _lib.ptr(graw), M_, _lib.ptr(dsave), _lib.ptr(gt), None, _lib.stream()), "bwd f16")
        torch.cuda.synchronize()
        outs.append(dsave.cpu())
    d0, d1, d2 = outs
    print(f"M={M_}: nan pattern equal {bool(torch.equal(torch.isnan(d0), torch.isnan(d1)))}; f16 run-to-run identical {bool(torch.equal(torch.nan_to_num(d1), torch.nan_to_num(d2)))}")
    off = 0
    names = [("pe", 63), ("de", 27)] + [(f"dy{l}", 256) for l in range(8)] + [("dg1", 128), ("dg2", 128)]
    for name, rows in names:
        a, b = torch.nan_to_num(d0[off:off + rows * Mp]), torch.nan_to_num(d1[off:off + rows * Mp])
        if rows >= 128:
            A_ = a.reshape(-1, rows, 32); B_ = b.reshape(-1, rows, 32)
            e = (A_ - B_).abs()
            blk = e.amax(dim=(1, 2))
            bad = (blk > 1e-4 * float(a.abs().max())).nonzero().flatten().tolist()
            rowerr = e.amax(dim=(0, 2))
            badrows = (rowerr > 1e-4 * float(a.abs().max())).nonzero().flatten().tolist()
            print(f"   {name}: maxdiff {float(e.max()):.2e} scale {float(a.abs().max()):.2e}  bad blocks {bad[:12]}{'...' if len(bad) > 12 else ''} ({len(bad)} of {A_.shape[0]})  bad mem-rows {badrows[:16]} ({len(badrows)})")
        off += rows * Mp
