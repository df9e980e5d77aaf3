import sys, os, itertools
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from dm_nerf_amd import _lib, autograd as G
from dm_nerf_amd.networks import dm_nerf as M
from oracle import ref_cpu as O
lib = _lib.load()
ins_num, N, S, seed = 13, 37, 64, 71
sd = O.make_weights(seed, ins_num, gain=1.7)
m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num); m.load_state_dict(sd); m = m.cuda()
g = torch.Generator().manual_seed(seed)
ro, rd = torch.randn(N, 3, generator=g).cuda(), torch.randn(N, 3, generator=g).cuda()
z = torch.sort(torch.rand(N, S, generator=g) * 5 + 1, -1)[0].cuda()
M_, C = N * S, ins_num + 1
graw = torch.randn(M_, 4 + C, generator=g).cuda()
Mp = G._row_len(M_)
flat = m.flat()
ref = None
for fwd, dg, wg, fill in itertools.product(("f32", "f16"), ("f32", "f16"), ("f32", "split"), ("nan", "empty")):
    raw = torch.empty(N, S, 4 + C, device="cuda")
    mk = (lambda n: torch.full((n,), float("nan"), device="cuda")) if fill == "nan" else (lambda n: torch.empty(n, device="cuda"))
    save = mk(lib.dmnerf_train_save_floats(M_))
    if fwd == "f32":
        _lib.check(lib.dmnerf_mlp_fwd_rays_train(_lib.ptr(m.blob()), ins_num, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), N, S, _lib.ptr(raw), _lib.ptr(save), _lib.stream()), "fwd")
    else:
        _lib.check(lib.dmnerf_mlp_fwd_rays_train_f16(_lib.ptr(m.blob_f16()), ins_num, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), N, S, _lib.ptr(raw), _lib.ptr(save), _lib.stream()), "fwd")
    dsave = mk(save.numel())
    gt = mk(Mp // 32 * (4 + C) * 32)
    if dg == "f32":
        _lib.check(lib.dmnerf_mlp_bwd_data(_lib.ptr(m.blob()), _lib.ptr(m.blob_t()), ins_num, _lib.ptr(save), _lib.ptr(graw), M_, _lib.ptr(dsave), _lib.ptr(gt), _lib.stream()), "bwd")
    else:
        _lib.check(lib.dmnerf_mlp_bwd_data_f16(_lib.ptr(m.blob_t_f16()), ins_num, _lib.ptr(save), _lib.ptr(graw), M_, _lib.ptr(dsave), _lib.ptr(gt), None, _lib.stream()), "bwd f16")
    jobs, n_jobs, outs, n_outs, pf = G.wgrad_plan(ins_num, M_, raw.device, split=wg == "split")
    part = mk(pf)
    out = mk(lib.dmnerf_param_count(ins_num))
    fn = lib.dmnerf_mlp_bwd_weights_split if wg == "split" else lib.dmnerf_mlp_bwd_weights
    _lib.check(fn(_lib.ptr(save), _lib.ptr(dsave), _lib.ptr(gt), M_, _lib.ptr(jobs), n_jobs, _lib.ptr(outs), n_outs, _lib.ptr(flat), ins_num, _lib.ptr(part), _lib.ptr(out), _lib.stream()), "wgrad")
    torch.cuda.synchronize()
    o = out.cpu()
    if ref is None:
        ref = o
    gs = G.split_flat_grads(m, o) if False else None
    # mlps.0: 256*63 + 256 ; mlps.1: 256*256+256
    n0 = 256 * 63 + 256
    n1 = 256 * 256 + 256
    e0 = float((o[:n0] - ref[:n0]).abs().max()) / float(ref[:n0].abs().max())
    e1 = float((o[n0:n0 + n1] - ref[n0:n0 + n1]).abs().max()) / float(ref[n0:n0 + n1].abs().max())
    er = float((o[n0 + n1:] - ref[n0 + n1:]).abs().max()) / float(ref[n0 + n1:].abs().max())
    print(f"fwd {fwd:3s} dgrad {dg:3s} wgrad {wg:5s} fill {fill:5s}: mlps.0 {e0:.2e}  mlps.1 {e1:.2e}  rest {er:.2e}  nans {int(torch.isnan(o).sum())}")
