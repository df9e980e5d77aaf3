"""GPU box: cold bursts against sustained work -- the same gemm_nt layer launched 1 .. 400 times back to back, and 3 / 30 / 150 renders of two
network shapes: the part's clock takes tens of milliseconds of continuous work to settle (profiles/r06/steady_state_r06.txt)."""
import sys, os, time, types, warnings, torch
warnings.filterwarnings('ignore')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dm_nerf_amd import _lib, config as Cfg, generic as G
from dm_nerf_amd.networks import helpers as H, render as R
lib = _lib.load()
M = 786432
for K, N in ((128, 128), (192, 192), (320, 320)):
    x = G._Act.empty(M, K, "cuda"); x.buf.normal_()
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    pk = G._Packed(W, torch.randn(N, device="cuda"), [(0, K)])
    y = G._Act.empty(M, N, "cuda")
    for reps in (1, 4, 20, 100, 400):
        torch.cuda.synchronize(); time.sleep(0.2)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        for i in range(reps):
            if i == reps - max(1, reps // 4): ev[1].record()
            G._linear_nt(x, pk, y.buf, y.ld, N, y.ld, M, relu=True)
        ev[2].record(); torch.cuda.synchronize()
        tail = ev[1].elapsed_time(ev[2]) / max(1, reps // 4)
        print(f"K=N={K}: {reps:3d} launches back to back: mean {ev[0].elapsed_time(ev[2]) / reps * 1e3:7.1f} us, last quarter {tail * 1e3:7.1f} us = {2.0 * M * K * N / tail / 1e9 / 157.3:.3f} of the roof", flush=True)
for D, Wd in ((6, 128), (8, 192)):
    args = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=D, netwidth=Wd, ins_num=13, device=torch.device("cuda:0"))
    pe, ve, mc, mf, _ = Cfg.create_nerf(args)
    Nr = 4096
    ro, rd = torch.randn(Nr, 3, device="cuda"), torch.randn(Nr, 3, device="cuda")
    z = H.z_val_sample(Nr, 4., 15., 64)
    ea = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    mac = sum(p.numel() for n, p in mc.named_parameters() if n.endswith("weight"))
    with torch.no_grad():
        R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ea); torch.cuda.synchronize()
        for reps in (3, 30, 150):
            time.sleep(0.2)
            t0 = time.perf_counter()
            for _ in range(reps): R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ea)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
            print(f"D={D} W={Wd}: {reps} renders back to back: {dt*1e3:.2f} ms = {2*mac*256*Nr/dt/1e12/157.3:.3f} of the roof", flush=True)
