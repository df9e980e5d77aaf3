import sys, time, types, warnings, torch
warnings.filterwarnings('ignore')
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dm_nerf_amd import _lib, config as Cfg
if os.environ.get('DMNERF_DIAG_LIB'):          # a diagnostic build of the library (timing experiments; results may be wrong on purpose)
    _lib.LIB_PATH = os.path.abspath(os.environ['DMNERF_DIAG_LIB'])
from dm_nerf_amd.networks import helpers as H, render as R
RENDER_ONLY = '--render-only' in sys.argv
BURST = '--burst' in sys.argv
SHAPES = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:] if not a.startswith('--')] or [(8, 256), (8, 192), (6, 128), (10, 320)]
for D, W in SHAPES:
    args = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=D, netwidth=W, ins_num=13, device=torch.device("cuda:0"))
    pe, ve, mc, mf, _ = Cfg.create_nerf(args)
    N = 4096
    ro, rd = torch.randn(N, 3, device="cuda"), torch.randn(N, 3, device="cuda")
    z = H.z_val_sample(N, 4., 15., 64)
    ea = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
    # SUSTAINED rate: the part's clock takes tens of milliseconds of continuous work to settle (a cold burst of 3 renders of a W = 128
    # network measures 0.54 of the roof, the 30th render 0.61; scripts/steady_state.py) -- warm up for 0.2 s, then time 30 renders.
    # --burst: the cold 3-render figure of the earlier rounds' tables.
    with torch.no_grad():
        R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ea); torch.cuda.synchronize()
        n_r = 3
        if not BURST:
            n_r = 30
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.2:
                R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ea); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_r): R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ea)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n_r
    mac = sum(p.numel() for n, p in mc.named_parameters() if n.endswith("weight"))
    if RENDER_ONLY:
        print(f'D={D} W={W}: render {dt*1e3:.2f} ms = {2*mac*256*N/dt/1e12/157.3:.3f} of the roof', flush=True)
        continue
    ta = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None)
    mc.train(); mf.train()
    out = R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ta); (out['rgb_fine'].sum() + out['ins_fine'].sum()).backward(); torch.cuda.synchronize()
    n_t = 2 if BURST else 10
    for _ in range(0 if BURST else 4):
        out = R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ta); (out['rgb_fine'].sum() + out['rgb_coarse'].sum() + out['ins_fine'].sum()).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_t):
        out = R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ta); (out['rgb_fine'].sum() + out['rgb_coarse'].sum() + out['ins_fine'].sum()).backward()
    torch.cuda.synchronize(); dtt = (time.perf_counter() - t0) / n_t
    tf = 2 * mac * 256 * N / dt / 1e12
    print(f"D={D} W={W} fused={mc._fused_ok()}: render {dt*1e3:.1f} ms/4096 rays = {N/dt/1e3:.0f} k rays/s ({tf:.1f} TFLOP/s = {tf/157.3:.2f} of the f32 MFMA roof); "
          f"fwd+bwd {dtt*1e3:.1f} ms = {N/dtt/1e3:.0f} k rays/s ({3*tf*dt/dtt:.1f} TFLOP/s on 3x the forward MACs)", flush=True)
