"""Diagnostic (GPU box): where a chain_kernel tile's time goes.  Needs a trace build (scripts/diag_chain.sh TRACE;
DMNERF_DIAG_LIB=diag_build/lib_chain_TRACE.so).  Shader-clock stamps of wave 0 on one steady-state tile per workgroup: tile start, per
layer the end of its chunk loop and the end of its epilogue, the end of the output stores, the start of the next tile -- averaged over the
workgroups, against the layer's MFMA work (16 NBB MFMAs of 64 cycles per chunk and wave)."""
import ctypes, os, sys, types, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dm_nerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["DMNERF_DIAG_LIB"])
from dm_nerf_amd import config as Cfg
from dm_nerf_amd.networks import helpers as H, render as R
lib = _lib.load()
lib.dmnerf_mlp_chain_set_trace.restype = ctypes.c_int
lib.dmnerf_mlp_chain_set_trace.argtypes = [ctypes.c_void_p]
D, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "6x128").split("x"))
args = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=D, netwidth=W, ins_num=13, device=torch.device("cuda:0"))
pe, ve, mc, mf, _ = Cfg.create_nerf(args)
N = 4096
ro, rd = torch.randn(N, 3, device="cuda"), torch.randn(N, 3, device="cuda")
z = H.z_val_sample(N, 4., 15., 64)
ea = types.SimpleNamespace(perturb=False, N_importance=128, is_train=False, N_ins=None)
cus = torch.cuda.get_device_properties(0).multi_processor_count
ticks = torch.zeros(cus * 64, dtype=torch.int64, device="cuda")
with torch.no_grad():
    R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ea); torch.cuda.synchronize()
    lib.dmnerf_mlp_chain_set_trace(ctypes.c_void_p(ticks.data_ptr()))
    R.dm_nerf(torch.stack([ro, rd]), pe, ve, mc, mf, z, ea); torch.cuda.synchronize()      # (the fine pass writes last: 24 tiles per workgroup)
    lib.dmnerf_mlp_chain_set_trace(None)
t = ticks.cpu().numpy().reshape(cus, 64).astype(np.float64)
nbb = W // 32
skips = [4]
print(f"D={D} W={W}: shader-clock ticks of wave 0, tile 3 of each workgroup's fine pass, mean over {cus} workgroups ")
scale = 1.0                                                                               # (s_memtime ticks are shader cycles)
tot = 0.0
for l in range(D):
    kin = 2 if l == 0 else (nbb + 2 if (l - 1) in skips else nbb)
    start = t[:, 0] if l == 0 else t[:, 2 * l]
    loop = (t[:, 1 + 2 * l] - start).mean() * scale
    epi = (t[:, 2 + 2 * l] - t[:, 1 + 2 * l]).mean() * scale
    work = kin * 16 * nbb * 64
    print(f"  layer {l}: {kin} chunks, chunk loop {loop:9.0f} cycles against {work} of MFMA work ({work / loop:.2f}); epilogue {epi:7.0f}")
out = (t[:, 40] - t[:, 2 * D]).mean() * scale
gap = (t[:, 41] - t[:, 40]).mean() * scale
whole = (t[:, 41] - t[:, 0]).mean() * scale
print(f"  output stores {out:7.0f}; wait for the next tile's encoding {gap:7.0f}; whole tile {whole:9.0f}")
r = [(t[:, 49 + i] - t[:, 48 + i]).mean() for i in range(5)]
print(f"  layer 2 chunk 1: rounds 0..3 {r[0]:6.0f} {r[1]:6.0f} {r[2]:6.0f} {r[3]:6.0f} ({4 * nbb * 64} of MFMA work each, + ~80 per stamp; round 3 holds the ring hand-over and the refill); fetch bookkeeping {r[4]:5.0f}")
