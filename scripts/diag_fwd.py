"""Diagnostic (GPU box): per-workgroup cycle stamps of the inference MLP kernel (build with -DDMN_FWD_TRACE)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dm_nerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("DMNERF_DIAG_LIB", "build_exp/lib_ftrace.so"))
from dm_nerf_amd.networks import dm_nerf as M
from dm_nerf_amd.networks import render as R

dev = torch.device("cuda:0")
lib = _lib.load()
lib.dmnerf_debug_fwd_trace.restype = ctypes.c_int
lib.dmnerf_debug_fwd_trace.argtypes = [ctypes.c_void_p]
m = M.DM_NeRF(8, 256, 63, 27, [4], 13).to(dev)
N, S = 4096, 192
ro, rd = torch.randn(N, 3, device=dev), torch.randn(N, 3, device=dev)
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 1, -1)[0]
nwg = (N * S + 127) // 128
tr = torch.zeros(8 * nwg, dtype=torch.int64, device=dev)
with torch.no_grad():
    for it in range(4):
        if it == 3:
            lib.dmnerf_debug_fwd_trace(ctypes.c_void_p(tr.data_ptr()))
        raw = R.run_network(m, ro, rd, z) if hasattr(R, "run_network") else None
        torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(nwg, 8)
d = lambda a, b: (t[:, b] - t[:, a]).astype(np.float64)
print(f"{nwg} workgroups; cycles (mean / p95):")
for name, a, b in (("prologue (inputs, table, encode)", 0, 1), ("mlps.0", 1, 2), ("trunk 7 stages + density", 2, 3), ("rgb + ins heads", 3, 4), ("output stores", 4, 5), ("whole workgroup", 0, 5)):
    x = d(a, b)
    print(f"  {name:34s} {x.mean():10.0f} {np.percentile(x, 95):10.0f}")
wall = t[:, 7]
span = (wall.max() - wall.min()) * 10e-9
tot = d(0, 5)
print(f"first->last WG start {span * 1e3:.3f} ms; sum of WG cycles / 256 CUs = {tot.sum() / 256 / 2.4e6:.3f} ms at 2.4 GHz; ideal MFMA {10836 * 64 * 24 / 2.4e6:.3f} ms")
