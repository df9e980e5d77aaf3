"""Diagnostic (GPU box): per-workgroup cycle stamps of the MLP kernels (`make -C dm_nerf_amd/csrc diag`).

Inference forward, training forward and dgrad of the fine-level shape (4096 rays x 192 samples): mean cycles per
128-sample workgroup between the stamps, against the MFMA issue time of the section.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dm_nerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("DMNERF_DIAG_LIB", "build_exp/lib_ftrace.so"))
from dm_nerf_amd import autograd as G
from dm_nerf_amd.networks import dm_nerf as M
from dm_nerf_amd.networks import render as R

dev = torch.device("cuda:0")
lib = _lib.load()
for fn in (lib.dmnerf_debug_fwd_trace, lib.dmnerf_debug_bwd_trace):
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
m = M.DM_NeRF(8, 256, 63, 27, [4], 13).to(dev)
N, S = 4096, 192
ro, rd = torch.randn(N, 3, device=dev), torch.randn(N, 3, device=dev)
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 1, -1)[0]
nwg = (N * S + 127) // 128


def report(title, t, rows):
    d = lambda a, b: (t[:, b] - t[:, a]).astype(np.float64)
    print(f"{title}: {nwg} workgroups; cycles mean / p95 (MFMA issue time of the section)")
    for name, a, b, mf in rows:
        x = d(a, b)
        print(f"  {name:34s} {x.mean():10.0f} {np.percentile(x, 95):10.0f}   ({mf * 64:8d})")


tr = torch.zeros(8 * nwg, dtype=torch.int64, device=dev)
with torch.no_grad():
    for it in range(4):
        if it == 3:
            lib.dmnerf_debug_fwd_trace(ctypes.c_void_p(tr.data_ptr()))
        R.run_network(m, ro, rd, z)
        torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(nwg, 8)
FWD = (("prologue (inputs, table)", 0, 1, 0), ("encode + mlps.0", 1, 2, 256), ("trunk 7 stages + skip + density", 2, 3, 7424),
       ("rgb + ins heads", 3, 4, 3200), ("output stores", 4, 5, 0), ("whole workgroup", 0, 5, 10880))
report("inference forward", t, FWD)
wall = t[:, 7]
print(f"  first->last WG start {(wall.max() - wall.min()) * 10e-6:.3f} ms; sum of WG cycles / 256 CUs = {(t[:, 5] - t[:, 0]).sum() / 256 / 2.4e6:.3f} ms at 2.4 GHz")

m.train()
tb = torch.zeros(8 * nwg, dtype=torch.int64, device=dev)
for it in range(3):
    tr.zero_()
    for p in m.parameters():
        p.grad = None
    if it == 2:
        lib.dmnerf_debug_fwd_trace(ctypes.c_void_p(tr.data_ptr()))
        lib.dmnerf_debug_bwd_trace(ctypes.c_void_p(tb.data_ptr()))
    raw = G.run_network_train(m, ro, rd, z)
    (raw * torch.randn_like(raw)).sum().backward()
    torch.cuda.synchronize()
lib.dmnerf_debug_fwd_trace(None); lib.dmnerf_debug_bwd_trace(None)
report("training forward", tr.cpu().numpy().reshape(nwg, 8), FWD)
report("dgrad", tb.cpu().numpy().reshape(nwg, 8),
       (("prologue (grad, masks, table, d raw^T)", 0, 1, 0), ("ins branch (3 quarters, dg2 / dq stores)", 1, 2, 64 + 512),
        ("rgb branch (VALU dg1, 2 quarters)", 2, 3, 512), ("trunk 8 stages", 3, 4, 8192), ("dy_0 store burst", 4, 5, 0),
        ("whole workgroup", 0, 5, 9280)))
