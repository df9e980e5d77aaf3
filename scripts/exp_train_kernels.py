"""Experiment harness (GPU box): HIP-event times of the three training kernels at the fine-network shape (4096 rays x 192
samples) for the library named by DMNERF_DIAG_LIB (default: the shipped one) -- used to compare build variants
(`make -C dm_nerf_amd/csrc variant NAME=... FLAGS=...`).  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dm_nerf_amd import _lib

if os.environ.get("DMNERF_DIAG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DMNERF_DIAG_LIB"])
from dm_nerf_amd import autograd as G
from dm_nerf_amd.networks import dm_nerf as M, render as R

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = M.DM_NeRF(8, 256, 63, 27, [4], 13).to(dev).train()
N, S = 4096, int(os.environ.get("EXP_S", "192"))
ro, rd = torch.randn(N, 3, device=dev), torch.randn(N, 3, device=dev)
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 1, -1)[0]
cot = None
reps = int(os.environ.get("EXP_REPS", "12"))
G.KERNEL_EVENTS = []
for it in range(reps + 2):
    if it == 2:
        G.KERNEL_EVENTS.clear()
    for p in m.parameters():
        p.grad = None
    raw = G.run_network_train(m, ro, rd, z)
    if cot is None:
        cot = torch.randn_like(raw)
    (raw * cot).sum().backward()
torch.cuda.synchronize()
ev = G.KERNEL_EVENTS
out = {"lib": os.path.basename(_lib.LIB_PATH), "S": S}
for tag in ("mlp_fwd_train", "mlp_bwd_data", "mlp_bwd_weights"):
    ms = [b.elapsed_time(e) for t, M_, b, e in ev if t == tag]
    out[tag] = {"mean_ms": float(np.mean(ms)), "min_ms": float(np.min(ms))}
with torch.no_grad():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R.run_network(m, ro, rd, z)
    e0.record()
    for _ in range(reps):
        R.run_network(m, ro, rd, z)
    e1.record(); torch.cuda.synchronize()
    out["mlp_fwd_inference"] = {"mean_ms": e0.elapsed_time(e1) / reps}
print(json.dumps(out))
