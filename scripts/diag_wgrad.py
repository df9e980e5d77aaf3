"""Diagnostic (GPU box): per-workgroup timing of the split-K weight-gradient kernel.

Runs one fine-level (4096 x 192 samples) and one coarse-level (4096 x 64) MLP backward with
dmnerf_wgrad_set_trace on and prints, per job of the plan, the slice count, chunks per slice and
the measured time per workgroup / per 32-sample chunk.  Used to fit chunk_cost() in csrc/wgrad.hip.
DMNERF_DIAG_SPLIT=1: the opt-in split-bf16 backward (csrc/wgrad_split.hip) and its plan; DMNERF_DIAG_SPLIT=f16: the split-f16
backward (csrc/wgrad_f16.hip; the split plan).
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dm_nerf_amd import _lib, autograd as G

if os.environ.get("DMNERF_DIAG_LIB"):          # timing experiments: an alternative build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["DMNERF_DIAG_LIB"])
from dm_nerf_amd.networks import dm_nerf as M

JOB = np.dtype([("a_off", "<i8"), ("b_off", "<i8"), ("a_R", "<i4"), ("b_R", "<i4"), ("a_row0", "<i4"), ("b_row0", "<i4"),
                ("part_off", "<i8"), ("bias_off", "<i8"), ("a_src", "<i4"), ("b_src", "<i4"), ("rowsA", "<i4"), ("rowsB", "<i4"),
                ("cls", "<i4"), ("chunk0", "<i4"), ("nchunk", "<i4"), ("pad", "<i4")])
CLS = [(8, 8), (4, 8), (8, 2), (4, 1), (1, 8), (1, 4), (2, 4), (3, 4), (4, 4)]


def main():
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    lib = _lib.load()
    ins_num = 13
    m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(dev).train()
    for S in (192,) if os.environ.get('DMNERF_DIAG_LIB') else (192, 64):
        N = 4096
        ro, rd = torch.randn(N, 3, device=dev), torch.randn(N, 3, device=dev)
        z = torch.sort(torch.rand(N, S, device=dev) * 4 + 1, -1)[0]
        Mtot = N * S
        sv = os.environ.get("DMNERF_DIAG_SPLIT", "0")
        split = {"0": None, "1": "bf16x3", "f16": "f16x2"}[sv]
        jobs, n_jobs, outs, n_outs, _ = G.wgrad_plan(ins_num, Mtot, dev, split=split or False)
        jh = jobs.cpu().numpy().view(JOB)
        assert JOB.itemsize * n_jobs == jobs.numel(), (JOB.itemsize, n_jobs, jobs.numel())
        ticks = torch.zeros(2 * n_jobs, dtype=torch.int64, device=dev)
        for it in range(3):
            for p in m.parameters():
                p.grad = None
            raw = G.run_network_train(m, ro, rd, z, split=split)
            cot = torch.randn_like(raw)
            if it == 2:
                lib.dmnerf_wgrad_set_trace(ctypes.c_void_p(ticks.data_ptr()))
            (raw * cot).sum().backward()
            torch.cuda.synchronize()
        lib.dmnerf_wgrad_set_trace(None)
        t = ticks.cpu().numpy().reshape(n_jobs, 2)
        t0 = t[:, 0].min()
        dur = (t[:, 1] - t[:, 0]) * 0.01          # us (100 MHz)
        start = (t[:, 0] - t0) * 0.01
        end = (t[:, 1] - t0) * 0.01
        print(f"== M = {Mtot} : {n_jobs} workgroups, kernel span {end.max():.0f} us, latest start {start.max():.0f} us")
        # group consecutive workgroups of the same job (same part stride pattern: same a_off/b_off/rows)
        key = [(int(j["a_off"]), int(j["b_off"]), int(j["a_row0"]), int(j["b_row0"]), int(j["a_src"])) for j in jh]
        i = 0
        while i < n_jobs:
            k = i
            while k < n_jobs and key[k] == key[i]:
                k += 1
            j = jh[i]
            nba, nbb = CLS[int(j["cls"])]
            d = dur[i:k]
            nch = jh["nchunk"][i:k]
            print(f"  job rows {int(j['rowsA']):3d}x{int(j['rowsB']):3d} cls ({nba},{nbb}) slices {k - i:3d} chunks/slice {int(nch.max()):5d}"
                  f"  wg time mean {d.mean():8.0f} max {d.max():8.0f} us   per chunk {1e3 * (d / nch).mean():7.0f} ns"
                  f"  (MFMA-ideal {({None: 256, 'bf16x3': 96, 'f16x2': 48}[split]) * nba * nbb / 2.4:7.0f} ns;  HBM at 6.3 TB/s / 256 CUs: {(nba + nbb) * 4096 / 24.6:7.0f} ns)")
            i = k


if __name__ == "__main__":
    main()
