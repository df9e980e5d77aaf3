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
                ("cls", "<i4"), ("chunk0", "<i4"), ("nchunk", "<i4"), ("follow", "<i4"), ("next", "<i4"), ("pad", "<i4")])
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
        jh = jobs.cpu().numpy().view(JOB)            # [n_jobs leaders (one per workgroup)] + [their follower items]
        assert bool((jh["follow"][:n_jobs] >= 0).all()) and bool((jh["follow"][n_jobs:] == -1).all())
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
        print(f"== M = {Mtot} : {n_jobs} workgroups ({len(jh)} items), kernel span {end.max():.0f} us, latest start {start.max():.0f} us, "
              f"workgroup time mean {dur.mean():.0f} min {dur.min():.0f} max {dur.max():.0f} us")
        # per workgroup: its items; least-squares fit  dur = sum over items (c_cls * nchunk) + o_cls per item
        items = [[i] + list(range(int(jh["next"][i]), int(jh["next"][i]) + int(jh["follow"][i]))) for i in range(n_jobs)]
        used = sorted(set(int(c) for c in jh["cls"]))
        A = np.zeros((n_jobs, 2 * len(used)))
        for w, its in enumerate(items):
            for k in its:
                c = used.index(int(jh["cls"][k]))
                A[w, c] += float(jh["nchunk"][k])
                A[w, len(used) + c] += 1.0
        sol, *_ = np.linalg.lstsq(A, dur * 1e3, rcond=None)
        res = A @ sol - dur * 1e3
        print("  fitted ns per 32-sample chunk / ns per item (ring fill + tile store), by shape class:")
        for c, cls in enumerate(used):
            nba, nbb = CLS[cls]
            n_it = int(A[:, len(used) + c].sum())
            print(f"    ({nba},{nbb}): {sol[c]:8.0f} ns per chunk, {sol[len(used) + c]:9.0f} ns per item   [{n_it} items;  MFMA-ideal "
                  f"{({None: 256, 'bf16x3': 96, 'f16x2': 48}[split]) * nba * nbb / 2.4:6.0f} ns;  HBM at 6.3 TB/s / 256 CUs: {(nba + nbb) * 4096 / 24.6:6.0f} ns]")
        print(f"  fit residual rms {np.sqrt((res ** 2).mean()) / 1e3:.1f} us, max {np.abs(res).max() / 1e3:.1f} us")
        worst = np.argsort(-dur)[:8]
        for w in worst:
            print(f"    slowest wg {w:3d}: {dur[w]:7.0f} us  items " + ", ".join(f"{CLS[int(jh['cls'][k])]}x{int(jh['nchunk'][k])}" for k in items[w]))


if __name__ == "__main__":
    main()
