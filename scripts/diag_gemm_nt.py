"""Diagnostic (GPU box): where a gemm_nt tile's time goes.  Needs a trace build of the library
(hipcc ... -DDMN_NT_TRACE -c gemm_nt.hip, linked like the Makefile does; DMNERF_DIAG_LIB=<that .so>).
For M = 786 432 samples and a few (K, N): shader-clock stamps of workgroup wave 0 per tile -- tile start, end of the K loop, end of the
epilogue -- averaged over the workgroups; against the MFMA work of the K loop (16 NBB MFMAs of 64 cycles per chunk and wave)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dm_nerf_amd import _lib
if os.environ.get("DMNERF_DIAG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DMNERF_DIAG_LIB"])
from dm_nerf_amd import generic as G
lib = _lib.load()
lib.dmnerf_gemm_nt_set_trace.restype = ctypes.c_int
lib.dmnerf_gemm_nt_set_trace.argtypes = [ctypes.c_void_p]
M = 786432
for K, N in ((320, 320), (192, 192), (128, 128), (63, 192)):
    x = G._Act.empty(M, K, "cuda"); x.buf.normal_()
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    pk = G._Packed(W, b, [(0, K)])
    y = G._Act.empty(M, N, "cuda")
    nbb = int(lib.dmnerf_gemm_nt_blocks(N)); occ = 2 if nbb <= 6 else 1          # csrc/gemm_nt.hip::nt_occupancy
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    grid = min((M + 127) // 128, cus * occ)
    ticks = torch.zeros(grid * 32 * 4, dtype=torch.int64, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(4):
        if it == 3:
            lib.dmnerf_gemm_nt_set_trace(ctypes.c_void_p(ticks.data_ptr()))
            ev[0].record()
        G._linear_nt(x, pk, y.buf, y.ld, N, y.ld, M, relu=True)
    ev[1].record(); torch.cuda.synchronize()
    lib.dmnerf_gemm_nt_set_trace(None)
    ms = ev[0].elapsed_time(ev[1])
    t = ticks.cpu().numpy().reshape(grid, 32, 4)
    ntile = min(32, (M // 128) // grid)
    t = t[:, :ntile, :3].astype(np.float64)
    kloop = (t[:, :, 1] - t[:, :, 0]).mean(); epi = (t[:, :, 2] - t[:, :, 1]).mean()
    gap = (t[:, 1:, 0] - t[:, :-1, 2]).mean() if ntile > 1 else 0.0
    whole = (t[:, -1, 2] - t[:, 0, 0]).mean() / ntile
    nchunk = (K + 31) // 32
    ideal = nchunk * 16 * nbb * 64
    tf = 2.0 * M * K * N / (ms * 1e-3) / 1e12
    print(f"K={K} N={N} NBB={nbb} occ={occ}: {ms * 1e3:.0f} us = {tf:.1f} TF ({tf / 157.3:.3f}); per tile (clock ticks of wave 0): K loop {kloop:.0f} (MFMA work {ideal}, "
          f"{ideal / kloop:.3f} if ticks are shader cycles), epilogue {epi:.0f}, tile-to-tile gap {gap:.0f}, tile period {whole:.0f}; tiles per workgroup {ntile}", flush=True)
