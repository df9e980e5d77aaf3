"""GPU box: where the opt-in split-f16 forward kernel (csrc/mlp_f16_impl.h) spends its time.

    DMNERF_DIAG_LIB=build_exp/lib_f16_<name>.so python scripts/diag_f16.py      (make -C dm_nerf_amd/csrc f16var NAME=.. FLAGS=..)

Times the fine-network launch (4096 x 192 samples, ins_num 13) of the library named by DMNERF_DIAG_LIB (default: the shipped
one).  A -DDMN_F16_TRACE build also reports the per-workgroup cycle stamps: prologue / mlps.0 / trunk / heads / outputs in
shader cycles (s_memtime), against the MFMA issue time of each section (32 cycles per v_mfma_f32_32x32x16_f16), and the shader
clock the chip granted (cycles / wall time)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dm_nerf_amd import _lib

if os.environ.get("DMNERF_DIAG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DMNERF_DIAG_LIB"])
from dm_nerf_amd.networks import dm_nerf as M, helpers as H
from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical

dev = torch.device("cuda:0")
torch.manual_seed(0)
ins_num = int(os.environ.get("EXP_INS", "13"))
m = M.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(dev)
K = dmsr_intrinsics(480, 640)
ro, rd = H.get_rays_k(480, 640, K, pose_spherical(30.0, -65.0, 7.0).to(dev))
N, S = 4096, int(os.environ.get("EXP_S", "192"))
ro, rd = ro.reshape(-1, 3)[:N].contiguous(), rd.reshape(-1, 3)[:N].contiguous()
z = H.z_val_sample(N, 4.0, 15.0, S, device=dev)
raw = torch.empty(N, S, 4 + ins_num + 1, device=dev)
lib = _lib.load()
blob = m.blob_f16()
n_wg = (N * S + 127) // 128
trace = None
TRAIN = bool(int(os.environ.get("EXP_TRAIN", "0")))          # the training forward (SAVE: rows + masks)
setter = "dmnerf_f16_train_set_trace" if TRAIN else "dmnerf_f16_set_trace"
if hasattr(lib, setter):
    trace = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
    getattr(lib, setter).argtypes = [ctypes.c_void_p]
    getattr(lib, setter)(ctypes.c_void_p(trace.data_ptr()))
save = torch.empty(lib.dmnerf_train_save_floats(N * S), device=dev) if TRAIN else None


def launch():
    if TRAIN:
        _lib.check(lib.dmnerf_mlp_fwd_rays_train_f16(_lib.ptr(blob), ins_num, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), N, S, _lib.ptr(raw), _lib.ptr(save), _lib.stream()), "f16 train")
        return
    _lib.check(lib.dmnerf_mlp_fwd_rays_f16(_lib.ptr(blob), ins_num, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), N, S, _lib.ptr(raw), _lib.stream()), "f16")


ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
for b, e in ev:
    b.record(); launch(); e.record()
torch.cuda.synchronize()
ts = sorted(b.elapsed_time(e) for b, e in ev[2:])
name = os.path.basename(os.environ.get("DMNERF_DIAG_LIB", "libdmnerf_hip.so"))
mfma = 24 * (140 + {1: 1, 2: 2, 3: 4, 4: 4}[(ins_num + 32) // 32])
print(f"{name}{' [training forward]' if TRAIN else ''}: {N} x {S} samples, ins_num {ins_num}: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms   "
      f"(MFMA issue alone at 2.4 GHz: {n_wg / 256 * mfma * 32 / 2.4e6:.3f} ms)")
if trace is not None:
    t = trace.cpu().numpy().reshape(n_wg, 8)
    d = np.diff(t[:, :6], axis=1)                      # prologue, mlps.0, trunk, heads, outputs
    wall = (t[:, 7] - t[:, 6]).astype(np.float64) * 10.0          # wall_clock64: 100 MHz -> ns
    cyc = (t[:, 5] - t[:, 0]).astype(np.float64)
    sect = ("prologue", "mlps.0", "trunk", "heads", "outputs")
    ideal = (0, 96 * 32, 116 * 24 * 32, (mfma - 96 - 116 * 24) * 32, 0)
    med = np.median(d, axis=0)
    for s_, c, i in zip(sect, med, ideal):
        print(f"  {s_:9s} {c:9.0f} cycles" + (f"   MFMA issue {i:7d}  ({i / c:.3f})" if i else ""))
    print(f"  total     {np.median(cyc):9.0f} cycles per workgroup, MFMA issue {mfma * 32} ({mfma * 32 / np.median(cyc):.3f});  "
          f"shader clock while running {np.median(cyc / wall):.3f} GHz (cycles / wall ns, median over workgroups)")
    span = (t[:, 7].max() - t[:, 6].min()) * 10.0 / 1e6
    print(f"  first start -> last end: {span:.3f} ms;  workgroups per CU {n_wg / 256:.1f}")
