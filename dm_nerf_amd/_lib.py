"""ctypes binding of libdmnerf_hip.so (the C ABI declared in include/dmnerf_hip.h).

PyTorch is plumbing only: it owns device memory and streams; every pointer handed to the
library is ``tensor.data_ptr()`` and every launch goes to ``torch.cuda.current_stream()``.
"""
import ctypes
import os

import torch  # must be imported first: the library binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdmnerf_hip.so")

c_f32p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_vp = ctypes.c_void_p
c_float = ctypes.c_float


class RenderArgs(ctypes.Structure):
    """``dmnerf_render_args`` (include/dmnerf_hip.h)."""
    _fields_ = [
        ("d_blob_coarse", c_vp), ("d_blob_fine", c_vp), ("ins_num", c_int),
        ("d_rays_o", c_vp), ("d_rays_d", c_vp), ("d_z_in", c_vp), ("d_t_rand", c_vp), ("d_u", c_vp),
        ("u_row_stride", c_i64), ("N", c_i64), ("S", c_int), ("n_imp", c_int),
        ("d_z_coarse", c_vp), ("d_raw_coarse", c_vp), ("d_rgb_coarse", c_vp), ("d_depth_coarse", c_vp),
        ("d_ins_coarse", c_vp), ("d_z_fine", c_vp), ("d_raw_fine", c_vp), ("d_rgb_fine", c_vp),
        ("d_depth_fine", c_vp), ("d_ins_fine", c_vp), ("d_weights_ws", c_vp),
        ("ev_fine_mlp_begin", c_vp), ("ev_fine_mlp_end", c_vp), ("fused_heads", c_int),
    ]


class RepackModel(ctypes.Structure):
    """``dmnerf_repack_model`` (include/dmnerf_hip.h)."""
    _fields_ = [("d_params_flat", c_vp), ("ins_num", c_int), ("d_flat_copy", c_vp), ("d_idx", c_vp), ("d_blob", c_vp),
                ("d_idx_t", c_vp), ("d_blob_t", c_vp), ("d_f_pos", c_vp)]


class ChainLayer(ctypes.Structure):
    """``dmnerf_chain_layer`` (include/dmnerf_hip.h)."""
    _fields_ = [("d_B", c_vp), ("d_bias", c_vp), ("ldb", c_int), ("from_act", c_int), ("from_x", c_int), ("relu", c_int)]


c_double = ctypes.c_double

# name -> (restype, argtypes); exactly the symbols include/dmnerf_hip.h declares
SIGNATURES = {
    "dmnerf_adam_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_double, c_vp, c_double, c_double, c_double, c_vp, c_vp]),
    "dmnerf_repack_train": (c_int, [ctypes.POINTER(RepackModel), c_int, c_vp]),
    "dmnerf_abi_version": (c_int, []),
    "dmnerf_last_error": (ctypes.c_char_p, []),
    "dmnerf_device_count": (c_int, []),
    "dmnerf_param_count": (c_i64, [c_int]),
    "dmnerf_blob_floats": (c_i64, [c_int]),
    "dmnerf_build_pack_index": (c_int, [c_int, c_vp, c_i64]),
    "dmnerf_pack_weights": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dmnerf_raygen": (c_int, [c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_raygen_select": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dmnerf_z_val_sample": (c_int, [c_vp, c_float, c_float, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_stratify": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_sample_pdf": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_sample_from_cdf": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_importance_resample": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_embed": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_mlp_fwd_embedded": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp, c_vp]),
    "dmnerf_mlp_fwd_rays": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_composite_fwd": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_render_rays_fwd": (c_int, [ctypes.POINTER(RenderArgs), c_vp]),
    "dmnerf_composite_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    "dmnerf_train_save_floats": (c_i64, [c_i64]),
    "dmnerf_mlp_fwd_embedded_train": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dmnerf_mlp_fwd_rays_train_split": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_mlp_fwd_rays_train_fused": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_mlp_fwd_rays_train": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_blob_t_floats": (c_i64, [c_int]),
    "dmnerf_build_pack_index_t": (c_int, [c_int, c_vp, c_i64]),
    "dmnerf_mlp_bwd_data": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dmnerf_manipulator_render": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_z_val_lerp": (c_int, [c_vp, c_float, c_float, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_sort_rows": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_ins_label_conf": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_exchanger": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_penalizer_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_float, c_float, c_float, c_vp, c_vp]),
    "dmnerf_penalizer_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_float, c_float, c_float, c_vp, c_vp, c_vp]),
    "dmnerf_penalizer_sums": (c_int, [c_vp, c_i64, c_vp, c_vp]),
    "dmnerf_penalizer_finish": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_f16x2_range_flags": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_penalizer_sums2": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dmnerf_loss_tail_fwd": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_loss_tail_bwd": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_composite_pen_fwd": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_float, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_composite_pen_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_float, c_float, c_float, c_vp, c_vp]),
    "dmnerf_wgrad_plan_sizes": (c_int, [c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_wgrad_plan": (c_int, [c_int, c_i64, c_int, c_vp, c_i64, c_vp, c_i64]),
    "dmnerf_mlp_bwd_weights": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_wgrad_plan_sizes_split": (c_int, [c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_wgrad_plan_split": (c_int, [c_int, c_i64, c_int, c_vp, c_i64, c_vp, c_i64]),
    "dmnerf_wgrad_plan_sizes_f16": (c_int, [c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_wgrad_plan_f16": (c_int, [c_int, c_i64, c_int, c_vp, c_i64, c_vp, c_i64]),
    "dmnerf_mlp_bwd_weights_split": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_head_product": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "dmnerf_fuse_heads": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "dmnerf_gemm": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_int, c_i64, c_vp, c_int, c_vp, c_i64, c_int, c_vp, c_int, c_vp]),
    "dmnerf_colsum": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_int, c_vp]),
    "dmnerf_ray_points": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_copy_cols": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp]),
    "dmnerf_gemm_nt_blocks": (c_int, [c_int]),
    "dmnerf_pack_nt": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "dmnerf_gemm_nt": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_int, c_int,
                               c_i64, c_int, c_vp, c_i64, c_int, c_vp]),
    "dmnerf_gemm_tn_ws_floats": (c_i64, [c_int, c_int, c_i64]),
    "dmnerf_gemm_tn": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_i64, c_int, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "dmnerf_mlp_chain_supported": (c_int, [c_int, c_int]),
    "dmnerf_mlp_chain": (c_int, [c_vp, c_i64, c_i64, c_int, ctypes.POINTER(ChainLayer), c_int, c_int, c_vp, c_i64, c_i64, c_vp]),
    "dmnerf_copy_cols_pad": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_vp]),
    "dmnerf_ray_embed": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp]),
    "dmnerf_wgrad_set_trace": (c_int, [c_vp]),
    "dmnerf_blob_split_words": (c_i64, [c_int]),
    "dmnerf_build_pack_index_split": (c_int, [c_int, c_vp, c_i64]),
    "dmnerf_pack_split": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dmnerf_blob_t_split_words": (c_i64, [c_int]),
    "dmnerf_build_pack_index_t_split": (c_int, [c_int, c_vp, c_i64]),
    "dmnerf_mlp_bwd_data_split": (c_int, [c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dmnerf_mlp_fwd_rays_split": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_blob_f16_words": (c_i64, [c_int]),
    "dmnerf_build_pack_index_f16": (c_int, [c_int, c_vp, c_i64, c_vp, c_i64]),
    "dmnerf_pack_f16": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dmnerf_mlp_fwd_rays_f16": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_blob_t_f16_words": (c_i64, [c_int]),
    "dmnerf_build_pack_index_t_f16": (c_int, [c_int, c_vp, c_i64]),
    "dmnerf_grad_scale": (c_int, [c_vp, c_i64, c_vp, c_vp]),
    "dmnerf_mlp_bwd_data_f16": (c_int, [c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_mlp_bwd_weights_f16": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_mlp_fwd_rays_train_f16": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "dmnerf_blob_fused_floats": (c_i64, [c_int]),
    "dmnerf_build_pack_index_fused": (c_int, [c_int, c_vp, c_i64]),
    "dmnerf_mlp_fwd_rays_fused": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    "dmnerf_ins_criterion_work_bytes": (c_i64, [c_i64, c_int]),
    "dmnerf_ins_criterion_flags_offset": (c_i64, [c_i64, c_int]),
    "dmnerf_ins_criterion_fwd": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_vp]),
    "dmnerf_ins_criterion_bwd": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]),
    "dmnerf_ins_criterion_fwd2": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dmnerf_ins_criterion_bwd2": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
}

_lib = None


def load():
    """Load the shared library (idempotent).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C dm_nerf_amd/csrc`).  dm_nerf_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.dmnerf_abi_version() != 8:
        raise RuntimeError("libdmnerf_hip.so ABI version mismatch")
    _lib = lib
    return lib


def last_error():
    return load().dmnerf_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    """Validate device / dtype / contiguity before handing raw pointers to the library."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("dm_nerf_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("dm_nerf_amd: tensor must be contiguous")


def f32(t):
    """float32 + contiguous view/copy of a GPU tensor (plumbing)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
