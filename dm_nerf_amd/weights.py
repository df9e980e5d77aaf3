"""Reference state_dict -> kernel weight blob (MFMA A-operand order)."""
import ctypes

import numpy as np
import torch

from . import _lib

# DM_NeRF.__init__ / state_dict order (networks/dm_nerf.py:59-78 of the reference)
PARAM_MODULES = [f"mlps.{i}" for i in range(8)] + [
    "rgb_feature_linear", "ins_feature_linear", "rgb_feature_linears.0", "ins_feature_linears.0",
    "density_linear", "ins_linear", "rgb_linear"]
PARAM_KEYS = [f"{m}.{p}" for m in PARAM_MODULES for p in ("weight", "bias")]

_index_cache = {}


def pack_index_host(ins_num):
    """int32 numpy gather index (host only; works without a GPU)."""
    lib = _lib.load()
    n = lib.dmnerf_blob_floats(ins_num)
    if n <= 0:
        raise ValueError(f"unsupported ins_num={ins_num}")
    idx = np.empty(n, dtype=np.int32)
    _lib.check(lib.dmnerf_build_pack_index(ins_num, idx.ctypes.data_as(ctypes.c_void_p), n), "dmnerf_build_pack_index")
    return idx


def pack_index_t_host(ins_num):
    """Gather index of the backward (W^T) blob."""
    lib = _lib.load()
    n = lib.dmnerf_blob_t_floats(ins_num)
    if n <= 0:
        raise ValueError(f"unsupported ins_num={ins_num}")
    idx = np.empty(n, dtype=np.int32)
    _lib.check(lib.dmnerf_build_pack_index_t(ins_num, idx.ctypes.data_as(ctypes.c_void_p), n), "dmnerf_build_pack_index_t")
    return idx


def pack_index_fused_host(ins_num):
    """Gather index of the fused-heads inference blob."""
    lib = _lib.load()
    n = lib.dmnerf_blob_fused_floats(ins_num)
    if n <= 0:
        raise ValueError(f"unsupported ins_num={ins_num}")
    idx = np.empty(n, dtype=np.int32)
    _lib.check(lib.dmnerf_build_pack_index_fused(ins_num, idx.ctypes.data_as(ctypes.c_void_p), n), "dmnerf_build_pack_index_fused")
    return idx


def pack_index(ins_num, device, transposed=False, fused=False):
    key = (ins_num, str(device), transposed, fused)
    if key not in _index_cache:
        host = pack_index_t_host(ins_num) if transposed else (pack_index_fused_host(ins_num) if fused else pack_index_host(ins_num))
        _index_cache[key] = torch.from_numpy(host).to(device)
    return _index_cache[key]


def fused_flat(flat, ins_num):
    """The flat parameter vector with the activation-free ``rgb_feature_linear`` / ``ins_feature_linear`` (dm_nerf.py:89,96)
    folded into the hidden layers that consume them -- formed on the device by ``dmnerf_fuse_heads`` (float64 accumulation,
    rounded once; csrc/heads.hip).  Inference only: the same function up to f32 re-association."""
    out = torch.empty_like(flat)
    _lib.check(_lib.load().dmnerf_fuse_heads(_lib.ptr(flat), ins_num, _lib.ptr(out), _lib.stream()), "dmnerf_fuse_heads")
    return out


def flat_params(state):
    """Concatenate parameters in reference order.  ``state``: mapping key -> tensor (all on one GPU)."""
    return torch.cat([state[k].detach().reshape(-1).float() for k in PARAM_KEYS])


def pack_blob_split(state, ins_num):
    """The opt-in split-bf16 inference blob: [table of the fused f32 blob | three bf16 planes of every weight]."""
    lib = _lib.load()
    flat = flat_params(state)
    _lib.require_gpu(flat)
    flat = fused_flat(flat, ins_num)
    total = lib.dmnerf_blob_split_words(ins_num)
    if total <= 0:
        raise ValueError(f"unsupported ins_num={ins_num}")
    tab = 4096
    key = (ins_num, str(flat.device), "split")
    if key not in _index_cache:
        n = (total - tab) * 2
        host = np.empty(n, dtype=np.int32)
        _lib.check(lib.dmnerf_build_pack_index_split(ins_num, host.ctypes.data_as(ctypes.c_void_p), n), "dmnerf_build_pack_index_split")
        _index_cache[key] = torch.from_numpy(host).to(flat.device)
    idx = _index_cache[key]
    blob = torch.empty(total, dtype=torch.float32, device=flat.device)
    idx_tab = pack_index(ins_num, flat.device, False, True)[:tab].contiguous()
    _lib.check(lib.dmnerf_pack_weights(_lib.ptr(flat), _lib.ptr(idx_tab), _lib.ptr(blob), tab, _lib.stream()), "dmnerf_pack_weights")
    _lib.check(lib.dmnerf_pack_split(_lib.ptr(flat), _lib.ptr(idx), _lib.ptr(blob[tab:]), total - tab, _lib.stream()), "dmnerf_pack_split")
    return blob


def split_mode(args):
    """``args.mfma_split`` -> None (off) | "bf16x3" (True or "bf16x3": three bf16 planes, six products) | "f16x2" (two f16
    planes, three products).  Both are f32-class, neither is the bitwise fmaf chain of the default kernels."""
    m = getattr(args, "mfma_split", False)
    if not m:
        return None
    if m is True or str(m) == "bf16x3":
        return "bf16x3"
    if str(m) == "f16x2":
        return "f16x2"
    raise ValueError(f"args.mfma_split = {m!r}: expected False, True / 'bf16x3' or 'f16x2'")


def pack_blob_f16(state, ins_num):
    """The opt-in split-f16 inference blob: [bias table f32 | two f16 planes of every (fused-heads) weight, 16 KiB groups]."""
    lib = _lib.load()
    flat = flat_params(state)
    _lib.require_gpu(flat)
    flat = fused_flat(flat, ins_num)
    total = lib.dmnerf_blob_f16_words(ins_num)
    if total <= 0:
        raise ValueError(f"unsupported ins_num={ins_num}")
    tab = 4096
    key = (ins_num, str(flat.device), "f16")
    if key not in _index_cache:
        n = (total - tab) * 2
        h_tab, h_str = np.empty(tab, dtype=np.int32), np.empty(n, dtype=np.int32)
        _lib.check(lib.dmnerf_build_pack_index_f16(ins_num, h_tab.ctypes.data_as(ctypes.c_void_p), tab,
                                                   h_str.ctypes.data_as(ctypes.c_void_p), n), "dmnerf_build_pack_index_f16")
        _index_cache[key] = (torch.from_numpy(h_tab).to(flat.device), torch.from_numpy(h_str).to(flat.device))
    idx_tab, idx = _index_cache[key]
    blob = torch.empty(total, dtype=torch.float32, device=flat.device)
    _lib.check(lib.dmnerf_pack_weights(_lib.ptr(flat), _lib.ptr(idx_tab), _lib.ptr(blob), tab, _lib.stream()), "dmnerf_pack_weights")
    _lib.check(lib.dmnerf_pack_f16(_lib.ptr(flat), _lib.ptr(idx), _lib.ptr(blob[tab:]), total - tab, _lib.stream()), "dmnerf_pack_f16")
    return blob


HEAD_F_FLOATS = 128 * 256        # layout.h::HEAD_F_FLOATS
TAB_T_FLOATS = 1024              # layout.h::TAB_T_FLOATS


def pack_blob_t_split(flat, ins_num):
    """The opt-in split-bf16 data-gradient blob: [VALU-head table of the W^T blob | three bf16 planes of every W^T]."""
    lib = _lib.load()
    _lib.require_gpu(flat)
    total = lib.dmnerf_blob_t_split_words(ins_num)
    if total <= 0:
        raise ValueError(f"unsupported ins_num={ins_num}")
    ext = torch.empty(flat.numel() + HEAD_F_FLOATS, dtype=torch.float32, device=flat.device)
    ext[:flat.numel()] = flat
    _lib.check(lib.dmnerf_head_product(_lib.ptr(flat), ins_num, _lib.ptr(ext[flat.numel():]), _lib.stream()), "dmnerf_head_product")
    key = (ins_num, str(flat.device), "split_t")
    if key not in _index_cache:
        n = (total - TAB_T_FLOATS) * 2
        host = np.empty(n, dtype=np.int32)
        _lib.check(lib.dmnerf_build_pack_index_t_split(ins_num, host.ctypes.data_as(ctypes.c_void_p), n), "dmnerf_build_pack_index_t_split")
        _index_cache[key] = torch.from_numpy(host).to(flat.device)
    idx = _index_cache[key]
    blob = torch.empty(total, dtype=torch.float32, device=flat.device)
    idx_tab = pack_index(ins_num, flat.device, True)[:TAB_T_FLOATS].contiguous()
    _lib.check(lib.dmnerf_pack_weights(_lib.ptr(ext), _lib.ptr(idx_tab), _lib.ptr(blob), TAB_T_FLOATS, _lib.stream()), "dmnerf_pack_weights")
    _lib.check(lib.dmnerf_pack_split(_lib.ptr(ext), _lib.ptr(idx), _lib.ptr(blob[TAB_T_FLOATS:]), total - TAB_T_FLOATS, _lib.stream()), "dmnerf_pack_split")
    return blob


def pack_blob_t_f16(flat, ins_num):
    """The opt-in split-f16 data-gradient blob: [VALU-head table of the W^T blob | two f16 planes of every W^T, 16 KiB groups]."""
    lib = _lib.load()
    _lib.require_gpu(flat)
    total = lib.dmnerf_blob_t_f16_words(ins_num)
    if total <= 0:
        raise ValueError(f"unsupported ins_num={ins_num}")
    ext = torch.empty(flat.numel() + HEAD_F_FLOATS, dtype=torch.float32, device=flat.device)
    ext[:flat.numel()] = flat
    _lib.check(lib.dmnerf_head_product(_lib.ptr(flat), ins_num, _lib.ptr(ext[flat.numel():]), _lib.stream()), "dmnerf_head_product")
    key = (ins_num, str(flat.device), "f16_t")
    if key not in _index_cache:
        n = (total - TAB_T_FLOATS) * 2
        host = np.empty(n, dtype=np.int32)
        _lib.check(lib.dmnerf_build_pack_index_t_f16(ins_num, host.ctypes.data_as(ctypes.c_void_p), n), "dmnerf_build_pack_index_t_f16")
        _index_cache[key] = torch.from_numpy(host).to(flat.device)
    idx = _index_cache[key]
    blob = torch.empty(total, dtype=torch.float32, device=flat.device)
    idx_tab = pack_index(ins_num, flat.device, True)[:TAB_T_FLOATS].contiguous()
    _lib.check(lib.dmnerf_pack_weights(_lib.ptr(ext), _lib.ptr(idx_tab), _lib.ptr(blob), TAB_T_FLOATS, _lib.stream()), "dmnerf_pack_weights")
    _lib.check(lib.dmnerf_pack_f16(_lib.ptr(ext), _lib.ptr(idx), _lib.ptr(blob[TAB_T_FLOATS:]), total - TAB_T_FLOATS, _lib.stream()), "dmnerf_pack_f16")
    return blob


def pack_blob(state, ins_num, out=None, transposed=False, fused=False, flat=None):
    """Build (or refresh in place) the kernel blob for one DM_NeRF model (``transposed``: the W^T
    blob of the backward data-gradient kernel; ``fused``: the inference blob with the feature linears folded
    into the hidden layers).  ``flat``: the flat parameter vector if the caller already has it."""
    lib = _lib.load()
    flat = flat_params(state) if flat is None else flat
    if flat.numel() != lib.dmnerf_param_count(ins_num):
        raise ValueError(f"parameter count {flat.numel()} != {lib.dmnerf_param_count(ins_num)} "
                         f"(only D=8, W=256, skips=[4], 63+27 input channels are supported)")
    _lib.require_gpu(flat)
    if fused:
        flat = fused_flat(flat, ins_num)
    if transposed:
        # the W^T blob also carries F = rgb_feature_linears.0.weight[:, :256] . rgb_feature_linear.weight (csrc/heads.hip):
        # formed on the device behind the flat parameters, gathered like any other weight
        ext = torch.empty(flat.numel() + HEAD_F_FLOATS, dtype=torch.float32, device=flat.device)
        ext[:flat.numel()] = flat
        _lib.check(lib.dmnerf_head_product(_lib.ptr(flat), ins_num, _lib.ptr(ext[flat.numel():]), _lib.stream()), "dmnerf_head_product")
        flat = ext
    idx = pack_index(ins_num, flat.device, transposed, fused)
    n = idx.numel()
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=flat.device)
    _lib.check(lib.dmnerf_pack_weights(_lib.ptr(flat), _lib.ptr(idx), _lib.ptr(out), n, _lib.stream()), "dmnerf_pack_weights")
    return out
