"""Drop-in for the render core of the reference's ``networks/manipulator.py``: ``exchanger``,
``manipulator_render``, ``manipulator_nerf``, ``manipulator``.  The evaluation / demo drivers (image IO, LPIPS,
pose JSON) of that file are out of scope; they call exactly these four functions."""
import ctypes

import torch

from .. import _lib
from . import helpers
from .render import run_network


def exchanger(ori_raw, tar_raws, ori_raw_pred, tar_raw_preds, move_labels):
    """``exchanger`` (networks/manipulator.py:18-83).  ``ori_raw`` is modified in place, as in the reference."""
    lib = _lib.load()
    _lib.require_gpu(ori_raw, ori_raw_pred, *tar_raws, *tar_raw_preds)
    N, S, ch = ori_raw.shape
    C = ch - 4
    T = len(move_labels)
    tr = [_lib.f32(t) for t in tar_raws]
    ta = [_lib.f32(t) for t in tar_raw_preds]
    oa = _lib.f32(ori_raw_pred)
    P = ctypes.c_void_p * T
    raws, accs = P(*[t.data_ptr() for t in tr]), P(*[t.data_ptr() for t in ta])
    labels = (ctypes.c_int * T)(*[int(v) for v in move_labels])
    ori_label = torch.empty(N, S, dtype=torch.int64, device=ori_raw.device)
    tar_label = torch.empty(N, S, dtype=torch.int64, device=ori_raw.device)
    _lib.check(lib.dmnerf_exchanger(_lib.ptr(ori_raw), raws, _lib.ptr(oa), accs, labels, T, N, S, C,
                                    _lib.ptr(ori_label), _lib.ptr(tar_label), _lib.stream()), "dmnerf_exchanger")
    return ori_raw, tar_raws, ori_label, tar_label


def manipulator_render(raw, z_vals, rays_d):
    """``manipulator_render`` (networks/manipulator.py:86-105) -> (rgb_map, weights, depth_map, ins_map [N,C])."""
    raw, z, d = _lib.f32(raw), _lib.f32(z_vals), _lib.f32(rays_d)
    _lib.require_gpu(raw, z, d)
    N, S, ch = raw.shape
    C = ch - 4
    f = dict(dtype=torch.float32, device=raw.device)
    rgb, w, depth, ins = torch.empty(N, 3, **f), torch.empty(N, S, **f), torch.empty(N, **f), torch.empty(N, C, **f)
    _lib.check(_lib.load().dmnerf_manipulator_render(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(d), N, S, C, _lib.ptr(rgb), _lib.ptr(w),
                                                     _lib.ptr(depth), _lib.ptr(ins), _lib.stream()), "dmnerf_manipulator_render")
    return rgb, w, depth, ins


def manipulator_z(N_rays, near, far, N_samples, device=None):
    """The depth grid of ``manipulator_nerf`` (:117-119): ``near (1 - t) + far t``."""
    dev = helpers._device(device)
    t = helpers.linspace01(N_samples, dev)
    z = torch.empty(int(N_rays), int(N_samples), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().dmnerf_z_val_lerp(_lib.ptr(t), float(near), float(far), int(N_rays), int(N_samples), _lib.ptr(z),
                                             _lib.stream()), "dmnerf_z_val_lerp")
    return z


def manipulator_nerf(rays, position_embedder, view_embedder, model, N_samples=None, near=None, far=None, z_vals=None, split=None):
    """``manipulator_nerf`` (networks/manipulator.py:108-134) -> (raw [N,S,4+C], z_vals).  ``split`` (extension): the opt-in
    split-operand network kernels, ``weights.split_mode(args)``."""
    rays_o, rays_d = rays
    if z_vals is None:
        z_vals = manipulator_z(rays_d.shape[0], near, far, N_samples, rays_d.device)
    with torch.no_grad():
        raw = run_network(model, rays_o, rays_d, z_vals, split=split)
    return raw, z_vals


def sort_rows(x):
    """``torch.sort(x, -1).values`` (exact permutation)."""
    x = _lib.f32(x)
    _lib.require_gpu(x)
    out = torch.empty_like(x)
    _lib.check(_lib.load().dmnerf_sort_rows(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(out), _lib.stream()), "dmnerf_sort_rows")
    return out


def manipulator(position_embedder, view_embedder, model_coarse, model_fine, ori_rays, f_tar_rays, args, us=None):
    """``manipulator`` (networks/manipulator.py:137-205) -> (final_rgb, final_ins, tar_rgb, tar_ins_accum).

    RNG: the reference calls ``sample_pdf(..., det=False)`` even at evaluation (:148,:170,:187): ``2 + T`` draws
    of ``torch.rand([N, N_importance])`` in the order original, each target, original again; the same draws are
    made here on the rays' device, or pass them as ``us`` (extension used by the tests).  ``args.mfma_split`` (extension, default
    off) evaluates the 3 + 4 T network launches on the opt-in split-operand kernels, as in ``dm_nerf``.
    """
    from .. import weights
    split = weights.split_mode(args)
    N_samples, N_importance, near, far = args.N_samples, args.N_importance, args.near, args.far
    dev = ori_rays.device
    Nr = ori_rays.shape[1]
    us = list(us) if us is not None else None
    draw = lambda: _lib.f32(us.pop(0)) if us is not None else torch.rand([Nr, N_importance], device=dev)
    pe, ve = position_embedder, view_embedder
    ori_raw, ori_z = manipulator_nerf(ori_rays, pe, ve, model_coarse, N_samples, near, far, split=split)
    _, ori_w, _, _ = manipulator_render(ori_raw, ori_z, ori_rays[1])
    ori_z_full = helpers.importance_resample(ori_z, ori_w, N_importance, u=draw())
    ori_raw_full, _ = manipulator_nerf(ori_rays, pe, ve, model_fine, z_vals=ori_z_full, split=split)
    _, _, _, ori_ins_accum = manipulator_render(ori_raw_full, ori_z_full, ori_rays[1])
    tar_raws, f_tar_z, f_tar_zs, tar_ins_accums = [], [], [], []
    tar_rgb = tar_ins_accum = None
    for tar_rays in f_tar_rays:
        tar_raw, tar_z = manipulator_nerf(tar_rays, pe, ve, model_coarse, N_samples, near, far, split=split)
        tar_raws.append(tar_raw); f_tar_z.append(tar_z)
        tar_rgb, tar_w, _, _ = manipulator_render(tar_raw, tar_z, tar_rays[1])
        tar_z_full, tar_zs = helpers.importance_resample(tar_z, tar_w, N_importance, u=draw(), return_samples=True)
        tar_raw_full, _ = manipulator_nerf(tar_rays, pe, ve, model_fine, z_vals=tar_z_full, split=split)
        _, _, _, tar_ins_accum = manipulator_render(tar_raw_full, tar_z_full, tar_rays[1])
        f_tar_zs.append(tar_zs); tar_ins_accums.append(tar_ins_accum)
    ori_raw, _, _, _ = exchanger(ori_raw, tar_raws, ori_ins_accum, tar_ins_accums, args.target_labels)
    # step 2: re-render the edited coarse field, resample, and evaluate the fine model on the merged depths
    _, ori_w, _, _ = manipulator_render(ori_raw, ori_z, ori_rays[1])
    _, ori_zs = helpers.importance_resample(ori_z, ori_w, N_importance, u=draw(), return_samples=True)
    f_tar_zs = torch.cat(f_tar_zs, dim=-1)
    ori_z = sort_rows(torch.cat([ori_z, ori_zs, f_tar_zs], dim=-1))
    for idx, tar_rays in enumerate(f_tar_rays):
        ori_raw, ori_z = manipulator_nerf(ori_rays, pe, ve, model_fine, z_vals=ori_z, split=split)
        tar_z = sort_rows(torch.cat([f_tar_z[idx], ori_zs, f_tar_zs], dim=-1))
        tar_raws[idx], _ = manipulator_nerf(tar_rays, pe, ve, model_fine, z_vals=tar_z, split=split)
    ori_raw, _, _, _ = exchanger(ori_raw, tar_raws, ori_ins_accum, tar_ins_accums, args.target_labels)
    final_rgb, _, _, final_ins = manipulator_render(ori_raw, ori_z, ori_rays[1])
    return final_rgb, final_ins, tar_rgb, tar_ins_accum
