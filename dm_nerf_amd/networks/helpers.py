"""Drop-in for the hot-path functions of the reference's ``networks/helpers.py``."""
import ctypes

import numpy as np
import torch

from .. import _lib

_const_cache = {}


def _device(device):
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def linspace01(steps, device):
    """``torch.linspace(0, 1, steps)`` evaluated on the CPU (the reference's values, bit for bit)
    and cached on ``device``; ROCm's own linspace kernel may round differently."""
    key = (int(steps), str(device))
    if key not in _const_cache:
        _const_cache[key] = torch.linspace(0., 1., steps=int(steps)).to(device)
    return _const_cache[key]


def get_rays_k(H, W, K, c2w, row0=0, nrows=None):
    """``get_rays_k`` (networks/helpers.py:50-61) -> ``rays_o, rays_d`` of shape [H, W, 3].

    ``K``: numpy 3x3 / 4x4 intrinsics, ``c2w``: [3or4, 4] tensor (any device; 12 floats are read on
    the host).  ``row0/nrows`` (extension) generate only a band of rows -> [nrows, W, 3]: this is
    how ranks shard a frame without a scatter.
    """
    H, W = int(H), int(W)
    nrows = H - row0 if nrows is None else int(nrows)
    dev = c2w.device if torch.is_tensor(c2w) and c2w.is_cuda else _device(None)
    K = np.asarray(K)
    intr = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[2, 2]], dtype=np.float64).astype(np.float32)
    c = (c2w.detach().cpu().numpy() if torch.is_tensor(c2w) else np.asarray(c2w)).astype(np.float32)[:3, :4]
    c = np.ascontiguousarray(c)
    rays_o = torch.empty(nrows, W, 3, dtype=torch.float32, device=dev)
    rays_d = torch.empty(nrows, W, 3, dtype=torch.float32, device=dev)
    if nrows == 0:                                   # an empty band (more ranks than rows): nothing to generate
        return rays_o, rays_d
    _lib.check(_lib.load().dmnerf_raygen(H, W, intr.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p),
                                         int(row0), nrows, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.stream()), "dmnerf_raygen")
    return rays_o, rays_d


def _camera(K, c2w):
    K = np.asarray(K)
    intr = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[2, 2]], dtype=np.float64).astype(np.float32)
    c = (c2w.detach().cpu().numpy() if torch.is_tensor(c2w) else np.asarray(c2w)).astype(np.float32)[:3, :4]
    return intr, np.ascontiguousarray(c)


def get_select_full(rgb, pose, K, ins_target, N_train):
    """``get_select_full`` (networks/helpers.py:99-111): a random batch of ``N_train`` pixels of one image.

    Same host RNG stream as the reference -- one ``np.random.choice(H*W, N_train, replace=False)`` -- but only
    the selected rays are generated (the reference builds all H*W rays every step and gathers).
    Returns ``target_c [N,3], target_i [N], batch_rays [2,N,3]`` on the image's device.
    """
    H, W, _ = rgb.shape
    _lib.require_gpu(rgb.contiguous())
    selected_index = np.random.choice(H * W, size=[N_train], replace=False)
    idx = torch.from_numpy(selected_index).to(rgb.device)
    intr, c = _camera(K, pose)
    rays = torch.empty(2, N_train, 3, dtype=torch.float32, device=rgb.device)
    _lib.check(_lib.load().dmnerf_raygen_select(int(H), int(W), intr.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p),
                                                _lib.ptr(idx), int(N_train), _lib.ptr(rays[0]), _lib.ptr(rays[1]), _lib.stream()),
               "dmnerf_raygen_select")
    target_c = rgb.reshape(-1, rgb.shape[-1])[idx]
    target_i = ins_target.reshape(-1)[idx]
    return target_c, target_i, rays


def get_select_crop(rgb, pose, K, ins_target, ins_index, crop_mask, N_train):
    """``get_select_crop`` (networks/helpers.py:64-95, the ScanNet batch): 30 % of the batch from the labelled
    pixels ``ins_index`` (flat indices, numpy), the rest from the central crop ``crop_mask`` (numpy, 1 = inside).

    The host RNG stream is the reference's: ``choice(len(ins_index), N_ins)`` then
    ``choice(|crop \\ labelled|, N_rgb)`` -- and, like the reference (:82-84), the second draw indexes the crop
    pixel list itself, not the set difference.  Only the selected rays are generated.
    Returns ``target_c [N_train,3], target_i [N_ins], batch_rays [2,N_train,3], N_ins``; unlabelled rays first.
    """
    H, W, _ = rgb.shape
    _lib.require_gpu(rgb.contiguous())
    ins_index = np.asarray(ins_index)
    N_ins = min(int(N_train * 0.3), len(ins_index))
    N_rgb = N_train - N_ins
    crop_indices = np.where(np.asarray(crop_mask).reshape(-1) == 1)[0]
    labeled_idx = ins_index[np.random.choice(ins_index.shape[0], size=[N_ins], replace=False)]
    # |set(crop) - set(labeled)| (helpers.py:81) without building two 300 k-element Python sets per step: the labelled
    # pixels are distinct (drawn without replacement from np.where output), so it is |crop| minus those inside the crop
    n_unlabeled = len(crop_indices) - int((np.asarray(crop_mask).reshape(-1)[labeled_idx] == 1).sum())
    unlabeled_idx = crop_indices[np.random.choice(n_unlabeled, size=[N_rgb], replace=False)]
    flat = np.concatenate([unlabeled_idx, labeled_idx]).astype(np.int64)
    idx = torch.from_numpy(flat).to(rgb.device)
    intr, c = _camera(K, pose)
    rays = torch.empty(2, N_train, 3, dtype=torch.float32, device=rgb.device)
    _lib.check(_lib.load().dmnerf_raygen_select(int(H), int(W), intr.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p),
                                                _lib.ptr(idx), int(N_train), _lib.ptr(rays[0]), _lib.ptr(rays[1]), _lib.stream()),
               "dmnerf_raygen_select")
    target_c = rgb.reshape(-1, rgb.shape[-1])[idx]
    target_i = ins_target.reshape(-1)[idx[N_rgb:]]
    return target_c, target_i, rays, N_ins


def _check_u(u, N, n, what):
    """``u``: the inverse-CDF draw, ``[n]`` (shared by all rays) or ``[N, n]``; returns (f32 contiguous u, row stride)."""
    if tuple(u.shape) not in ((n,), (N, n)):
        raise ValueError(f"{what}: u must be [{n}] or [{N}, {n}], got {tuple(u.shape)}")
    u = _lib.f32(u)
    _lib.require_gpu(u)
    return u, (0 if u.dim() == 1 else n)


def z_val_sample(N_rays, near, far, N_samples, device=None):
    """``z_val_sample`` (networks/helpers.py:114-119): ``near + linspace(0,1,S) * (far - near)`` -> [N, S]."""
    dev = _device(device)
    t = linspace01(N_samples, dev)
    z = torch.empty(int(N_rays), int(N_samples), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().dmnerf_z_val_sample(_lib.ptr(t), float(near), float(far), int(N_rays), int(N_samples),
                                               _lib.ptr(z), _lib.stream()), "dmnerf_z_val_sample")
    return z


def sample_pdf(bins, weights, N_samples, det=False, u=None, return_aux=False):
    """``sample_pdf`` (networks/helpers.py:123-155).  ``u`` (extension) pins the random draw."""
    bins, weights = _lib.f32(bins), _lib.f32(weights)
    _lib.require_gpu(bins, weights)
    N, nb = bins.shape
    if weights.shape != (N, nb - 1):
        raise ValueError("sample_pdf: weights must be [N, bins-1]")
    if u is None:
        u = linspace01(N_samples, bins.device) if det else torch.rand([N, N_samples], device=bins.device)
    u, stride = _check_u(u, N, int(N_samples), "sample_pdf")
    samples = torch.empty(N, N_samples, dtype=torch.float32, device=bins.device)
    cdf = torch.empty(N, nb, dtype=torch.float32, device=bins.device) if return_aux else None
    inds = torch.empty(N, N_samples, dtype=torch.int64, device=bins.device) if return_aux else None
    _lib.check(_lib.load().dmnerf_sample_pdf(_lib.ptr(bins), _lib.ptr(weights), _lib.ptr(u), stride, N, nb, int(N_samples),
                                             _lib.ptr(samples), _lib.ptr(cdf), _lib.ptr(inds), _lib.stream()), "dmnerf_sample_pdf")
    if return_aux:
        return samples, cdf, inds
    return samples


def sample_from_cdf(bins, cdf, u):
    """Stage-isolated inverse-CDF step (helpers.py:139-153) -> (samples, inds int64)."""
    bins, cdf, u = _lib.f32(bins), _lib.f32(cdf), _lib.f32(u)
    _lib.require_gpu(bins, cdf, u)
    N, nb = bins.shape
    ns = u.shape[-1]
    u, stride = _check_u(u, N, ns, "sample_from_cdf")
    if cdf.shape != bins.shape:
        raise ValueError("sample_from_cdf: cdf and bins must both be [N, n_bins]")
    samples = torch.empty(N, ns, dtype=torch.float32, device=bins.device)
    inds = torch.empty(N, ns, dtype=torch.int64, device=bins.device)
    _lib.check(_lib.load().dmnerf_sample_from_cdf(_lib.ptr(bins), _lib.ptr(cdf), _lib.ptr(u), stride, N, nb, ns,
                                                  _lib.ptr(samples), _lib.ptr(inds), _lib.stream()), "dmnerf_sample_from_cdf")
    return samples, inds


def stratify(z_vals, t_rand):
    """Stratified jitter of render.py:42-47 with the draw passed in."""
    z, t = _lib.f32(z_vals), _lib.f32(t_rand)
    _lib.require_gpu(z, t)
    if t.shape != z.shape:
        raise ValueError(f"stratify: t_rand must have z_vals' shape {tuple(z.shape)}, got {tuple(t.shape)}")
    out = torch.empty_like(z)
    _lib.check(_lib.load().dmnerf_stratify(_lib.ptr(z), _lib.ptr(t), z.shape[0], z.shape[1], _lib.ptr(out), _lib.stream()), "dmnerf_stratify")
    return out


def importance_resample(z_coarse, weights_coarse, N_importance, det=True, u=None, return_samples=False):
    """render.py:66-70 fused: z_mid, sample_pdf(z_mid, w[...,1:-1]), sort(cat(z_coarse, z_samples))."""
    z, w = _lib.f32(z_coarse), _lib.f32(weights_coarse)
    _lib.require_gpu(z, w)
    N, S = z.shape
    if u is None:
        u = linspace01(N_importance, z.device) if det else torch.rand([N, N_importance], device=z.device)
    u, stride = _check_u(u, N, int(N_importance), "importance_resample")
    if w.shape != z.shape:
        raise ValueError("importance_resample: weights must be [N, S] like z_coarse")
    z_fine = torch.empty(N, S + N_importance, dtype=torch.float32, device=z.device)
    zs = torch.empty(N, N_importance, dtype=torch.float32, device=z.device) if return_samples else None
    _lib.check(_lib.load().dmnerf_importance_resample(_lib.ptr(z), _lib.ptr(w), _lib.ptr(u), stride, N, S, int(N_importance),
                                                      _lib.ptr(z_fine), _lib.ptr(zs), _lib.stream()), "dmnerf_importance_resample")
    return (z_fine, zs) if return_samples else z_fine
