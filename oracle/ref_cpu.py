"""CPU oracle for the DM-NeRF ray-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, float32) restatement of the reference algorithm
for the path SURVEY.md section 8 scopes.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``dm_nerf_amd/`` imports it, and the product path raises when the HIP
library is missing instead of falling back to this code.

Parity pinning: every function below is checked bit-for-bit against the *imported*
reference (``/root/reference``) in the build container by
``tests/golden/make_golden.py`` (which also writes the committed fixtures under
``tests/golden/*.npz``); ``tests/test_oracle_golden.py`` re-checks the oracle against
those fixtures everywhere (CPU, no reference needed).

Each function cites the reference lines it follows (paths relative to
/root/reference).  Unlike the reference, every tensor is created on the device of
its inputs -- no ``torch.set_default_tensor_type`` side effect (config.py:149-153).
"""
import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# positional encoding  (networks/dm_nerf.py:8-55)
# --------------------------------------------------------------------------------------

def freq_bands(multires):
    """``2.**linspace(0, L-1, L)`` -- exactly 1,2,4,...  (networks/dm_nerf.py:25)."""
    return 2. ** torch.linspace(0., multires - 1, steps=multires)


def embed(x, multires):
    """``Embedder.embed`` (networks/dm_nerf.py:37-38) for include_input=True, log sampling.

    Layout: ``[x | sin(x f0) | cos(x f0) | sin(x f1) | ...]`` in blocks of ``x.shape[-1]``.
    """
    outs = [x]
    for freq in freq_bands(multires):
        f = freq.to(x.device)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


def embed_out_dim(multires, d=3):
    return d + 2 * multires * d


# --------------------------------------------------------------------------------------
# the DM-NeRF MLP  (networks/dm_nerf.py:58-106)
# --------------------------------------------------------------------------------------

PARAM_ORDER = (
    [f"mlps.{i}" for i in range(8)]
    + ["rgb_feature_linear", "ins_feature_linear", "rgb_feature_linears.0",
       "ins_feature_linears.0", "density_linear", "ins_linear", "rgb_linear"]
)


def param_shapes(ins_num, W=256, input_ch_pts=63, input_ch_views=27, D=8, skips=(4,)):
    """Layer shapes of ``DM_NeRF.__init__`` (networks/dm_nerf.py:59-78), state_dict order."""
    shapes = {}
    shapes["mlps.0"] = (W, input_ch_pts)
    for i in range(D - 1):
        shapes[f"mlps.{i + 1}"] = (W, W + input_ch_pts) if i in skips else (W, W)
    shapes["rgb_feature_linear"] = (W, W)
    shapes["ins_feature_linear"] = (W, W)
    shapes["rgb_feature_linears.0"] = (W // 2, W + input_ch_views)
    shapes["ins_feature_linears.0"] = (W // 2, W)
    shapes["density_linear"] = (1, W)
    shapes["ins_linear"] = (ins_num + 1, W // 2)
    shapes["rgb_linear"] = (3, W // 2)
    return shapes


def make_weights(seed, ins_num, W=256, sigma_bias=0.0, gain=1.0, sigma_gain=1.0, head_gain=1.0, D=8, input_ch_pts=63, input_ch_views=27):
    """Deterministic synthetic weights from a numpy seed (no 2.8 MB blobs in the repo).

    ``nn.Linear``-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)) scaled by ``gain``;
    ``sigma_bias`` shifts the density head so rays see surfaces ("trained-like").
    ``sigma_gain`` / ``head_gain`` (applied after the draw, so seeds keep their meaning) scale the density
    head and the rgb / object-code output layers: ``PEAKY`` below gives opaque surfaces in empty space
    (sigma in about [-140, 36], half of the coarse pdf bins below 1e-5), the regime of a trained scene.
    Returns ``{name.weight / name.bias: float32 torch tensor}`` keyed as the
    reference state_dict.
    """
    rng = np.random.RandomState(seed)
    sd = {}
    for name, (o, i) in param_shapes(ins_num, W, input_ch_pts, input_ch_views, D).items():
        bound = gain / np.sqrt(i)
        sd[name + ".weight"] = torch.from_numpy(rng.uniform(-bound, bound, size=(o, i)).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy(rng.uniform(-bound, bound, size=(o,)).astype(np.float32))
    if sigma_gain != 1.0:
        sd["density_linear.weight"] = sd["density_linear.weight"] * np.float32(sigma_gain)
        sd["density_linear.bias"] = sd["density_linear.bias"] * np.float32(sigma_gain)
    if head_gain != 1.0:
        sd["ins_linear.weight"] = sd["ins_linear.weight"] * np.float32(head_gain)
        sd["rgb_linear.weight"] = sd["rgb_linear.weight"] * np.float32(head_gain)
    sd["density_linear.bias"] = sd["density_linear.bias"] + np.float32(sigma_bias)
    return sd


# "trained-like" synthetic weights: make_weights(seed, ins_num, **PEAKY)
PEAKY = dict(gain=2.0, sigma_gain=100.0, sigma_bias=-10.0, head_gain=4.0)


def mlp_forward(sd, x, input_ch_pts=63, input_ch_views=27, skips=(4,), D=8, return_acts=False):
    """``DM_NeRF.forward`` (networks/dm_nerf.py:80-106).

    ``sd`` is a state_dict-like mapping; ``x`` is ``[M, 63+27]``.  Output ``[M, 4+C]`` =
    ``cat[rgb(3), density(1), ins(C)]`` -- raw, no output activations (:105).
    """
    input_pts, input_dirs = torch.split(x, [input_ch_pts, input_ch_views], dim=-1)
    h = input_pts
    acts = []
    for i in range(D):
        h = F.linear(h, sd[f"mlps.{i}.weight"], sd[f"mlps.{i}.bias"])
        h = F.relu(h)
        if i in skips:
            h = torch.cat([h, input_pts], -1)          # order [h, pts]  (:87)
        acts.append(h)
    rgb_feature = F.linear(h, sd["rgb_feature_linear.weight"], sd["rgb_feature_linear.bias"])   # no act (:89)
    rgb_feature = torch.cat([rgb_feature, input_dirs], -1)
    rgb_feature = F.relu(F.linear(rgb_feature, sd["rgb_feature_linears.0.weight"], sd["rgb_feature_linears.0.bias"]))
    ins_feature = h.detach()                                                                    # (:95)
    ins_feature = F.linear(ins_feature, sd["ins_feature_linear.weight"], sd["ins_feature_linear.bias"])
    ins_feature = F.relu(F.linear(ins_feature, sd["ins_feature_linears.0.weight"], sd["ins_feature_linears.0.bias"]))
    density = F.linear(h, sd["density_linear.weight"], sd["density_linear.bias"])
    rgb = F.linear(rgb_feature, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
    ins = F.linear(ins_feature, sd["ins_linear.weight"], sd["ins_linear.bias"])
    out = torch.cat([rgb, density, ins], -1)
    if return_acts:
        return out, acts
    return out


# --------------------------------------------------------------------------------------
# samplers / ray generation  (networks/helpers.py)
# --------------------------------------------------------------------------------------

def get_rays_k(H, W, K, c2w):
    """``get_rays_k`` (networks/helpers.py:50-61).  K: numpy 3x3 / 4x4; c2w: f32 tensor [3or4,4]."""
    dev = c2w.device
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=dev),
                          torch.linspace(0, H - 1, H, device=dev), indexing='ij')
    i = i.t()
    j = j.t()
    dirs = torch.stack([(i - K[0, 2]) / K[0, 0], (j - K[1, 2]) / K[1, 1], K[2, 2] * torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def z_val_sample(N_rays, near, far, N_samples, device="cpu"):
    """``z_val_sample`` (networks/helpers.py:114-119): ``near + linspace(0,1,S)*(far-near)``."""
    near_t = near * torch.ones(size=(N_rays, 1), device=device)
    far_t = far * torch.ones(size=(N_rays, 1), device=device)
    t_vals = torch.linspace(0., 1., steps=N_samples, device=device)
    z = near_t + t_vals * (far_t - near_t)
    return z.expand([N_rays, N_samples])


def stratify(z_vals, t_rand):
    """Stratified jitter (networks/render.py:42-47) with the random draw passed in."""
    mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    return lower + (upper - lower) * t_rand


def sample_pdf(bins, weights, N_samples, det=False, u=None, return_aux=False):
    """``sample_pdf`` (networks/helpers.py:123-155).  ``u`` overrides the draw (same shape/order)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0., 1., steps=N_samples, device=bins.device)
            u = u.expand(list(cdf.shape[:-1]) + [N_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [N_samples], device=bins.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    matched_shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(matched_shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(matched_shape), 2, inds_g)
    denom = (cdf_g[..., 1] - cdf_g[..., 0])
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    samples = bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])
    if return_aux:
        return samples, cdf, inds
    return samples


def sample_from_cdf(bins, cdf, u):
    """Stage-isolated tail of ``sample_pdf`` (helpers.py:139-153): identical (cdf,u) -> inds, samples."""
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo = torch.gather(cdf, 1, below)
    cdf_hi = torch.gather(cdf, 1, above)
    b_lo = torch.gather(bins, 1, below)
    b_hi = torch.gather(bins, 1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return b_lo + t * (b_hi - b_lo), inds


# --------------------------------------------------------------------------------------
# volume rendering  (networks/render.py)
# --------------------------------------------------------------------------------------

def render_train(raw, z_vals, rays_d):
    """``render_train`` (networks/render.py:6-28)."""
    dev = raw.device
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.tensor([1e10], device=dev).expand(dists[..., :1].shape)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    ins_labels = raw[..., 4:]
    alpha = 1. - torch.exp(-F.relu(raw[..., 3]) * dists)
    weights = alpha * torch.cumprod(
        torch.cat([torch.ones((alpha.shape[0], 1), device=dev), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    weights_ins = weights.clone().detach()
    ins_map = torch.sum(weights_ins[..., None] * ins_labels, -2)
    ins_map = torch.sigmoid(ins_map)
    ins_map = ins_map[..., :-1]
    return rgb_map, weights, depth_map, ins_map


def dm_nerf(rays, sd_coarse, sd_fine, z_vals_coarse, perturb=0., N_importance=128,
            is_train=False, N_ins=None, t_rand=None, u=None, multires=10, multires_views=4,
            z_fine_override=None):
    """``dm_nerf`` (networks/render.py:31-96) with weights as state_dicts.

    RNG: when ``perturb > 0`` the reference draws ``torch.rand([N,64])`` (:46) then
    ``torch.rand([N,N_importance])`` (helpers.py:135); pass ``t_rand`` / ``u`` to pin them,
    otherwise they are drawn here in that order.  ``z_fine_override`` (tests only) replaces the
    resampled depths, to compare fine-level quantities without the ill-conditioned inverse-CDF step.
    """
    rays_o, rays_d = rays
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
    if perturb > 0.:
        if t_rand is None:
            t_rand = torch.rand(z_vals_coarse.shape, device=z_vals_coarse.device)
        z_vals_coarse = stratify(z_vals_coarse, t_rand)

    def run(sd, z):
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
        pts_flat = torch.reshape(pts, [-1, 3])
        e_pos = embed(pts_flat, multires)
        dirs = torch.reshape(viewdirs[:, None].expand(pts.shape), [-1, 3])
        e_dir = embed(dirs, multires_views)
        depth = sum(1 for k in sd if k.startswith("mlps.") and k.endswith(".weight"))       # netdepth (config.py:31)
        raw = mlp_forward(sd, torch.cat([e_pos, e_dir], -1), input_ch_pts=e_pos.shape[-1], input_ch_views=e_dir.shape[-1], D=depth)
        return torch.reshape(raw, list(pts.shape[:-1]) + [raw.shape[-1]])

    raw_coarse = run(sd_coarse, z_vals_coarse)
    rgb_coarse, weights_coarse, depth_coarse, ins_coarse = render_train(raw_coarse, z_vals_coarse, rays_d)
    z_vals_mid = .5 * (z_vals_coarse[..., 1:] + z_vals_coarse[..., :-1])
    z_samples = sample_pdf(z_vals_mid, weights_coarse[..., 1:-1], N_importance, det=(perturb == 0.), u=u)
    z_samples = z_samples.detach()
    z_vals_fine, _ = torch.sort(torch.cat([z_vals_coarse, z_samples], -1), -1)
    if z_fine_override is not None:
        z_vals_fine = z_fine_override
    raw_fine = run(sd_fine, z_vals_fine)
    rgb_fine, weights_fine, depth_fine, ins_fine = render_train(raw_fine, z_vals_fine, rays_d)
    if is_train and N_ins is not None:
        ins_fine = ins_fine[-N_ins:]
        ins_coarse = ins_coarse[-N_ins:]
    return {'rgb_fine': rgb_fine, 'ins_fine': ins_fine, 'z_vals_fine': z_vals_fine, 'raw_fine': raw_fine,
            'raw_coarse': raw_coarse, 'rgb_coarse': rgb_coarse, 'ins_coarse': ins_coarse,
            'z_vals_coarse': z_vals_coarse, 'depth_fine': depth_fine, 'depth_coarse': depth_coarse}


# --------------------------------------------------------------------------------------
# emptiness penalizer  (networks/penalizer.py) -- SURVEY 8(f)-1, the consumer of raw_* / z_vals_* / depth_*
# --------------------------------------------------------------------------------------

def emptiness_penalizer(raw, z_vals, depths, rays_d, tolerance, deta_w):
    """``emptiness_penalizer`` (networks/penalizer.py:5-55), device-explicit."""
    dev = raw.device
    delta_H, delta_W = torch.tensor([0.4], device=dev), torch.tensor([deta_w], device=dev)
    gauss = lambda dd, dh, dw: torch.exp(-(dd ** 2) / (2 * (dw ** 2))) / (dh * torch.sqrt(torch.tensor([2 * np.pi], device=dev))) + 1e-8
    norm = torch.norm(rays_d[..., None, :], dim=-1)
    dists_before = (depths - tolerance) * norm
    dists_after = (depths + tolerance) * norm
    depth_dist = depths * norm
    p_dists = z_vals * norm
    delta_dist = depth_dist - p_dists
    penalize_weights = gauss(delta_dist, delta_H, delta_W)
    penalize_weights_air = 1 - penalize_weights
    mask_before = (p_dists < dists_before).type(torch.float32)
    mask_after = (p_dists > dists_after).type(torch.float32)
    mask_middle = 1 - (mask_after + mask_before)
    pred_ins = torch.sigmoid(raw[..., 4:])
    gt = torch.zeros_like(pred_ins)
    gt[..., -1] = 1
    loss_before = -gt * torch.log(pred_ins + 1e-8) - (1 - gt) * torch.log(1 - pred_ins + 1e-8)
    loss_before = loss_before * (penalize_weights_air * mask_before)[..., None]
    loss_before = torch.sum(loss_before) / (pred_ins.shape[-1] * torch.maximum(torch.sum(mask_before), torch.tensor([1e-8], device=dev)))
    pm = pred_ins[..., -1]
    gtm = torch.zeros_like(pm)
    loss_middle = -gtm * torch.log(pm + 1e-8) - (1 - gtm) * torch.log(1 - pm + 1e-8)
    loss_middle = loss_middle * (penalize_weights * mask_middle)
    loss_middle = torch.sum(loss_middle) / torch.maximum(torch.sum(mask_middle), torch.tensor([1e-8], device=dev))
    return loss_before + loss_middle


def ins_penalizer(raw, z_vals, depth, rays_d, tolerance, deta_w):
    """``ins_penalizer`` (networks/penalizer.py:58-62): depth is detached."""
    return emptiness_penalizer(raw, z_vals, depth[..., None].detach(), rays_d, tolerance, deta_w)


# --------------------------------------------------------------------------------------
# object-code loss  (networks/evaluator.py:19-74) -- SURVEY 8(f)-2, the consumer of ins_coarse / ins_fine
# --------------------------------------------------------------------------------------

def ins_cost_matrices(pred_ins, gt_labels, ins_num):
    """Cost matrices of ``hungarian`` (networks/evaluator.py:41-70): rows = the labels present in ``gt_labels``
    in ascending order (the one-hot compaction of :21-26, rows beyond ``valid`` belong to the all-zero columns),
    columns = predicted channels.  Returns ``cost_ce, cost_siou [ins_num, ins_num], valid``."""
    present = torch.unique(gt_labels)
    valid = len(present)
    onehot = torch.zeros(gt_labels.shape[0], ins_num)
    onehot[..., :valid] = F.one_hot(gt_labels.long())[..., present.long()]
    P = pred_ins.permute([1, 0])[None, :, :]                     # [1, channel, ray]
    G = onehot.permute([1, 0])[:, None, :]                       # [row, 1, ray]
    cost_ce = torch.mean(-G * torch.log(P + 1e-8) - (1 - G) * torch.log(1 - P + 1e-8), dim=-1)
    TP = torch.sum(P * G, dim=-1)
    FP = torch.sum(P, dim=-1) - TP
    FN = torch.sum(G, dim=-1) - TP
    cost_siou = 1.0 - TP / (TP + FP + FN + 1e-6)
    return cost_ce, cost_siou, valid


def ins_assignment(cost_ce, cost_siou, valid, ins_num):
    """``reorder`` (evaluator.py:43-54): scipy's rectangular assignment on the ``valid`` label rows of
    ``cost_ce + cost_siou``; the unmatched channels follow in ``set`` order.  -> ``rows, cols`` (numpy)."""
    from scipy.optimize import linear_sum_assignment
    with torch.no_grad():
        rows, cols = linear_sum_assignment((cost_ce + cost_siou)[:valid].cpu().numpy())
    if ins_num - valid > 0:
        cols = np.concatenate([cols, np.array(list(set(range(ins_num)) - set(cols)))])
    return rows, cols


def ins_criterion(pred_ins, gt_labels, ins_num):
    """``ins_criterion`` (networks/evaluator.py:19-37) -> ``(loss, valid_ce, invalid_ce, valid_siou)``:
    mean matched cross-entropy + mean matched soft-IoU cost + mean prediction of the unmatched channels."""
    cost_ce, cost_siou, valid = ins_cost_matrices(pred_ins, gt_labels, ins_num)
    rows, cols = ins_assignment(cost_ce, cost_siou, valid, ins_num)
    valid_ce = torch.mean(cost_ce[rows, cols[:valid]])
    invalid_ce = torch.mean(pred_ins[:, cols[valid:]]) if len(cols) != valid else torch.tensor([0])
    valid_siou = torch.mean(cost_siou[rows, cols[:valid]])
    return valid_ce + invalid_ce + valid_siou, valid_ce, invalid_ce, valid_siou


# --------------------------------------------------------------------------------------
# manipulation render  (networks/manipulator.py:18-205) -- SURVEY 8(f)-3
# --------------------------------------------------------------------------------------

def exchanger(ori_raw, tar_raws, ori_raw_pred, tar_raw_preds, move_labels):
    """``exchanger`` (networks/manipulator.py:18-83).  Mutates ``ori_raw`` in place, like the reference."""
    ori_pred_label = torch.argmax(torch.sigmoid(ori_raw[..., 4:]), dim=-1)
    ori_accum_label = torch.argmax(torch.sigmoid(ori_raw_pred[..., :-1]), dim=-1)
    ori_accum_label = ori_accum_label[:, None].repeat(1, ori_pred_label.shape[-1])
    tar_pred_label_temp = None
    for idx, move_label in enumerate(move_labels):
        tar_raw = tar_raws[idx]
        tar_raw_pre = tar_raw_preds[idx]
        ori_occludes = (ori_accum_label != move_label) * (ori_pred_label == move_label)
        ori_pred_label[ori_occludes == True] = ori_accum_label[ori_occludes == True]
        fillings = (ori_accum_label == move_label) * (ori_pred_label != move_label)
        tar_pred_label = torch.argmax(torch.sigmoid(tar_raw[..., 4:]), dim=-1)
        tar_pred_label_temp = tar_pred_label
        tar_accum_label = torch.argmax(torch.sigmoid(tar_raw_pre[..., :-1]), dim=-1)
        tar_accum_label = tar_accum_label[:, None].repeat(1, tar_pred_label.shape[-1])
        tar_occludes = (tar_accum_label != move_label) * (tar_pred_label == move_label)
        tar_pred_label[tar_occludes == True] = tar_accum_label[tar_occludes == True]
        operation_mask = torch.zeros_like(ori_pred_label)
        ori_move_mask, tar_move_mask = torch.zeros_like(ori_pred_label), torch.zeros_like(tar_pred_label)
        ori_move_mask[ori_pred_label == move_label] = -2
        tar_move_mask[tar_pred_label == move_label] = 1
        reduced_mask = tar_move_mask - ori_move_mask
        operation_mask[reduced_mask == 0] = -1
        operation_mask[reduced_mask == 1] = 1
        operation_mask[reduced_mask == 2] = 0
        operation_mask[reduced_mask == 3] = 1
        ori_raw[fillings] = tar_raw[fillings]
        ori_raw[operation_mask == 1] = tar_raw[operation_mask == 1]
        ori_raw[operation_mask == 0] = ori_raw[operation_mask == 0] * 0
    return ori_raw, tar_raws, ori_pred_label, tar_pred_label_temp


def manipulator_render(raw, z_vals, rays_d):
    """``manipulator_render`` (networks/manipulator.py:86-105): like render_train, but the object map keeps all
    C channels and is not detached."""
    dev = raw.device
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.tensor([1e10], device=dev).expand(dists[..., :1].shape)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    alpha = 1. - torch.exp(-F.relu(raw[..., 3]) * dists)
    weights = alpha * torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1), device=dev), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    ins_map = torch.sigmoid(torch.sum(weights[..., None] * raw[..., 4:], -2))
    depth_map = torch.sum(weights * z_vals, -1)
    return rgb_map, weights, depth_map, ins_map


def manipulator_z(N_rays, near, far, N_samples, device="cpu"):
    """z grid of ``manipulator_nerf`` (:114-119): ``near (1 - t) + far t`` (rounds differently from z_val_sample)."""
    near_, far_ = near * torch.ones(size=(N_rays, 1), device=device), far * torch.ones(size=(N_rays, 1), device=device)
    t_vals = torch.linspace(0., 1., steps=N_samples, device=device)
    return (near_ * (1. - t_vals) + far_ * t_vals).expand([N_rays, N_samples])


def manipulator_nerf(rays, sd, N_samples=None, near=None, far=None, z_vals=None):
    """``manipulator_nerf`` (networks/manipulator.py:108-134) with the model as a state_dict."""
    rays_o, rays_d = rays
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
    if z_vals is None:
        z_vals = manipulator_z(rays_d.shape[0], near, far, N_samples, rays_d.device)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    e = torch.cat([embed(pts.reshape(-1, 3), 10), embed(viewdirs[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
    raw = mlp_forward(sd, e)
    return torch.reshape(raw, list(pts.shape[:-1]) + [raw.shape[-1]]), z_vals


def manipulator(sd_coarse, sd_fine, ori_rays, f_tar_rays, N_samples, N_importance, near, far, target_labels, us=None, net=None, probe=None):
    """``manipulator`` (networks/manipulator.py:137-205).  ``us``: optional list of the ``2 + T`` uniform draws
    [N, N_importance] in the order the reference makes them (original, each target, original again).
    Test hooks (both default to the plain restatement): ``net`` replaces ``manipulator_nerf`` (same signature: another evaluation of
    the same network, e.g. in float64); ``probe(event, **tensors)`` is called with the inputs of every DISCRETE decision of the chain
    -- the three-plus-T resamplings (``"resample"``: bins, weights, u) and the two exchanger rounds (``"exchange"``: the raws and
    accumulated object maps) -- so that a test can tell which rays sit next to a threshold (oracle/manip_margins.py)."""
    us = list(us) if us is not None else None
    draw = lambda: us.pop(0) if us is not None else None
    net = net or manipulator_nerf
    probe = probe or (lambda event, **kw: None)

    def resample(tag, mid, w, n, u):
        probe("resample", tag=tag, bins=mid, weights=w, u=u)
        return sample_pdf(mid, w, n, u=u)
    ori_raw, ori_z = net(ori_rays, sd_coarse, N_samples, near, far)
    _, ori_w, _, _ = manipulator_render(ori_raw, ori_z, ori_rays[1])
    ori_mid = .5 * (ori_z[..., 1:] + ori_z[..., :-1])
    ori_zs = resample("ori", ori_mid, ori_w[..., 1:-1], N_importance, draw())
    ori_z_full, _ = torch.sort(torch.cat([ori_z, ori_zs], dim=-1), dim=-1)
    ori_raw_full, _ = net(ori_rays, sd_fine, N_samples, near, far, z_vals=ori_z_full)
    _, _, _, ori_ins_accum = manipulator_render(ori_raw_full, ori_z_full, ori_rays[1])
    tar_raws, f_tar_z, f_tar_zs, tar_ins_accums = [], [], [], []
    tar_rgb = tar_ins_accum = None
    for k, tar_rays in enumerate(f_tar_rays):
        tar_raw, tar_z = net(tar_rays, sd_coarse, N_samples, near, far)
        tar_raws.append(tar_raw); f_tar_z.append(tar_z)
        tar_rgb, tar_w, _, _ = manipulator_render(tar_raw, tar_z, tar_rays[1])
        tar_mid = .5 * (tar_z[..., 1:] + tar_z[..., :-1])
        tar_zs = resample(f"tar{k}", tar_mid, tar_w[..., 1:-1], N_importance, draw())
        tar_z_full, _ = torch.sort(torch.cat([tar_z, tar_zs], dim=-1), dim=-1)
        tar_raw_full, _ = net(tar_rays, sd_fine, z_vals=tar_z_full)
        _, _, _, tar_ins_accum = manipulator_render(tar_raw_full, tar_z_full, tar_rays[1])
        f_tar_zs.append(tar_zs); tar_ins_accums.append(tar_ins_accum)
    probe("exchange", tag="coarse", ori_raw=ori_raw, tar_raws=tar_raws, ori_acc=ori_ins_accum, tar_accs=tar_ins_accums, labels=target_labels)
    ori_raw, _, _, _ = exchanger(ori_raw, tar_raws, ori_ins_accum, tar_ins_accums, target_labels)
    _, ori_w, _, _ = manipulator_render(ori_raw, ori_z, ori_rays[1])
    ori_zs = resample("edited", .5 * (ori_z[..., 1:] + ori_z[..., :-1]), ori_w[..., 1:-1], N_importance, draw())
    f_tar_zs = torch.cat(f_tar_zs, dim=-1)
    ori_z, _ = torch.sort(torch.cat([ori_z, ori_zs, f_tar_zs], dim=-1), dim=-1)
    for idx, tar_rays in enumerate(f_tar_rays):
        ori_raw, ori_z = net(ori_rays, sd_fine, z_vals=ori_z)
        tar_z, _ = torch.sort(torch.cat([f_tar_z[idx], ori_zs, f_tar_zs], dim=-1), dim=-1)
        tar_raws[idx], _ = net(tar_rays, sd_fine, z_vals=tar_z)
    probe("exchange", tag="fine", ori_raw=ori_raw, tar_raws=tar_raws, ori_acc=ori_ins_accum, tar_accs=tar_ins_accums, labels=target_labels)
    ori_raw, _, _, _ = exchanger(ori_raw, tar_raws, ori_ins_accum, tar_ins_accums, target_labels)
    final_rgb, _, _, final_ins = manipulator_render(ori_raw, ori_z, ori_rays[1])
    return final_rgb, final_ins, tar_rgb, tar_ins_accum


def manipulate_frame(sd_coarse, sd_fine, H, W, K, ori_pose, trans, N_test, N_samples, N_importance, near, far, target_labels, us=None,
                     net=None, probe=None):
    """The per-pose body of ``manipulator_eval`` (networks/manipulator.py:230-274): original rays of ``ori_pose``, target rays of
    ``trans @ ori_pose`` (:235), the chunk loop of ``N_test`` rays with its ragged last chunk (:241-244) around ``manipulator``
    (one transformation: ``tar_batch_rays[None]``, :255), the four accumulations (:260-266) and the ``[H, W, .]`` reshape
    (:271-272).  ``us``: per chunk the ``2 + 1`` draws ``manipulator`` makes.  Returns
    ``(full_rgb [H,W,3], full_ins [H,W,C], full_tar_rgb [H,W,3], full_tar_ins [H,W,C], tar_pose)``."""
    ori_pose, trans = torch.as_tensor(ori_pose, dtype=torch.float32), torch.as_tensor(trans, dtype=torch.float32)
    ro, rd = get_rays_k(H, W, K, ori_pose)
    ro, rd = torch.reshape(ro, [-1, 3]).float(), torch.reshape(rd, [-1, 3]).float()
    tar_pose = trans @ ori_pose
    to, td = get_rays_k(H, W, K, tar_pose)
    to, td = torch.reshape(to, [-1, 3]).float(), torch.reshape(td, [-1, 3]).float()
    cols = [[], [], [], []]
    for c, step in enumerate(range(0, H * W, N_test)):
        n = min(N_test, H * W - step)
        ori = torch.stack([ro[step:step + n], rd[step:step + n]], dim=0)
        tar = torch.stack([to[step:step + n], td[step:step + n]], dim=0)[None, ...]
        out = manipulator(sd_coarse, sd_fine, ori, tar, N_samples, N_importance, near, far, target_labels,
                          us=None if us is None else us[c], net=net, probe=probe)
        for col, t in zip(cols, out):
            col.append(t)
    full = [torch.cat(col, dim=0) for col in cols]
    return tuple(t.reshape([H, W, t.shape[-1]]) for t in full) + (tar_pose,)


# --------------------------------------------------------------------------------------
# synthetic scene; used by tests and bench
# --------------------------------------------------------------------------------------

def dmsr_intrinsics(H=480, W=640, camera_angle_x=0.69):
    """DM-SR K convention ``[[f,0,W/2],[0,-f,H/2],[0,0,-1]]`` (datasets/loader_dmsr.py:136-137)."""
    focal = .5 * W / np.tan(.5 * camera_angle_x)
    return np.array([[focal, 0, 0.5 * W], [0, -focal, 0.5 * H], [0, 0, -1]])


def pose_spherical(theta, phi, radius):
    """Restated ``pose_spherical`` (tools/pose_generator.py:29-34) -> 4x4 float32 c2w."""
    t = np.eye(4); t[2, 3] = radius
    ph = phi / 180. * np.pi
    rx = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]])
    th = theta / 180. * np.pi
    ry = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
    c2w = ry @ rx @ t
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ c2w
    return torch.from_numpy(c2w.astype(np.float32))


def scannet_scene(H=480, W=640, ins_num=7):
    """Synthetic ScanNet-form frame: a uniform-random image, a blocky label map with ``ins_num`` = unlabelled (what
    ``ins_processor.load_semantic_instance`` produces, loader_scannet.py:128-137), the crop mask of ``crop_data``
    (:24-29) for the shipped crop 640 x 480, the per-image labelled-pixel list of ``selected_pixels`` (:139-151)."""
    gen = torch.Generator().manual_seed(1010)
    rgb = torch.rand(H, W, 3, generator=gen)
    rr, cc = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    lab = (((rr // 60) * 3 + (cc // 80)) % (ins_num + 1)).astype(np.int8)            # 8 values; 7 = unlabelled
    crop = np.zeros((H, W)); crop[0:H, 0:W] = 1; crop = crop.astype(np.int8)         # crop_width 640, crop_height 480
    ins = lab.reshape(-1).copy(); ins[crop.reshape(-1) == 0] = ins_num
    ins_index = np.where(ins != ins_num)[0]
    return rgb, lab, crop, ins_index


def fuse_heads(state):
    """What the library's ``dmnerf_fuse_heads`` (csrc/heads.hip; inference-only ``args.fuse_heads`` and the split modes) must
    compute, stated with torch ops on a state_dict -- the CHECKER of that kernel, not a path of the product (which calls
    ``weights.fused_flat``): the activation-free ``rgb_feature_linear`` / ``ins_feature_linear`` (dm_nerf.py:89,96) folded into
    the hidden layers that consume them.  Returns a copy of ``state`` whose ``rgb_feature_linears.0`` /
    ``ins_feature_linears.0`` hold ``W_hidden[:, :256] @ W_feature`` (float64, rounded once) and the matching biases."""
    st = {k: v.detach() for k, v in state.items()}
    for feat, hid in (("rgb_feature_linear", "rgb_feature_linears.0"), ("ins_feature_linear", "ins_feature_linears.0")):
        Wf, bf = st[feat + ".weight"].double(), st[feat + ".bias"].double()
        Wh, bh = st[hid + ".weight"].double(), st[hid + ".bias"].double()
        n_in = Wf.shape[0]                                         # 256 feature columns (then the 27 dir columns, rgb only)
        Wn = Wh.clone()
        Wn[:, :n_in] = Wh[:, :n_in] @ Wf
        st[hid + ".weight"] = Wn.float()
        st[hid + ".bias"] = (Wh[:, :n_in] @ bf + bh).float()
    return st
