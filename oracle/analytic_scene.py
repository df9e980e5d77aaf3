"""Test infrastructure (like oracle/ref_cpu.py: only tests/, scripts/ and bench.py's checkers may import it).

An ANALYTIC scene with a known answer, for convergence tests that do not need a dataset: a few large, well-separated
objects -- four spheres on a ground disc -- seen by DM-SR-style cameras (tools/pose_generator.py:29-34 poses at radius 7,
datasets/loader_dmsr.py:136-137 intrinsics, near 4 / far 15 like configs/dmsr/train/study.txt).  Images and per-pixel object
labels come from exact ray / primitive intersection in float64 (Lambert shading from one fixed light), so the targets do not
depend on any renderer under test; labels are object ids 0 .. 4 and ``EMPTY`` (= ins_num, "no object": the value the
reference's loaders give unlabelled pixels) where a ray leaves the scene."""
import numpy as np
import torch

from .ref_cpu import dmsr_intrinsics, get_rays_k, pose_spherical

NEAR, FAR = 4.0, 15.0
SPHERES = (   # centre, radius, colour
    ((-1.15, -1.05, 0.00), 0.80, (0.85, 0.20, 0.15)),
    ((1.20, -0.95, 0.10), 0.90, (0.15, 0.65, 0.25)),
    ((-1.05, 1.20, -0.05), 0.75, (0.20, 0.30, 0.85)),
    ((1.10, 1.15, 0.15), 0.95, (0.90, 0.75, 0.15)),
)
GROUND_Z, GROUND_R, GROUND_RGB = -0.80, 2.9, (0.55, 0.55, 0.60)       # a disc: label 4
BACKGROUND_RGB = (0.95, 0.95, 0.95)
LIGHT = np.array([0.35, -0.25, 0.90]) / np.linalg.norm([0.35, -0.25, 0.90])
N_OBJECTS = len(SPHERES) + 1


def render_view(H, W, c2w, ins_num):
    """-> rgb [H, W, 3] float32 in [0, 1], labels [H, W] int64 (0 .. N_OBJECTS - 1, or ins_num where nothing is hit)."""
    K = dmsr_intrinsics(H, W)
    ro, rd = get_rays_k(H, W, K, c2w)
    o = ro.reshape(-1, 3).double().numpy()
    d = rd.reshape(-1, 3).double().numpy()
    n = o.shape[0]
    t_hit = np.full(n, np.inf)
    rgb = np.tile(np.array(BACKGROUND_RGB), (n, 1))
    lab = np.full(n, ins_num, dtype=np.int64)
    dd = (d * d).sum(-1)
    for k, (c, r, col) in enumerate(SPHERES):
        oc = o - np.array(c)
        b = (oc * d).sum(-1)
        disc = b * b - dd * ((oc * oc).sum(-1) - r * r)
        t = (-b - np.sqrt(np.maximum(disc, 0.0))) / dd
        hit = (disc > 0) & (t > 0) & (t < t_hit)
        p = o[hit] + d[hit] * t[hit, None]
        nrm = (p - np.array(c)) / r
        shade = 0.35 + 0.65 * np.clip(nrm @ LIGHT, 0.0, 1.0)
        rgb[hit] = np.array(col) * shade[:, None]
        lab[hit] = k
        t_hit[hit] = t[hit]
    t = (GROUND_Z - o[:, 2]) / np.where(np.abs(d[:, 2]) > 1e-12, d[:, 2], 1e-12)
    p = o + d * t[:, None]
    hit = (t > 0) & (t < t_hit) & ((p[:, 0] ** 2 + p[:, 1] ** 2) < GROUND_R ** 2)
    # soft contact shadows would need a second ray; a radial tint is enough to give the disc structure
    tint = 0.80 + 0.20 * np.sqrt(p[hit, 0] ** 2 + p[hit, 1] ** 2) / GROUND_R
    rgb[hit] = np.array(GROUND_RGB) * tint[:, None] * (0.35 + 0.65 * LIGHT[2])
    lab[hit] = len(SPHERES)
    return (torch.from_numpy(rgb.reshape(H, W, 3).astype(np.float32)), torch.from_numpy(lab.reshape(H, W)))


def make_views(H, W, thetas, ins_num, phi=-65.0, radius=7.0):
    """-> poses [V, 4, 4], images [V, H, W, 3], labels [V, H, W]."""
    poses = torch.stack([pose_spherical(float(th), phi, radius) for th in thetas])
    ims, labs = zip(*[render_view(H, W, p, ins_num) for p in poses])
    return poses, torch.stack(ims), torch.stack(labs)


def psnr(a, b):
    return float(-10.0 * torch.log10(((a.double() - b.double()) ** 2).mean()))


def purity(pred, gt):
    """Permutation-invariant label quality (the Hungarian-matched loss, evaluator.py:19-74, leaves the channel <-> object
    assignment free): sum over predicted channels of their largest overlap with one ground-truth label, / pixels."""
    tot = 0
    for c in torch.unique(pred):
        tot += int(torch.bincount(gt[pred == c]).max())
    return tot / pred.numel()
