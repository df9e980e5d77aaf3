"""GPU: the optimisation step with / without the fused loss tail (dm_nerf_amd/losses.py, DMNERF_FUSED_TAIL) and with the two levels'
network backwards on one stream vs two (autograd.overlapped_backward, DMNERF_OVERLAP_BWD; default: where it removes a partial
round) at the per-rank shards of an 8-way split (384 / 512 rays) and at the shipped batches (3072 / 4096), eager and as a HIP graph.
    python scripts/time_overlap.py [steps]          -> one JSON line"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda", 0)
    from dm_nerf_amd import distributed as D
    from dm_nerf_amd.networks import helpers as H
    from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical
    pe, ve, mc, mf = B.build_models(dev)
    K = dmsr_intrinsics(B.H_IMG, B.W_IMG)
    ro, rd = H.get_rays_k(B.H_IMG, B.W_IMG, K, pose_spherical(30.0, -65.0, 7.0).to(dev), row0=0, nrows=8)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    z = H.z_val_sample(4096, B.NEAR, B.FAR, B.S_COARSE, device=dev)
    out = {}
    variants = {"separate_losses_one_stream": {"DMNERF_FUSED_TAIL": "0", "DMNERF_OVERLAP_BWD": "0"},
                "fused_tail_one_stream": {"DMNERF_FUSED_TAIL": "1", "DMNERF_OVERLAP_BWD": "0"},
                "fused_tail_two_streams": {"DMNERF_FUSED_TAIL": "1", "DMNERF_OVERLAP_BWD": "1"},
                "default": {}}
    for n in (384, 512, 1024, 3072, 4096):
        for rep in range(2):
            for name, env in variants.items():
                for k in ("DMNERF_FUSED_TAIL", "DMNERF_OVERLAP_BWD"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                r = B.train_leg(mc, mf, ro, rd, z, steps, dev, n=n)
                out.setdefault(f"n{n}", {}).setdefault(name, []).append(round(r["ms_per_step"], 4))
        for name in ("separate_losses_one_stream", "default"):
            for k in ("DMNERF_FUSED_TAIL", "DMNERF_OVERLAP_BWD"):
                os.environ.pop(k, None)
            os.environ.update(variants[name])
            g = B.graph_train_leg(mc, mf, ro, rd, z, steps, dev, n)
            out[f"n{n}"]["graph_" + name] = round(g["ms_per_step"], 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
