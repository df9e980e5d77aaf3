"""GPU box: K optimisation steps (the recipe of bench.py `train`) on a batch of N rays, for a rocprofv3 kernel trace:

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o step -- python $REPO/scripts/step_profile.py --rays 384 --steps 20

Prints ms per step; `scripts/step_profile.py --summarize $OUT` lists launches per step and GPU time per kernel name."""
import argparse
import csv
import glob
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarize(d, steps):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not f:
        print("no kernel trace under", d); return
    rows = list(csv.DictReader(open(f[-1])))
    agg = {}
    for r in rows:
        k = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += dur
    tot_n = sum(a[0] for a in agg.values()); tot_t = sum(a[1] for a in agg.values())
    t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
    print(f"{tot_n} launches, {tot_t / 1e3:.2f} ms of kernel time, span {(t1 - t0) / 1e6:.2f} ms  ({tot_n / steps:.1f} launches and {tot_t / steps:.0f} us of kernel time per step incl. warm-up steps)")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"  {n / steps:7.2f} /step  {t / steps:9.1f} us/step  {t / n:8.1f} us each   {k[:110]}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=384)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--mode", default="")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True): the update as multi-tensor kernels")
    ap.add_argument("--summarize", default=None)
    a = ap.parse_args()
    if a.summarize:
        return summarize(a.summarize, a.steps + 3)
    import torch
    from dm_nerf_amd import config as Cfg, distributed as D
    from dm_nerf_amd.networks import helpers as H
    from dm_nerf_amd.synthetic import dmsr_intrinsics, pose_spherical
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cargs = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=8, netwidth=256, ins_num=13, device=dev)
    _, _, mc, mf, _ = Cfg.create_nerf(cargs)
    mc.train(); mf.train()
    n = a.rays
    K = dmsr_intrinsics(480, 640)
    ro, rd = H.get_rays_k(480, 640, K, pose_spherical(30.0, -65.0, 7.0).to(dev), row0=0, nrows=-(-n // 640))
    rays = torch.stack([ro.reshape(-1, 3)[:n], rd.reshape(-1, 3)[:n]])
    z = H.z_val_sample(n, 4.0, 15.0, 64, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    target = torch.rand(n, 3, device=dev, generator=g)
    labels = torch.randint(0, 9, (n,), device=dev, generator=g)
    args = types.SimpleNamespace(perturb=1.0, N_importance=128, is_train=True, N_ins=None, penalize=True, tolerance=0.05, deta_w=0.05,
                                 mfma_split=a.mode or False)
    params = list(mc.parameters()) + list(mf.parameters())
    if a.graph:
        from dm_nerf_amd.graphed import GraphedTrainStep
        opt = torch.optim.Adam(params, lr=torch.tensor(5e-4, device=dev), capturable=True)
        gs = GraphedTrainStep((mc, mf), opt, args, 13, rays, z, target, labels)
        one = gs.step
    else:
        opt = torch.optim.Adam(params, lr=5e-4, fused=True) if a.fused_adam else torch.optim.Adam(params, lr=5e-4)
        one = lambda: D.sharded_train_step(rays, z, target, labels, (mc, mf), args, opt, 13)[0]
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one()
    torch.cuda.synchronize()
    print(f"{n} rays, mode {a.mode or 'f32'}{' (graph)' if a.graph else ''}{' (fused Adam)' if a.fused_adam else ''}: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms per step")


if __name__ == "__main__":
    main()
