"""GPU box: a loop of ins_criterion forward + backward for a kernel trace (per-kernel times of cr_partial / cr_solve / cr_bwd):

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o crit -- python $REPO/scripts/time_criterion.py
    grep -E "cr_" $OUT/*/crit_kernel_stats.csv

DMNERF_DIAG_LIB=<lib.so>: an alternative build.  Prints the losses (they must not change with the build)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                    # noqa: E402
from dm_nerf_amd import _lib                                    # noqa: E402
if os.environ.get("DMNERF_DIAG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DMNERF_DIAG_LIB"])
from dm_nerf_amd.networks import evaluator as E                 # noqa: E402

dev = torch.device("cuda:0")
for N, C, nl in ((4096, 13, 9), (3072, 13, 13), (4096, 93, 40)):
    g = torch.Generator(device=dev).manual_seed(0)
    pred = torch.rand(N, C, device=dev, generator=g).requires_grad_(True)
    lab = torch.randint(0, nl, (N,), device=dev, generator=g)
    for _ in range(30):
        out = E.ins_criterion(pred, lab, C)
        out[0].backward()
    torch.cuda.synchronize()
    print(f"N={N} C={C}: " + " ".join(f"{float(x.detach()):.7f}" for x in out) + f"  grad sum {float(pred.grad.double().abs().sum()):.9e}")
