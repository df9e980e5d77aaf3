"""Diagnostic (host cores only, no GPU): how far is each f32-class evaluation of the manipulation frame (oracle/manip_margins.py)
from the reference's recorded run ON THIS HOST -- per variant and per output the number of pixels beyond 1e-5 / 1e-4 / 1e-3 / 1e-2."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import manip_margins as MM
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "manipulator_frame.npz"))
g = {k: torch.from_numpy(g[k]) if g[k].ndim else g[k].item() for k in g.files}
print("threads", torch.get_num_threads(), torch.__config__.show().split("\n")[2:6])
ref = [g[k] for k in MM.OUTPUTS]
for nt in (None, 8, 1):
    if nt:
        torch.set_num_threads(nt)
    for name, net in MM.variant_nets().items():
        t0 = time.time()
        o = MM.run_variant(g, net)
        dt = time.time() - t0
        e = torch.stack([(o[k] - ref[k]).abs().amax(-1) for k in range(4)], 1)
        print(f"threads {torch.get_num_threads():3d} {name:20s} {dt:5.1f} s  max " + " ".join(f"{float(e[:, k].max()):.1e}" for k in range(4))
              + "  n>1e-5 " + str([int((e[:, k] > 1e-5).sum()) for k in range(4)]) + " n>1e-4 " + str([int((e[:, k] > 1e-4).sum()) for k in range(4)])
              + " n>1e-3 " + str([int((e[:, k] > 1e-3).sum()) for k in range(4)]), flush=True)
