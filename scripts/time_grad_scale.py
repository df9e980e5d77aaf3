import sys, torch
sys.path.insert(0, "/root/repo")
from dm_nerf_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for n in (786432 * 18, 262144 * 18, 1000003):
    g = torch.randn(n, device=dev) * 1e-5
    g[n // 3] = 3.7e-4
    sc = torch.zeros(4, device=dev)
    for _ in range(3):
        _lib.check(lib.dmnerf_grad_scale(_lib.ptr(g), n, _lib.ptr(sc), _lib.stream()), "gs")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        _lib.check(lib.dmnerf_grad_scale(_lib.ptr(g), n, _lib.ptr(sc), _lib.stream()), "gs")
    b.record(); torch.cuda.synchronize()
    mx = float(g.abs().max())
    import math
    want = 2.0 ** (6 - math.frexp(mx)[1])
    print(n, "us per call", a.elapsed_time(b) / 20 * 1e3, "scale", sc.tolist()[:2], "want", want, "GB/s", n * 4 / (a.elapsed_time(b) / 20 * 1e-3) / 1e9)
    assert sc[0].item() == want and sc[2].item() == 0 and sc[3].item() == 0
