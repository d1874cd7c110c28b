"""Latency of the backward kernels against (samples, batch) for one layer shape (HIP events)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
from bnn_hip import ops, _lib as L
dev = torch.device("cuda", 0)
K, N = int(sys.argv[1]), int(sys.argv[2])
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for S, B in ((1, 16), (1, 32), (1, 64), (1, 128), (2, 128), (4, 128), (8, 128)):
    x = torch.randn(S, B, K, device=dev)
    gy = torch.randn(S, B, N, device=dev)
    y = torch.rand(S, B, N, device=dev)
    v = torch.rand(S, B, N, device=dev) + 0.1
    wl = torch.randn(K, N, device=dev) * 0.1; rl = torch.full((K, N), -4.0, device=dev)
    wb = torch.randn(N, K, device=dev) * 0.1; rb = torch.full((N, K), -4.0, device=dev)
    bm = torch.zeros(N, device=dev); br = torch.full((N,), -4.0, device=dev)
    t_lr = timeit(lambda: ops.lr_linear_bwd(x, gy, y, v, wl, rl, bm, br, n_samples=S, sigma_p=1.0, relu=True,
                                            eps_mode=L.EPS_PHILOX, seed=1, layer_id=1, want_gx=False))
    t_lrx = timeit(lambda: ops.lr_linear_bwd(x, gy, y, v, wl, rl, bm, br, n_samples=S, sigma_p=1.0, relu=True,
                                             eps_mode=L.EPS_PHILOX, seed=1, layer_id=1, want_gx=True))
    t_bbb = timeit(lambda: ops.bbb_linear_bwd(x, gy, y, wb, rb, bm, br, n_samples=S, prior=ops.PriorSpec(False, 1.0),
                                              math_mode=L.MATH_F32, relu=True, eps_mode=L.EPS_PHILOX, seed=1, layer_id=1,
                                              want_gx=False))
    t_bbbx = timeit(lambda: ops.bbb_linear_bwd(x, gy, y, wb, rb, bm, br, n_samples=S, prior=ops.PriorSpec(False, 1.0),
                                               math_mode=L.MATH_F32, relu=True, eps_mode=L.EPS_PHILOX, seed=1, layer_id=1,
                                               want_gx=True))
    print(f"K {K} N {N} S {S} B {B}: LR prep+weights {t_lr:.1f} us (+input {t_lrx - t_lr:.1f}); "
          f"BBB relu+weights {t_bbb:.1f} us (+input {t_bbbx - t_bbb:.1f})", flush=True)
