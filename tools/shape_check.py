import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesian-neural-network_amd"))
from bnn_hip import ops, _lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for (K, N, B, S) in [(1, 50, 128, 1), (4, 50, 128, 2), (8, 64, 128, 1), (64, 1200, 128, 1), (33, 7, 100, 3), (1, 1, 128, 1), (50, 1, 128, 1),
                     (1200, 10, 128, 1), (784, 1200, 300, 1), (16, 16, 513, 1)]:
    x = torch.randn(B, K, device=dev)
    for lr in (False, True):
        for mm in (L.MATH_F32, L.MATH_BF16):
            wm = torch.randn((K, N) if lr else (N, K), device=dev) * 0.3
            wr = torch.full_like(wm, -3.0)
            bm = torch.randn(N, device=dev); br = torch.full((N,), -3.0, device=dev)
            if lr:
                out = ops.lr_linear_fwd(x, wm, wr, bm, br, n_samples=S, sigma_p=1.0, math_mode=mm, relu=False, y_dtype=torch.float32,
                                        eps_mode=L.EPS_ZERO, want_kl=True)
                ref = x @ wm + bm
            else:
                out = ops.bbb_linear_fwd(x, wm, wr, bm, br, n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=mm, relu=False,
                                         y_dtype=torch.float32, eps_mode=L.EPS_ZERO, want_stats=True)
                ref = x @ wm.t() + bm
            y = out["y"]
            err = float((y - ref.unsqueeze(0)).abs().max()); sc = float(ref.abs().max()) + 1e-6
            tol = 2e-5 if mm == L.MATH_F32 else 2e-2
            ok = err <= tol * sc
            bad += not ok
            print(("ok  " if ok else "BAD ") + f"K {K} N {N} B {B} S {S} {'LR ' if lr else 'BBB'} math {mm}: err {err:.3e} scale {sc:.3e}", flush=True)
print("bad:", bad)
