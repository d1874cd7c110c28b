"""Diagnostic only: where a WAVE of the headline kernel (K1b2, 1200 x 1200, 256 pairs) spends a k-step, from shader-clock stamps
summed over its steps (build: make -C bayesian-neural-network_amd/csrc stamps; never a timed build; the stamps themselves cost a few
per cent).  usage: stamps_k1b2.py [bf16|bf16x3]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BNN_HIP_LIB"] = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_stamps.so")
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import numpy as np, torch
from bnn_hip import ops, _lib as L

dev = torch.device("cuda:0")
x3 = len(sys.argv) > 1 and sys.argv[1] == "bf16x3"
S, B, K, N = 256, 128, 1200, 1200
g = torch.Generator().manual_seed(3)
w_mu = ((torch.rand(N, K, generator=g) - 0.5) * 0.4).to(dev); w_rho = (-5 + torch.rand(N, K, generator=g)).to(dev)
b_mu = ((torch.rand(N, generator=g) - 0.5) * 0.4).to(dev); b_rho = (-5 + torch.rand(N, generator=g)).to(dev)
xf = torch.rand(S, B, K, generator=g).to(dev)
x = xf.to(torch.bfloat16)
x_lo = (xf - x.float()).to(torch.bfloat16) if x3 else None
sig = torch.log1p(torch.exp(w_rho))
out = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)
ws = ops.bbb_workspace(S, N, dev)
dbg = torch.zeros(1024 * 8 * 8, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
for eps, name in ((L.EPS_PHILOX, "on-chip epsilon"), (L.EPS_ZERO, "eps = 0")):
    kw = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16X3 if x3 else L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, seed=1,
              layer_id=1, workspace=ws, out=out, form=L.FORM_GEMM, eps_mode=eps, want_stats=True, w_sigma=sig, x_lo=x_lo)
    for _ in range(5):
        ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw)
    e1.record(); e1.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 5] != 0]
    steps = d[:, 5].astype(np.float64)
    names = ("staging issue + parameter reads (+ barrier B1 in the PS form)", "generator, w, statistics", "x reads + MFMAs", "closing wait (vmcnt / lgkmcnt)", "barrier")
    print(f"== {name}: {e0.elapsed_time(e1) * 1e2:.1f} us per launch (stamps build), {len(d)} waves stamped, {steps[0]:.0f} k-steps each")
    tot = 0.0
    for i, nm in enumerate(names):
        per = d[:, i] / steps
        tot += np.median(per)
        print(f"   {nm:64s} median {np.median(per):7.0f} cycles per k-step   p10 {np.percentile(per, 10):7.0f}   p90 {np.percentile(per, 90):7.0f}")
    print(f"   sum of medians {tot:7.0f} cycles per wave and k-step")
