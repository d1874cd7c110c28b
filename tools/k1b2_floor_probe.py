"""The headline kernel (K1b2, 1200 x 1200, 256 pairs) and its vector-only skeleton under the profiler's counters: what does
a k-step of one wave really cost?  Launches the kernel N times per variant (tuning build: BNN_TUNE_K1B 0 = the product
kernel, 59 = everything but its vector work compiled out -- no DMA, no x reads, no MFMAs, no barrier, no waits --, 8 = no
DMA only; plus eps = 0 through the kernel's own arguments) as PLAIN launches, so that `rocprofv3 --pmc ...` attributes the
counters per dispatch; tools/k1b2_floor_summary.py turns the CSV into per-wave-step figures.

usage (GPU box):  BNN_HIP_LIB=.../libbnn_hip_tune.so rocprofv3 --pmc <counters> -d <dir> -- python3 tools/k1b2_floor_probe.py
      without the profiler it prints HIP-event times per variant."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
from bnn_hip import ops, _lib as L

dev = torch.device("cuda:0")
S, B, K, N = 256, 128, 1200, 1200
g = torch.Generator().manual_seed(3)
w_mu = ((torch.rand(N, K, generator=g) - 0.5) * 0.4).to(dev)
w_rho = (-5 + torch.rand(N, K, generator=g)).to(dev)
b_mu = ((torch.rand(N, generator=g) - 0.5) * 0.4).to(dev)
b_rho = (-5 + torch.rand(N, generator=g)).to(dev)
x = torch.rand(S, B, K, generator=g).to(dev).to(torch.bfloat16)
sig = torch.log1p(torch.exp(w_rho))
out = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)
ws = ops.bbb_workspace(S, N, dev)
REPS = int(os.environ.get("PROBE_REPS", "6"))
tune_lib = os.environ.get("BNN_HIP_LIB", "").endswith("tune.so")
variants = [("product", 0, L.EPS_PHILOX)]
if tune_lib:
    variants += [("vector work only", 59, L.EPS_PHILOX), ("no DMA", 8, L.EPS_PHILOX), ("memory side only (eps = 0)", 0, L.EPS_ZERO),
                 ("eps = 0, vector work only", 59, L.EPS_ZERO)]
else:
    variants += [("memory side only (eps = 0)", 0, L.EPS_ZERO)]
for name, tn, eps in variants:
    os.environ["BNN_TUNE_K1B"] = str(tn)
    kw = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, seed=1, layer_id=1,
              workspace=ws, out=out, form=L.FORM_GEMM, eps_mode=eps, want_stats=True, w_sigma=sig)
    assert ops.bbb_plan(x, w_mu, w_rho, b_mu, b_rho, **kw)["waves"] == 8
    ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw)
    e1.record()
    torch.cuda.synchronize()
    print(f"VARIANT {name:32s} tune {tn:2d} eps {eps}: {e0.elapsed_time(e1) * 1e3 / REPS:8.1f} us per launch ({REPS + 1} launches)", flush=True)
