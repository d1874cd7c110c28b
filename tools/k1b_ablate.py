"""K1b (block-GEMM form of the BBB layer, 1200 x 1200, 256 minibatch pairs) with parts switched off through its own
arguments: on-chip Philox eps against eps = 0 (no generator), statistics on / off, sigma hoisted or not.  Measurement
tool (HIP events around graph-captured back-to-back launches)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import ops, _lib as L
from bench import kernel_alone_us

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
for name, kw in (("philox, stats, hoisted sigma", dict(eps_mode=L.EPS_PHILOX, want_stats=True, w_sigma=sig)),
                 ("philox, no stats, hoisted sigma", dict(eps_mode=L.EPS_PHILOX, want_stats=False, w_sigma=sig)),
                 ("eps = 0, stats, hoisted sigma", dict(eps_mode=L.EPS_ZERO, want_stats=True, w_sigma=sig)),
                 ("eps = 0, no stats, hoisted sigma", dict(eps_mode=L.EPS_ZERO, want_stats=False, w_sigma=sig)),
                 ("philox, stats, softplus in the loop", dict(eps_mode=L.EPS_PHILOX, want_stats=True))):
    common = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, seed=1,
                  layer_id=1, workspace=ws if kw["want_stats"] else None, out=out, form=L.FORM_GEMM, **kw)
    plan = ops.bbb_plan(x, w_mu, w_rho, b_mu, b_rho, **common)
    tunes = [0]
    if os.environ.get("BNN_HIP_LIB", "").endswith("tune.so") and name.startswith("eps = 0, no stats") and plan["waves"] == 4:   # (K1b's knobs)
        # tuning build: BNN_TUNE_K1B bits -- 1 no barrier, 2 no vmcnt wait, 4 no parameter loads, 8 no x DMA, 16 no LDS reads, 32 no MFMAs
        tunes = [0, 1, 3, 4, 8, 12, 16, 32, 48, 60, 63]
    elif os.environ.get("BNN_HIP_LIB", "").endswith("tune.so") and name.startswith("philox") and "hoisted" in name and plan["waves"] == 4:
        tunes = [0, 12, 60, 63]      # the generator (and the statistics) with the memory path / everything else compiled out
    elif os.environ.get("BNN_HIP_LIB", "").endswith("tune.so") and "hoisted" in name and plan["waves"] == 8:
        tunes = [0, 8, 56, 59]       # K1b2: no DMA; also no x reads, no MFMAs; also no barrier, no waits (its in-situ vector floor)
    for tn in tunes:
        os.environ["BNN_TUNE_K1B"] = str(tn)
        us = kernel_alone_us(lambda: ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **common), torch.cuda.current_stream(), per_graph=8, reps=10)
        print(f"{name:40s} tune {tn:2d} form {plan['form']} blocks {plan['blocks']}: {us:8.1f} us per launch", flush=True)
