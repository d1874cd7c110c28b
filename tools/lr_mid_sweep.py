"""LR hidden layer (1200 x 1200, batch 128, per-sample bf16 x and x^2) at 4 .. 32 samples per launch: K3a (tile form) against K3b
over prepared fragments (block GEMM; tune build: BNN_TUNE_LRNW = waves per block) + the prepare launch.  HIP events around
graph-captured back-to-back launches."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
from bnn_hip import ops, _lib as L
from bench import kernel_alone_us
dev = torch.device("cuda:0")
B, K, N = 128, 1200, 1200
g = torch.Generator().manual_seed(3)
dw = [((torch.rand(K, N, generator=g) - 0.5) * 0.4).to(dev), (-5 + torch.rand(K, N, generator=g)).to(dev),
      ((torch.rand(N, generator=g) - 0.5) * 0.4).to(dev), (-5 + torch.rand(N, generator=g)).to(dev)]
st = torch.cuda.current_stream()
us_prep = kernel_alone_us(lambda: ops.lr_prepare(*dw), st, per_graph=8, reps=10)
print(f"lr_prepare: {us_prep:.1f} us", flush=True)
frag, wsp = ops.lr_prepare(*dw)
for S in [int(v) for v in os.environ.get("SWEEP_S", "4,8,10,16,24,32").split(",")]:
    x = torch.rand(S, B, K, generator=g).to(dev).to(torch.bfloat16)
    xsq = (x.float() ** 2).to(torch.bfloat16)
    y = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)
    ysq = torch.empty_like(y)
    kw = dict(n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=3, layer_id=2,
              want_kl=False, x_sq=xsq, out=y, out_sq=ysq)
    row = [f"S={S:3d}"]
    us = kernel_alone_us(lambda: ops.lr_linear_fwd(x, *dw, form=L.FORM_TILE, **kw), st, per_graph=8, reps=10)
    row.append(f"K3a {us:6.1f}")
    if os.environ.get("SWEEP_K3S"):               # tune build: K3s (32-feature groups x K slices) past its one-round limit
        os.environ["BNN_TUNE_LRS_MAXUNITS"], os.environ["BNN_TUNE_LRS_MAXBLOCKS"] = "400", "1200"
        scratch = ops.lr_split_scratch(S, B, N, dev)
        try:
            pl = ops.lr_plan(x, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, **kw)
            us = kernel_alone_us(lambda: ops.lr_linear_fwd(x, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, **kw), st, per_graph=8, reps=10)
            row.append(f"K3s ksl{pl['k_slices']}[b{pl['blocks']}] {us:6.1f}")
        except Exception as e:
            row.append(f"K3s n/a ({type(e).__name__})")
    for nw in (4, 8, 16):
        os.environ["BNN_TUNE_LRNW"] = str(nw)
        pl = ops.lr_plan(x, *dw, form=L.FORM_GEMM, w_frag=frag, **kw)
        us = kernel_alone_us(lambda: ops.lr_linear_fwd(x, *dw, form=L.FORM_GEMM, w_frag=frag, **kw), st, per_graph=8, reps=10)
        row.append(f"K3b nw{nw}[b{pl['blocks']}] {us:6.1f}")
    print(" | ".join(row), flush=True)
