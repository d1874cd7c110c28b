"""First LR layer (784 x 1200, batch 128) for S samples of ONE minibatch: K3s with the products made once (fp32 x read by the kernel,
epilogues spread over the slice blocks) against K3b over prepared fragments (needs the bf16 cast + squares and a prepare
launch ahead of it: timed separately).  Needs a build whose kLrsMaxShared admits the sample counts swept."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
from bnn_hip import ops, _lib as L
from bench import kernel_alone_us
dev = torch.device("cuda:0")
B, K, N = 128, 784, 1200
g = torch.Generator().manual_seed(3)
dw = [((torch.rand(K, N, generator=g) - 0.5) * 0.4).to(dev), (-5 + torch.rand(K, N, generator=g)).to(dev),
      ((torch.rand(N, generator=g) - 0.5) * 0.4).to(dev), (-5 + torch.rand(N, generator=g)).to(dev)]
st = torch.cuda.current_stream()
x = torch.rand(B, K, generator=g).to(dev)
x16 = x.to(torch.bfloat16); xsq = (x16.float() ** 2).to(torch.bfloat16)
frag, _ = ops.lr_prepare(*dw)
us_prep = kernel_alone_us(lambda: ops.lr_prepare(*dw), st, per_graph=8, reps=10)
us_cast = kernel_alone_us(lambda: ops.eval_prepare([], cast=x, want_sq=True), st, per_graph=8, reps=10)
print(f"prepare {us_prep:.1f} us, cast + squares {us_cast:.1f} us", flush=True)
scratch = ops.lr_split_scratch(1, B, N, dev)
for S in [int(v) for v in os.environ.get("SWEEP_S", "4,8,16,23,32,48,64").split(",")]:
    y = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev); ysq = torch.empty_like(y)
    kw = dict(n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=3, layer_id=1,
              want_kl=False, out=y, out_sq=ysq)
    row = [f"S={S:3d}"]
    try:
        pl = ops.lr_plan(x, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, **kw)
        us = kernel_alone_us(lambda: ops.lr_linear_fwd(x, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, **kw), st, per_graph=8, reps=10)
        row.append(f"K3s shared[b{pl['blocks']}] {us:6.1f}")
    except ops.BnnHipError as e:
        row.append("K3s shared: declined")
    pl = ops.lr_plan(x16, *dw, form=L.FORM_GEMM, w_frag=frag, x_sq=xsq, **kw)
    us = kernel_alone_us(lambda: ops.lr_linear_fwd(x16, *dw, form=L.FORM_GEMM, w_frag=frag, x_sq=xsq, **kw), st, per_graph=8, reps=10)
    row.append(f"K3b[w{pl['waves']} b{pl['blocks']}] {us:6.1f} (+ {us_prep + us_cast:.1f})")
    print(" | ".join(row), flush=True)
