"""What would K3b (1200 x 1200, 256 pairs) gain if x^2 were formed on chip from the x tile (bf16 math defines it as bf16(bf16(x)^2),
so it could be)?  Python-side ablation: the same launch with x_sq aliased to x (its tile loads then hit the lines the x loads
brought: no second stream from the fabric) and / or without the y^2 output.  Wrong numbers, right traffic.  Measurement tool."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
from bnn_hip import ops, _lib as L
from bench import kernel_alone_us
dev = torch.device("cuda:0")
S, B, K, N = 256, 128, 1200, 1200
g = torch.Generator().manual_seed(3)
dw = [((torch.rand(K, N, generator=g) - 0.5) * 0.4).to(dev), (-5 + torch.rand(K, N, generator=g)).to(dev),
      ((torch.rand(N, generator=g) - 0.5) * 0.4).to(dev), (-5 + torch.rand(N, generator=g)).to(dev)]
st = torch.cuda.current_stream()
frag, wsp = ops.lr_prepare(*dw)
x = torch.rand(S, B, K, generator=g).to(dev).to(torch.bfloat16)
xsq = (x.float() ** 2).to(torch.bfloat16)
y = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)
ysq = torch.empty_like(y)
for rnd in range(2):
    for name, xq, oq in (("product: x^2 streamed, y^2 written", xsq, ysq), ("x^2 aliased to x", x, ysq), ("no y^2 output", xsq, None),
                         ("x^2 aliased to x, no y^2 output", x, None)):
        kw = dict(n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=3, layer_id=2,
                  want_kl=False, x_sq=xq, out=y, out_sq=oq, form=L.FORM_GEMM, w_frag=frag)
        us = kernel_alone_us(lambda: ops.lr_linear_fwd(x, *dw, **kw), st, per_graph=8, reps=10)
        print(f"{name:40s} {us:7.1f} us per launch", flush=True)
