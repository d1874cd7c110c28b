"""K1b2 against K1b on one layer launch (tune build: BNN_TUNE_PAIRS switches the pair-sharing form off / on)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
from bnn_hip import ops, _lib as L
dev = torch.device("cuda:0")
for (S, B, K, N) in ((24, 128, 784, 1200), (6, 128, 64, 64)):
    g = torch.Generator().manual_seed(3)
    w_mu = ((torch.rand(N, K, generator=g) - 0.5) * 0.4).to(dev); w_rho = (-5 + torch.rand(N, K, generator=g)).to(dev)
    b_mu = ((torch.rand(N, generator=g) - 0.5) * 0.4).to(dev); b_rho = (-5 + torch.rand(N, generator=g)).to(dev)
    x = torch.rand(S, B, K, generator=g).to(dev).to(torch.bfloat16)
    sig = torch.log1p(torch.exp(w_rho))
    res = {}
    for pairs in ("0", "1"):
        os.environ["BNN_TUNE_PAIRS"] = pairs
        os.environ["BNN_TUNE_K1B"] = os.environ.get("EXPERIMENT", "0") if pairs == "1" else "0"
        kw = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16, relu=True, y_dtype=torch.float32, seed=1, layer_id=1,
                  eps_mode=L.EPS_PHILOX, want_stats=True, want_scalars=True, w_sigma=sig, form=L.FORM_GEMM)
        plan = ops.bbb_plan(x, w_mu, w_rho, b_mu, b_rho, **kw)
        out = ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw)
        torch.cuda.synchronize()
        res[pairs] = (out["y"].clone(), out["log_prior"].clone(), out["log_q"].clone(), plan)
    a, b = res["0"], res["1"]
    dy = (a[0] - b[0]).abs()
    bad = (dy > 0).nonzero()
    print(S, B, K, N, "plans", a[3]["waves"], b[3]["waves"], b[3]["blocks"], "max |dy|", float(dy.max()), "mismatching", int((dy > 0).sum()), "of", dy.numel(),
          "first bad", bad[0].tolist() if len(bad) else None, "dlp", float((a[1] - b[1]).abs().max()), "dlq", float((a[2] - b[2]).abs().max()), flush=True)
    if len(bad):
        s_, m_, n_ = bad[:, 0], bad[:, 1], bad[:, 2]
        print("   bad samples", sorted(set(s_.tolist()))[:12], "rows", sorted(set(m_.tolist())), "features", sorted(set(n_.tolist()))[:12], len(set(n_.tolist())))
        one = dy[1]
        print("   sample 1: bad rows", (one > 0).any(1).nonzero().flatten().tolist(), "bad count per row", (one > 0).sum(1)[:20].tolist())
