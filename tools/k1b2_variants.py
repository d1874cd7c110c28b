"""K1b2 build variants (tools/build_k1b2_variants.sh) against the product library: outputs and statistics of one layer
launch bit for bit (each library in a child process, through BNN_HIP_LIB), then the 1200 x 1200 / 256-pair launch timed
(HIP events around graph-captured back-to-back launches).  usage: k1b2_variants.py p2r3 p4r2 p4r3 ..."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "bayesian-neural-network_amd")

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [PKG, REPO]
    import torch
    from bnn_hip import ops, _lib as L
    from bench import kernel_alone_us
    dev = torch.device("cuda:0")
    tag = sys.argv[2]
    outs = {}
    for ci, (S, B, K, N) in enumerate(((24, 128, 784, 1200), (6, 128, 64, 64), (9, 200, 100, 72), (256, 128, 1200, 1200))):
        g = torch.Generator().manual_seed(3 + ci)
        w_mu = ((torch.rand(N, K, generator=g) - 0.5) * 0.4).to(dev); w_rho = (-5 + torch.rand(N, K, generator=g)).to(dev)
        b_mu = ((torch.rand(N, generator=g) - 0.5) * 0.4).to(dev); b_rho = (-5 + torch.rand(N, generator=g)).to(dev)
        x = torch.rand(S, B, K, generator=g).to(dev).to(torch.bfloat16)
        sig = torch.log1p(torch.exp(w_rho))
        kw = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16, relu=True, y_dtype=torch.float32, seed=1, layer_id=1,
                  eps_mode=L.EPS_PHILOX, want_stats=True, want_scalars=True, w_sigma=sig, form=L.FORM_GEMM)
        plan = ops.bbb_plan(x, w_mu, w_rho, b_mu, b_rho, **kw)
        out = ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw)
        torch.cuda.synchronize()
        outs[ci] = dict(y=out["y"].cpu(), lp=out["log_prior"].cpu(), lq=out["log_q"].cpu(), plan=(plan["waves"], plan["blocks"]))
        if S == 256:
            yb = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)
            ws = ops.bbb_workspace(S, N, dev)
            for name, em, st in (("philox, stats", L.EPS_PHILOX, True), ("eps = 0, no stats", L.EPS_ZERO, False)):
                kt = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, seed=1, layer_id=1,
                          eps_mode=em, want_stats=st, workspace=ws if st else None, out=yb, w_sigma=sig, form=L.FORM_GEMM)
                us = kernel_alone_us(lambda: ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kt), torch.cuda.current_stream(), per_graph=8, reps=10)
                print(f"{tag:8s} {name:20s} waves {plan['waves']:2d} blocks {plan['blocks']:5d}: {us:8.1f} us per launch", flush=True)
    torch.save(outs, sys.argv[3])
    sys.exit(0)

import torch
variants = ["product"] + sys.argv[1:]
res = {}
for rnd in range(2):                      # two interleaved rounds of timings on the one box
    for v in variants:
        env = dict(os.environ)
        if v != "product":
            env["BNN_HIP_LIB"] = os.path.join(PKG, "bnn_hip", f"libbnn_hip_{v}.so")
        f = f"/tmp/k1b2_{v}.pt"
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v, f], env=env, check=True)
        res[v] = torch.load(f)
base = res["product"]
for v in variants[1:]:
    for ci in base:
        a, b = base[ci], res[v][ci]
        dy = (a["y"] - b["y"]).abs()
        print(f"{v}: case {ci} plans {a['plan']} / {b['plan']} mismatching y {int((dy > 0).sum())} of {dy.numel()} max {float(dy.max()):.3g}"
              f" dlp {float((a['lp'] - b['lp']).abs().max()):.3g} dlq {float((a['lq'] - b['lq']).abs().max()):.3g}", flush=True)
