#!/usr/bin/env python3
"""Layer-2 (1200x1200, batch 128, bf16) launch time by MC samples per launch and kernel form: HIP events around
back-to-back graph replays of the one launch.  With the tuning build (`make -C .../csrc tune`, BNN_HIP_LIB=...tune.so)
BNN_TUNE_KSL forces the slice count.  usage: few_sample_sweep.py [K N]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import torch
from bnn_hip import _lib as L, ops

K, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1200, 1200)
B = 128
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
wm = ((torch.rand((N, K), generator=g) - 0.5) * 0.4).to(dev)
wr = (torch.rand((N, K), generator=g) - 5.0).to(dev)
bm = torch.zeros(N, device=dev)
br = torch.full((N,), -4.5, device=dev)
sig = ops.softplus(wr)


def time_launch(fn, reps=200):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(10):
                fn()
        for _ in range(3):
            gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps // 10):
            gr.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps // 10 * 10)


alg = 8 * (K * N + N) + B * K * 2 + B * N * 2
for S in [int(v) for v in os.environ.get("SWEEP_S", "4,8,16,32,64").split(",")]:
    x16 = torch.rand(S, B, K, generator=g).to(dev).to(torch.bfloat16)
    y = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)
    ws = ops.bbb_workspace(S, N, dev)
    scratch = ops.split_scratch(S, B, N, dev)
    base = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16,
                eps_mode=L.EPS_PHILOX, seed=3, layer_id=2, want_stats=True, workspace=ws, out=y)
    row = [f"S={S:3d}"]
    for name, kw in (("tile", dict(form=L.FORM_TILE)), ("gemm", dict(form=L.FORM_GEMM)), ("gemm+sig", dict(form=L.FORM_GEMM, w_sigma=sig)),
                     ("kslice", dict(form=L.FORM_GEMM_KSLICE, split_scratch=scratch)),
                     ("kslice+sig", dict(form=L.FORM_GEMM_KSLICE, split_scratch=scratch, w_sigma=sig)),
                     ("auto+sig", dict(form=L.FORM_AUTO, split_scratch=scratch, w_sigma=sig))):
        try:
            pl = ops.bbb_plan(x16, wm, wr, bm, br, **base, **kw)
            us = time_launch(lambda: ops.bbb_linear_fwd(x16, wm, wr, bm, br, **base, **kw))
            row.append(f"{name}[f{pl['form']} ks{pl['k_slices']} b{pl['blocks']}] {us:6.1f}us {alg * S / us / 8e6:.2f}")
        except Exception as e:      # noqa: BLE001
            row.append(f"{name} ERR {type(e).__name__}")
    print(" | ".join(row), flush=True)
