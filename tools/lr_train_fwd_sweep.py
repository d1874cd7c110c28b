#!/usr/bin/env python3
"""LR layer forward as the training step launches it (fp32 activations in and out, the variance saved for the
backward, KL sums) by MC samples per launch: HIP events around graph replays of the one launch.  With the tuning build
(BNN_HIP_LIB=...libbnn_hip_tune.so) BNN_TUNE_LRR forces the number of k-range classes.  usage: lr_train_fwd_sweep.py [K N]"""
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
wm = ((torch.rand((K, N), generator=g) - 0.5) * 0.4).to(dev)
wr = (torch.rand((K, N), generator=g) - 5.0).to(dev)
bm = torch.zeros(N, device=dev)
br = torch.full((N,), -4.5, device=dev)


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
    return e0.elapsed_time(e1) * 1e3 / reps


for S in (1, 2, 3, 5, 8):
    for xdt in (torch.float32, torch.bfloat16):
        x = torch.rand((S, B, K), generator=g).to(dev).to(xdt)
        for want_v in (True, False):
            fn = lambda: ops.lr_linear_fwd(x, wm, wr, bm, br, n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True,
                                           y_dtype=torch.float32, eps_mode=L.EPS_PHILOX, seed=1, layer_id=1, sample_offset=0,
                                           want_kl=True, want_v=want_v)
            print(f"S={S} x={str(xdt)[6:]:8s} want_v={int(want_v)} R={os.environ.get('BNN_TUNE_LRR', 'auto')}: {time_launch(fn):6.1f} us", flush=True)
