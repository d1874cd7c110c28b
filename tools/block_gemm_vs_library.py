"""K1g (ops.bbb_sampled_matmul, 256 x 256 block form) against the vendor BLAS reached through torch.mm on the same
operands: interleaved rounds in ONE process (the same device, the same clocks), HIP events around R back-to-back calls.
The library call computes only x . w^T per sample (no bias / ReLU / conversion); K1g includes them.  Measurement tool:
nothing here is on the product path.
usage: python tools/block_gemm_vs_library.py [S B N K]..."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
from bnn_hip import ops

dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(4, 1024, 4096, 4096), (4, 4096, 4096, 4096), (1, 4096, 4096, 4096)]
ROUNDS, REPS = 5, 20


def timed(fn, reps):
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record(s)
    for _ in range(reps):
        fn()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for S, B, N, K in shapes:
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.rand((S, B, K), generator=g).to(dev).to(torch.bfloat16)
    w = ((torch.rand((S, N, K), generator=g) - 0.5) * 0.4).to(dev).to(torch.bfloat16)
    b = ((torch.rand((S, N), generator=g) - 0.5) * 0.4).to(dev)
    y = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)
    ylib = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)

    def ours():
        ops.bbb_sampled_matmul(x, w, b, n_samples=S, relu=True, y_dtype=torch.bfloat16, out=y)

    def lib():
        for s_ in range(S):
            torch.mm(x[s_], w[s_].t(), out=ylib[s_])

    ours(); lib()
    torch.cuda.synchronize()
    ref = torch.relu(ylib.float() + b[:, None, :])
    err = float((y.float() - ref).abs().max()) / float(ref.abs().max())
    flops = 2.0 * S * B * N * K
    res = {"ours": [], "lib": []}
    for r in range(ROUNDS):
        res["ours"].append(timed(ours, REPS))
        res["lib"].append(timed(lib, REPS))
    for k, v in res.items():
        v.sort()
        med = v[len(v) // 2]
        print(f"S={S} B={B} N={N} K={K} {k:5s}: median {med:8.1f} us  min {v[0]:8.1f} us  {flops / med * 1e-6:7.1f} TFLOP/s "
              f"({flops / med * 1e-6 / 2500:.3f} of 2.5 PF)", flush=True)
    print(f"   max |ours - relu(lib + b)| / scale = {err:.2e}", flush=True)
