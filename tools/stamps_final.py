"""Diagnostic only: phase stamps of the fused last-layer kernel (K1c) inside a one-sample evaluation
(-DBNN_STAMPS build).  Rows 0..KS-1 of the stamp buffer belong to its K-slice blocks (the earlier
layer kernels of the evaluation wrote the same rows before it)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BNN_HIP_LIB"] = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_stamps.so")
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import bench, bnn_hip
from bnn_hip import engine
dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
net, x, y = bench.build_net(bench.DIMS["mnist"], False, 128, dev, "classification")
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
ev = engine.GraphedElbo(net, x, y, 1, capture=False)
for _ in range(300): ev.replay()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)[:8]
names = {1: "params", 2: "w ready", 3: "mfma done", 4: "prebarrier", 5: "barrier", 6: "slab/epilogue", 7: "ticket taken", 10: "tiles summed (last block)", 11: "fin_sample done", 12: "fin: partials in", 13: "fin: nll done", 14: "fin: wave sums done", 15: "fin: folded"}
for b in range(8):
    if d[b, 0] == 0: continue
    t0 = d[b, 0]
    print(f"block {b}: " + "  ".join(f"{names[i]} {int(d[b, i] - t0)}" for i in (1, 2, 3, 4, 5, 6, 7, 10, 12, 13, 14, 15, 11) if d[b, i] > t0))
