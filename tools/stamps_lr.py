"""Diagnostic only: phase shares of one K3a (LR layer) launch from in-kernel shader-clock stamps
(build: hipcc -DBNN_STAMPS ... -o libbnn_hip_stamps.so; never a timed build)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BNN_HIP_LIB"] = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_stamps.so")
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import numpy as np, torch
from bnn_hip import ops, _lib as L

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K, N, B = 1200, int(sys.argv[2]) if len(sys.argv) > 2 else 1200, 128
torch.manual_seed(0)
wmu = torch.empty(K, N, device=dev).uniform_(-0.2, 0.2); wrho = torch.empty(K, N, device=dev).uniform_(-5, -4)
bmu = torch.empty(N, device=dev).uniform_(-0.2, 0.2); brho = torch.empty(N, device=dev).uniform_(-5, -4)
x = torch.rand(S, B, K, device=dev).to(torch.bfloat16)
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
ws = ops.lr_workspace(N, dev); out = torch.empty(S, B, N, dtype=torch.bfloat16, device=dev)
def go():
    ops.lr_linear_fwd(x, wmu, wrho, bmu, brho, n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16,
                      relu=True, y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=1, layer_id=1, want_kl=True,
                      workspace=ws, out=out)
for _ in range(3000): go()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
print("blocks stamped:", len(d))
segs = [("start -> first (M, rho) in registers", 0, 1), ("k-loop: sigma^2, x, x^2, MFMAs (all steps of wave 0)", 1, 3),
        ("bias, KL sums, mean slab write", 3, 4), ("barrier wait", 4, 5), ("mean reduce, variance slab + reduce", 5, 6),
        ("epilogue: eps_act, y store", 6, 7)]
tot = d[:, 7] - d[:, 0]
for nme, i0, i1 in segs:
    seg = d[:, i1] - d[:, i0]
    print(f"{nme:55s} median {np.median(seg):8.0f} cyc   p90 {np.percentile(seg,90):8.0f}")
print(f"{'total (wave 0)':55s} median {np.median(tot):8.0f} cyc")
rt = (d[:, 9] - d[:, 8]).astype(np.float64)   # 100 MHz ticks
clk = tot / np.maximum(rt, 1) * 100e6 / 1e9
print("in-kernel clock GHz (median):", np.median(clk), " kernel-wave wall us (median):", np.median(rt) / 100.0)
span = (d[:, 9].max() - d[:, 8].min()) / 100.0
print("first start -> last end over stamped blocks of LAST launch (us):", span)
