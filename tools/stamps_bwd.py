"""Diagnostic only: phase shares of one BBB weight-gradient launch (bbb_bwd_weights_kernel) from in-kernel
shader-clock stamps (build: make stamps -> libbnn_hip_stamps.so; never a timed build)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BNN_HIP_LIB"] = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_stamps.so")
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import numpy as np, torch
from bnn_hip import ops, _lib as L
from bnn_hip.ops import PriorSpec

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K, N, B = 1200, 1200, 128
torch.manual_seed(0)
wmu = torch.empty(N, K, device=dev).uniform_(-0.2, 0.2); wrho = torch.empty(N, K, device=dev).uniform_(-5, -4)
bmu = torch.empty(N, device=dev).uniform_(-0.2, 0.2); brho = torch.empty(N, device=dev).uniform_(-5, -4)
x = torch.rand(S, B, K, device=dev); gy = torch.randn(S, B, N, device=dev)
glp = torch.full((S,), 1e-3, device=dev); glq = torch.full((S,), -1e-3, device=dev)
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
def go():
    ops.bbb_linear_bwd(x, gy, None, wmu, wrho, bmu, brho, n_samples=S, prior=PriorSpec(sigma_p=1.0), math_mode=L.MATH_BF16,
                       relu=False, eps_mode=L.EPS_PHILOX, seed=1, layer_id=1, g_log_prior=glp, g_log_q=glq, want_gx=False)
for _ in range(300): go()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
print("blocks stamped:", len(d), " samples:", S)
segs = [("start -> params (mu, rho, softplus) + first group in registers", 0, 1), ("first group: 32 MFMAs", 1, 2),
        ("rest of sample 0's groups (loads + MFMAs)", 2, 3), ("sample 0 epilogue: Philox, fold into G/H", 3, 4),
        ("remaining samples + stores", 4, 7)]
tot = d[:, 7] - d[:, 0]
for nme, i0, i1 in segs:
    seg = d[:, i1] - d[:, i0]
    print(f"{nme:65s} median {np.median(seg):8.0f} cyc   p90 {np.percentile(seg,90):8.0f}")
print(f"{'total (wave 0)':65s} median {np.median(tot):8.0f} cyc")
rt = (d[:, 9] - d[:, 8]).astype(np.float64)   # 100 MHz ticks
print("kernel-wave wall us (median):", np.median(rt) / 100.0, " p90:", np.percentile(rt, 90) / 100.0)
print("first start -> last end over stamped blocks of LAST launch (us):", (d[:, 9].max() - d[:, 8].min()) / 100.0)
starts = (d[:, 8] - d[:, 8].min()) / 100.0
print("block start offsets us: median %.2f  p90 %.2f  max %.2f" % (np.median(starts), np.percentile(starts, 90), starts.max()))
