"""Diagnostic only: phase shares of one K1 launch from in-kernel shader-clock stamps
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
CONC = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # the ABI's concurrency hint (4: the bench's tile plan)
torch.manual_seed(0)
wmu = torch.empty(N, K, device=dev).uniform_(-0.2, 0.2); wrho = torch.empty(N, K, device=dev).uniform_(-5, -4)
bmu = torch.empty(N, device=dev).uniform_(-0.2, 0.2); brho = torch.empty(N, device=dev).uniform_(-5, -4)
x = torch.rand(S, B, K, device=dev).to(torch.bfloat16)
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
ws = ops.bbb_workspace(S, N, dev); out = torch.empty(S, B, N, dtype=torch.bfloat16, device=dev)
def go():
    ops.bbb_linear_fwd(x, wmu, wrho, bmu, brho, n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16,
                       relu=True, y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=1, layer_id=1, want_stats=True,
                       workspace=ws, out=out, concurrency=CONC)
for _ in range(3000): go()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
print("blocks stamped:", len(d))
names = ["start->params", "params->w ready (softplus+philox+BM)", "w->mfma done(all steps)", "mfma->prebarrier(bias,slab)", "barrier wait", "epilogue"]
tot = d[:, 6] - d[:, 0]
for i, nme in enumerate(names):
    seg = d[:, i + 1] - d[:, i]
    print(f"{nme:45s} median {np.median(seg):8.0f} cyc   p90 {np.percentile(seg,90):8.0f}")
print(f"{'total (wave 0)':45s} median {np.median(tot):8.0f} cyc")
rt = (d[:, 9] - d[:, 8]).astype(np.float64)   # 100 MHz ticks
clk = tot / np.maximum(rt, 1) * 100e6 / 1e9
print("in-kernel clock GHz (median):", np.median(clk), " kernel-wave wall us (median):", np.median(rt) / 100.0)
span = (d[:, 9].max() - d[:, 8].min()) / 100.0
print("first start -> last end over stamped blocks of LAST launch (us):", span)
