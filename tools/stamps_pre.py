"""Diagnostic only: phase shares of one matmul-only launch (K1 over pre-sampled weights) from in-kernel
shader-clock stamps (make stamps build).  usage: stamps_pre.py [N] [conc]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BNN_HIP_LIB"] = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_stamps.so")
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import numpy as np, torch
from bnn_hip import ops, _lib as L

dev = torch.device("cuda:0")
K, N, B = 1200, int(sys.argv[1]) if len(sys.argv) > 1 else 1200, 128
CONC = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.manual_seed(0)
w = (torch.randn(1, N, K, device=dev) * 0.1).to(torch.bfloat16); b = torch.randn(1, N, device=dev)
x = torch.rand(1, B, K, device=dev).to(torch.bfloat16)
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
out = torch.empty(1, B, N, dtype=torch.bfloat16, device=dev)
scr = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
def go():
    ops.bbb_sampled_matmul(x, w, b, n_samples=1, relu=True, y_dtype=torch.bfloat16, out=out, concurrency=CONC)
for _ in range(300):
    scr.zero_()            # evict: the launch under test starts with cold caches, as behind a producer kernel
    go()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
print("blocks stamped:", len(d))
names = ["start->first weights arrive", "(no generator work)", "k-loop: x loads + MFMAs (all steps)", "mfma->prebarrier(bias,slab)", "barrier wait", "epilogue"]
tot = d[:, 6] - d[:, 0]
for i, nme in enumerate(names):
    seg = d[:, i + 1] - d[:, i]
    print(f"{nme:45s} median {np.median(seg):8.0f} cyc   p90 {np.percentile(seg,90):8.0f}")
print(f"{'total (wave 0)':45s} median {np.median(tot):8.0f} cyc")
rt = (d[:, 9] - d[:, 8]).astype(np.float64)   # 100 MHz ticks
print("kernel-wave wall us (median):", np.median(rt) / 100.0, " p90:", np.percentile(rt, 90) / 100.0)
print("first start -> last end over stamped blocks of LAST launch (us):", (d[:, 9].max() - d[:, 8].min()) / 100.0)
starts = (d[:, 8] - d[:, 8].min()) / 100.0
print("block start offsets us: median %.2f  p90 %.2f  max %.2f" % (np.median(starts), np.percentile(starts, 90), starts.max()))
