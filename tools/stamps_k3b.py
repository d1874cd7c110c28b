"""Diagnostic only: where a block of K3b (lr_fwd_gemm_kernel, prepared fragments) spends its time, from in-kernel
shader-clock stamps of wave 0 (build: make -C bayesian-neural-network_amd/csrc stamps; never a timed build).
usage: stamps_k3b.py [n_samples]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BNN_HIP_LIB"] = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_stamps.so")
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import numpy as np, torch
from bnn_hip import ops, _lib as L

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K, N, B = 1200, 1200, 128
torch.manual_seed(0)
wmu = torch.empty(K, N, device=dev).uniform_(-0.2, 0.2); wrho = torch.empty(K, N, device=dev).uniform_(-5, -4)
bmu = torch.empty(N, device=dev).uniform_(-0.2, 0.2); brho = torch.empty(N, device=dev).uniform_(-5, -4)
x = torch.rand(S, B, K, device=dev)
x16, xsq = ops.cast_bf16(x, want_sq=True)
wfrag, ws = ops.lr_prepare(wmu, wrho, bmu, brho)
dbg = torch.zeros(8192 * 16, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
out = torch.empty(S, B, N, dtype=torch.bfloat16, device=dev); osq = torch.empty_like(out)
def go():
    ops.lr_linear_fwd(x16, wmu, wrho, bmu, brho, n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True,
                      y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=1, layer_id=1, want_kl=True, workspace=ws, out=out,
                      x_sq=xsq, out_sq=osq, w_frag=wfrag, form=L.FORM_GEMM)
print(ops.lr_plan(x16, wmu, wrho, bmu, brho, n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16,
                  eps_mode=L.EPS_PHILOX, want_kl=True, workspace=ws, out=out, x_sq=xsq, out_sq=osq, w_frag=wfrag, form=L.FORM_GEMM))
for _ in range(30): go()
torch.cuda.synchronize()
dbg.zero_(); torch.cuda.synchronize()
go(); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
print("blocks stamped:", len(d))
segs = [("start -> prologue loads landed + barrier", 0, 1), ("k loop (all steps)", 1, 2), ("bias, activation noise, stores issued", 2, 3),
        ("stores drained", 3, 4)]
tot = d[:, 4] - d[:, 0]
for nme, i0, i1 in segs:
    seg = d[:, i1] - d[:, i0]
    print(f"{nme:45s} median {np.median(seg):9.0f} cyc   p10 {np.percentile(seg,10):9.0f}   p90 {np.percentile(seg,90):9.0f}")
print(f"{'total (wave 0)':45s} median {np.median(tot):9.0f} cyc")
rt = (d[:, 9] - d[:, 8]).astype(np.float64)   # 100 MHz ticks
print("in-kernel clock GHz (median):", np.median(tot / np.maximum(rt, 1) * 0.1), " block wall us (median):", np.median(rt) / 100.0)
t0 = d[:, 8].min()
print("launch span, first block start -> last block end (us):", (d[:, 9].max() - t0) / 100.0)
starts = np.sort((d[:, 8] - t0) / 100.0)
print("block start times (us), deciles:", np.round(np.percentile(starts, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]), 1))
print("sum of block walls / 256 CUs (us):", rt.sum() / 100.0 / 256)
