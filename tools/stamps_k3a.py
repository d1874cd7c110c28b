"""Diagnostic only: where a block of K3a (one LR layer, few samples) spends its time, from in-kernel shader-clock stamps of
wave 0 (build: make -C bayesian-neural-network_amd/csrc stamps; never a timed build).
usage: stamps_k3a.py [n_samples] [K] [N]      K3_FORM=tile|kslice (default kslice: K3s), BNN_TUNE_LRKSL=n slices, X_F32=1 fp32 layer input"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BNN_HIP_LIB"] = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_stamps.so")
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import numpy as np, torch
from bnn_hip import ops, _lib as L

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
B = 128
torch.manual_seed(0)
wmu = torch.empty(K, N, device=dev).uniform_(-0.2, 0.2); wrho = torch.empty(K, N, device=dev).uniform_(-5, -4)
bmu = torch.empty(N, device=dev).uniform_(-0.2, 0.2); brho = torch.empty(N, device=dev).uniform_(-5, -4)
x16 = torch.rand(B, K, device=dev)
if os.environ.get("X_F32", "0") != "1":
    x16 = x16.to(torch.bfloat16)
ws = ops.lr_workspace(N, dev)
dbg = torch.zeros(8192 * 16, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
out = torch.empty(S, B, N, dtype=torch.bfloat16, device=dev); osq = torch.empty_like(out)
FORM = {"tile": L.FORM_TILE, "kslice": L.FORM_GEMM_KSLICE}[os.environ.get("K3_FORM", "kslice")]
kw = dict(n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=1,
          layer_id=1, want_kl=True, workspace=ws, out=out, out_sq=osq, form=FORM,
          split_scratch=ops.lr_split_scratch(S, B, N, dev) if FORM == L.FORM_GEMM_KSLICE else None)
def go():
    ops.lr_linear_fwd(x16, wmu, wrho, bmu, brho, **kw)
print(ops.lr_plan(x16, wmu, wrho, bmu, brho, **kw), "form:", os.environ.get("K3_FORM", "kslice"))
for _ in range(30): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): go()
e1.record(); e1.synchronize()
print("back-to-back launches: %.2f us each" % (e0.elapsed_time(e1) * 1e3 / 200))
dbg.zero_(); torch.cuda.synchronize()
go(); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
print("blocks stamped:", len(d))
names = {0: "start", 1: "phase 1 done / first parameters used", 2: "all loads issued", 3: "past the barrier / k loop done", 4: "slab written",
         5: "slab barrier", 6: "products done", 7: "end"}
if FORM == L.FORM_GEMM_KSLICE:
    names = {0: "start", 1: "all loads issued, parameters landed", 2: "operand tiles parked", 3: "past the barrier", 4: "products done",
             5: "partials stored, ticket taken", 6: "last arriver: slices summed", 7: "end"}
    last = d[d[:, 7] != 0]
    print("last arrivers:", len(last))
    for a, b in ((5, 6), (6, 7)):
        seg = last[:, b] - last[:, a]
        print(f"{names[a]:38s} -> {names[b]:38s} median {np.median(seg):8.0f} cyc   p10 {np.percentile(seg,10):8.0f}   p90 {np.percentile(seg,90):8.0f}")
    d[:, 7] = np.where(d[:, 7] != 0, d[:, 7], d[:, 5])
idx = [i for i in range(8) if (d[:, i] != 0).all() and not (FORM == L.FORM_GEMM_KSLICE and i in (6, 7))]
for a, b in zip(idx[:-1], idx[1:]):
    seg = d[:, b] - d[:, a]
    print(f"{names[a]:38s} -> {names[b]:38s} median {np.median(seg):8.0f} cyc   p10 {np.percentile(seg,10):8.0f}   p90 {np.percentile(seg,90):8.0f}")
tot = d[:, 7] - d[:, 0]
rt = (d[:, 9] - d[:, 8]).astype(np.float64)   # 100 MHz ticks
print(f"total (wave 0) median {np.median(tot):.0f} cyc; in-kernel clock GHz {np.median(tot / np.maximum(rt, 1) * 0.1):.2f}; block wall us (median) {np.median(rt) / 100.0:.2f}")
t0 = d[:, 8].min()
print("launch span, first block start -> last block end (us):", (d[:, 9].max() - t0) / 100.0)
print("block start times (us), deciles:", np.round(np.percentile((d[:, 8] - t0) / 100.0, [0, 10, 50, 90, 100]), 2))
