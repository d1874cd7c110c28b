"""Diagnostic only: timeline of one chain launch (bnn_bbb_chain_fwd) from in-kernel real-time stamps (100 MHz) of
wave 0 of every block (build: make -C bayesian-neural-network_amd/csrc stamps; never a timed build)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BNN_HIP_LIB"] = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_stamps.so")
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO)
import numpy as np, torch, bnn_hip, networks
from bnn_hip import engine, synth
dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification", mu_init=[-0.2, 0.2],
          rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=False)
net = networks.BayesianNetwork(mp).to(dev).train()
x, y = synth.synth_batch("classification", 128, 784, 10)
x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
ev = engine.GraphedElbo(net, x, y, S, capture=False)
for _ in range(20): ev.replay()
torch.cuda.synchronize(); dbg.zero_(); torch.cuda.synchronize()
ev.replay(); torch.cuda.synchronize()
assert ev.chain
d = dbg.cpu().numpy().reshape(-1, 16).astype(np.float64)
live = d[:, 8] != 0
t0 = d[live, 8].min()
idx = np.nonzero(live)[0]
print("blocks stamped:", len(idx), "first..last block index", idx.min(), idx.max())
def show(name, sel, cols):
    if not sel.any(): return
    row = []
    for c, label in cols:
        v = (d[sel, c] - t0) / 100.0
        v = v[d[sel, c] != 0]
        if len(v): row.append(f"{label} med {np.median(v):6.2f} max {v.max():6.2f}")
    print(f"{name:28s} n={int(sel.sum()):4d} | " + " | ".join(row))
b = np.arange(len(d))
hid_lo = idx[idx >= 150].min() if (idx >= 150).any() else 0
show("layer 0 (blocks 0..149)", live & (b < 152), [(8, "start"), (9, "stored"), (12, "signalled")])
hidden = live & (b >= 160) & (d[:, 9] != 0) & (d[:, 11] != 0)
show("hidden layer (all)", hidden, [(8, "start"), (9, "weights drawn"), (10, "wait over"), (11, "matmul done"), (12, "signalled")])
starts = d[:, 8]
early = hidden & ((starts - t0) / 100.0 < 3.0)
show("hidden, resident from start", early, [(8, "start"), (9, "weights drawn"), (10, "wait over"), (11, "matmul done"), (12, "signalled")])
show("hidden, started later", hidden & ~early, [(8, "start"), (9, "weights drawn"), (10, "wait over"), (11, "matmul done"), (12, "signalled")])
rows = live & (d[:, 9] == 0) & (d[:, 10] != 0) & (b >= 160)
show("output layer rows/stats", rows, [(8, "start"), (10, "waits over"), (11, "published")])
print("launch span (first start -> last stamp), us:", (d[live][:, 8:13].max() - t0) / 100.0)
