"""Diagnostic only: where a block of K1c (the output layer + finalize of one (minibatch, sample) pair, one block per pair) spends its
time inside the headline launch group (256 pairs), from thread 0's shader-clock stamps (build: make -C bayesian-neural-network_amd/csrc
stamps; never a timed build).  Stamps: 0 entry | 1 first k-step's parameters in registers | 2 its weights sampled | 3 k loop done |
4 slabs written | 5 barrier | 6 slabs reduced, logits in LDS | 10 logits complete | 11 fin_sample done (statistics of the other layers,
NLL)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BNN_HIP_LIB"] = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_stamps.so")
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import numpy as np, torch
import bnn_hip
from bnn_hip import engine
from bench import build_net, DIMS, make_evaluator

dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
G = 256
net, x, y = build_net(DIMS["mnist"], False, 128, dev, "classification", n_minibatches=G)
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
os.environ["BNN_HIP_DBG_PTR"] = str(dbg.data_ptr())
ev = make_evaluator(engine, net, x, y, 1, G, graph=False)
for _ in range(3):
    ev.replay()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)[:G].astype(np.float64)
names = {1: "entry -> first k-step's parameters in registers", 2: "-> its weights sampled", 3: "-> k loop done (all of the wave's k-steps)",
         4: "-> slabs written", 5: "-> barrier", 6: "-> slabs reduced, logits in LDS", 10: "-> logits complete", 11: "-> fin_sample done"}
prev = 0
clk = 100e6                                                      # nominal: on this device the counter runs near the shader clock -- read the PROPORTIONS (the launch takes 28.6 us)
for i in (1, 2, 3, 4, 5, 6, 10, 11):
    dt = (d[:, i] - d[:, prev]) / clk * 1e6
    print(f"  {names[i]:52s} median {np.median(dt):6.2f} us   p10 {np.percentile(dt, 10):6.2f}   p90 {np.percentile(dt, 90):6.2f}")
    prev = i
tot = (d[:, 11] - d[:, 0]) / clk * 1e6
print(f"  block total (entry -> fin_sample done)               median {np.median(tot):6.2f} us   p90 {np.percentile(tot, 90):6.2f}")
