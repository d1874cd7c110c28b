"""BBB network (784-1200-1200-10, one minibatch of 128) at 6 .. 24 MC samples per evaluation through engine.GraphedElbo with the
sample count from which sigma = softplus(rho) is hoisted into the prepare launch at its product value and moved (HOIST_MIN=<n,...>):
us per evaluation.  Measurement tool."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine
from bench import build_net, DIMS

dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
mins = [engine.SIGMA_HOIST_MIN_SAMPLES] + [int(v) for v in os.environ.get("HOIST_MIN", "").split(",") if v]
net, x, y = build_net(DIMS["mnist"], False, 128, dev, "classification", n_minibatches=1)
for S in [int(v) for v in os.environ.get("SWEEP_S", "6,8,10,12,16,24").split(",")]:
    row = [f"S={S:2d}"]
    for rnd in range(2):
        for m in mins:
            engine.SIGMA_HOIST_MIN_SAMPLES = m
            ev = engine.GraphedElbo(net, x[0], y[0], S)
            for _ in range(20):
                ev.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 300
            e0.record()
            for _ in range(n):
                ev.replay()
            e1.record()
            e1.synchronize()
            row.append(f"hoist from {m}: {e0.elapsed_time(e1) * 1e3 / n:7.2f} us")
            del ev
    print(" | ".join(row), flush=True)
