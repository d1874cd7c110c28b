"""BBB network (784-1200-1200-10, one minibatch of 128): us per evaluation (engine.GraphedElbo) by MC samples per evaluation with
the K-slice count of the K-sliced launches forced (tune build: BNN_TUNE_KSL; 0 = the plan's own choice).  Measurement tool."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine
from bench import build_net, DIMS

dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
net, x, y = build_net(DIMS["mnist"], False, 128, dev, "classification", n_minibatches=1)
for S in [int(v) for v in os.environ.get("SWEEP_S", "8,10,12,13,14,16,20,24").split(",")]:
    row = [f"S={S:2d}"]
    for ksl in [int(v) for v in os.environ.get("SWEEP_KSL", "0,1,2,3,4,5").split(",")]:
        if ksl:
            os.environ["BNN_TUNE_KSL"] = str(ksl)
        else:
            os.environ.pop("BNN_TUNE_KSL", None)
        try:
            ev = engine.GraphedElbo(net, x[0], y[0], S)
        except Exception as e:
            row.append(f"ksl {ksl}: n/a")
            continue
        for _ in range(10):
            ev.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        for _ in range(n):
            ev.replay()
        e1.record()
        e1.synchronize()
        row.append(f"ksl {ksl}: {e0.elapsed_time(e1) * 1e3 / n:6.1f}")
        del ev
    print(" | ".join(row), flush=True)
