"""One ELBO evaluation at a time (one minibatch of 128, one MC sample, each evaluation waiting for the previous one): us per
evaluation of the BBB and LR networks through engine.GraphedElbo, hipGraph replays back to back.  Measurement tool."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine
from bench import build_net, DIMS

dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for lr in (False, True):
    net, x, y = build_net(DIMS["mnist"], lr, 128, dev, "classification", n_minibatches=1)
    ev = engine.GraphedElbo(net, x[0], y[0], S, capture=os.environ.get("CAPTURE", "graph") if os.environ.get("CAPTURE") else True)
    for _ in range(20):
        ev.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 400
    e0.record()
    for _ in range(n):
        ev.replay()
    e1.record()
    e1.synchronize()
    print(f"{'LR ' if lr else 'BBB'} S={S}: {e0.elapsed_time(e1) * 1e3 / n:7.2f} us per evaluation", flush=True)
