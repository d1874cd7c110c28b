"""The headline launch group (256 minibatches x 1 sample, one launch per layer) replayed as a hipGraph and as a recorded launch list
(engine.GraphedElbo(capture="calls")): us per launch group, alternating rounds on one box.  Measurement tool."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine
from bench import build_net, DIMS, make_evaluator, run_groups

dev = torch.device("cuda:0")
bnn_hip.set_math(os.environ.get("MATH", "bf16"))
lr = os.environ.get("VARIANT", "bbb") == "lr"
G = int(os.environ.get("GROUP", "256"))
net, x, y = build_net(DIMS["mnist"], lr, 128, dev, "classification", n_minibatches=G)
evs = {"graph": make_evaluator(engine, net, x, y, 1, G, graph=True), "calls": make_evaluator(engine, net, x, y, 1, G, graph="calls")}
for rnd in range(3):
    for name, ev in evs.items():
        dt = run_groups(ev, 40, 5, None)
        print(f"{'LR ' if lr else 'BBB'} G={G} {name:6s}: {dt * 1e6 / 40:8.1f} us per launch group = {G * 40 / dt:9.0f} samples/s", flush=True)
