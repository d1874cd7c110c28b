"""The headline launch group (256 minibatches x 1 sample) with the output layer + finalize as one block per pair (K1c + the sums
launch) against a sampling launch + the row-split form (K1r; engine.FINAL_ROWS_ALONE_MIN_SAMPLES): us per launch group in the steady
state, alternating rounds on one box.  Measurement tool."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine
from bench import build_net, DIMS, make_evaluator, run_groups, settle
dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
net, x, y = build_net(DIMS["mnist"], False, 128, dev, "classification", n_minibatches=256)
evs = {}
for name, v in (("K1c (one block per pair)", 10 ** 9), ("K1s + K1r (rows)", 64)):
    engine.FINAL_ROWS_ALONE_MIN_SAMPLES = v
    evs[name] = make_evaluator(engine, net, x, y, 1, 256, graph=True)
    assert evs[name].rows_alone == (v == 64)
settle(evs["K1c (one block per pair)"], ms=300)
for rnd in range(3):
    for name, ev in evs.items():
        run_groups(ev, 20, 0, None)
        dt = run_groups(ev, 40, 0, None)
        print(f"{name:28s}: {dt * 1e6 / 40:8.1f} us per launch group = {256 * 40 / dt:9.0f} samples/s", flush=True)
