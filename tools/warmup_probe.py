"""Does the headline need more than the driver's 5 warm-up steps?  One process: the evaluator is built, 5 untimed replays, then
consecutive timed windows of 20 replays each (us per launch group).  Measurement tool."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine
from bench import build_net, DIMS, make_evaluator, run_groups
dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
t0 = time.time()
net, x, y = build_net(DIMS["mnist"], False, 128, dev, "classification", n_minibatches=256)
ev = make_evaluator(engine, net, x, y, 1, 256, graph=True)
torch.cuda.synchronize()
print(f"built after {time.time() - t0:.2f} s", flush=True)
if os.environ.get("IDLE_MS"):
    time.sleep(float(os.environ["IDLE_MS"]) / 1e3)
row = []
dt = run_groups(ev, 20, 5, None)
row.append(dt * 1e6 / 20)
for _ in range(7):
    dt = run_groups(ev, 20, 0, None)
    row.append(dt * 1e6 / 20)
print("windows of 20 replays, us per launch group:", " ".join(f"{v:7.1f}" for v in row), flush=True)
# a SECOND evaluator of the same configuration, built while the device is in its steady state: does it start slow too?
ev2 = make_evaluator(engine, net, x, y, 1, 256, graph=True)
row = [run_groups(ev2, 20, 5, None) * 1e6 / 20] + [run_groups(ev2, 20, 0, None) * 1e6 / 20 for _ in range(5)]
print("second evaluator (fresh buffers, busy device):     ", " ".join(f"{v:7.1f}" for v in row), flush=True)
row = [run_groups(ev, 20, 0, None) * 1e6 / 20 for _ in range(3)]
print("back to the first evaluator:                       ", " ".join(f"{v:7.1f}" for v in row), flush=True)
time.sleep(1.0)
row = [run_groups(ev, 20, 0, None) * 1e6 / 20 for _ in range(4)]
print("the first evaluator after one second of idle:      ", " ".join(f"{v:7.1f}" for v in row), flush=True)
