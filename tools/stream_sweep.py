"""Run-to-run spread of one-sample evaluations/s against evaluator streams (HW-queue mapping)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import bench
import bnn_hip
from bnn_hip import engine
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bnn_hip.set_math("bf16")
lr = len(sys.argv) > 1 and sys.argv[1] == "lr"
net, x, y = bench.build_net(bench.DIMS["mnist"], lr, 128, dev, "classification")
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for rep in range(3):
    for nstr in (3, 4, 5, 6, 8):
        for E in (1, 4):
            evs = bench.make_evaluators(engine, net, x, y, 1, nstr, per_replay=E)
            dt = bench.run_steps(evs, 2400, 240, None)
            print(f"rep {rep} streams {nstr} evals/graph {E}: {dt*1e6/2400:.2f} us/evaluation", flush=True)
            del evs
