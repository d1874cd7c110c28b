"""Throughput of one-sample evaluations (4 evaluations in flight) against the forced tile plan."""
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
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nstr = int(sys.argv[2]) if len(sys.argv) > 2 else 4
net, x, y = bench.build_net(bench.DIMS["mnist"], False, 128, dev, "classification")
for R in (0, 1, 2, 4):
    for nw in (0, 3, 4, 6, 8, 12):
        os.environ["BNN_HIP_BBB_R"] = str(R)
        os.environ["BNN_HIP_BBB_WAVES"] = str(nw)
        evs = bench.make_evaluators(engine, net, x, y, S, nstr)
        dt = bench.run_steps(evs, 2000, 200, None)
        e1 = bench.make_evaluators(engine, net, x, y, S, 1)
        d1 = bench.run_steps(e1, 600, 60, None)
        print(f"S {S} R {R} nw {nw}: {nstr} in flight {dt*1e6/2000:.2f} us/evaluation; alone {d1*1e6/600:.2f}", flush=True)
        del evs, e1
