"""Evaluation latency / throughput at one MC sample against the last-layer launch form."""
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
net, x, y = bench.build_net(bench.DIMS["mnist"], False, 128, dev, "classification")
for fuse, ks in (("1", "0"), ("1", "1"), ("1", "2"), ("1", "3"), ("1", "4"), ("1", "8"), ("0", "0")):
    os.environ["BNN_HIP_FUSE_FINAL"] = fuse
    if ks != "0":
        os.environ["BNN_HIP_FINAL_KS"] = ks
    else:
        os.environ.pop("BNN_HIP_FINAL_KS", None)
    e1 = bench.make_evaluators(engine, net, x, y, 1, 1)
    d1 = bench.run_steps(e1, 1000, 100, None)
    e3 = bench.make_evaluators(engine, net, x, y, 1, 3, per_replay=4)
    d3 = bench.run_steps(e3, 2400, 240, None)
    print(f"fuse {fuse} KS {ks}: alone {d1*1e6/1000:.2f} us; 3 in flight {d3*1e6/2400:.2f} us/evaluation", flush=True)
    del e1, e3
