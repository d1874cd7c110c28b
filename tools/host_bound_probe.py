"""Is the one-sample bench host-bound?  (a) host cost of replaying a graph of 4 trivial kernels;
(b) evaluations/s against the number of evaluator streams."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import bench
import bnn_hip
from bnn_hip import engine

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
a = torch.zeros(64, device=dev)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(4):
            a.add_(1.0)
torch.cuda.synchronize()
for n in (1, 3):
    strs = [torch.cuda.Stream() for _ in range(n)]
    t0 = time.perf_counter()
    for i in range(3000):
        with torch.cuda.stream(strs[i % n]):
            g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"trivial 4-kernel graph, {n} stream(s): host loop {1e6*(t1-t0)/3000:.2f} us/replay, total {1e6*(t2-t0)/3000:.2f}", flush=True)

bnn_hip.set_math("bf16")
net, x, y = bench.build_net(bench.DIMS["mnist"], False, 128, dev, "classification")
for variant in (False, True):
    net, x, y = bench.build_net(bench.DIMS["mnist"], variant, 128, dev, "classification")
    for E in (1, 2, 4, 8):
        for nstr in (1, 3, 4, 6, 8, 12):
            evs = bench.make_evaluators(engine, net, x, y, 1, nstr, per_replay=E)
            dt = bench.run_steps(evs, 4800, 480, None)
            print(f"{'LR ' if variant else 'BBB'} evals/graph {E} streams {nstr}: {dt*1e6/4800:.2f} us/evaluation", flush=True)
            del evs
