"""Two evaluators of the headline workload (256 minibatches x 1 sample) replayed alternately on two streams against one evaluator
on one stream: does the next launch group's head fill the tail of this one's kernels?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine
import bench

dev = torch.device("cuda:0")
bnn_hip.set_math(os.environ.get("PROBE_MATH", "bf16"))
G = 256
lr = os.environ.get("PROBE_LR", "0") == "1"
net, x, y = bench.build_net(bench.DIMS["mnist"], lr, 128, dev, "classification", n_minibatches=G)
one = engine.GraphedElbo(net, x, y, 1, stacked=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a = engine.GraphedElbo(net, x, y, 1, stacked=True, stream=s1)
b = engine.GraphedElbo(net, x, y, 1, stacked=True, stream=s2)
for rnd in range(3):
    for name, evs in (("one stream", [one]), ("two streams", [a, b])):
        for _ in range(4):
            for e in evs:
                e.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for i in range(n):
            evs[i % len(evs)].replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name:12s} round {rnd}: {dt * 1e6 / n:8.1f} us per launch group -> {G * n / dt:9.0f} samples/s", flush=True)
