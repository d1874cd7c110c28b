"""A/B on one box: the wide network's evaluation (4 MC samples, batch 1024 / 4096) with the sampling launches (K1s) on a side
stream beside the matmuls (engine.SAMPLE_BESIDE_MATMUL) against everything on one stream -- interleaved rounds."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine
import bench

dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
for B in (1024, 4096):
    net, x, y = bench.build_net(bench.DIMS["wide"], False, B, dev, "regression", n_minibatches=1)
    evs = {}
    for flag in (True, False):
        engine.SAMPLE_BESIDE_MATMUL = flag
        evs[flag] = engine.GraphedElbo(net, x[0], y[0], 4)
    for rnd in range(3):
        for flag in (True, False):
            ev = evs[flag]
            for _ in range(5):
                ev.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 30 if B == 1024 else 12
            for _ in range(n):
                ev.replay()
            torch.cuda.synchronize()
            print(f"batch {B} side stream {flag!s:5s} round {rnd}: {(time.perf_counter() - t0) * 1e6 / n:8.1f} us per evaluation", flush=True)
