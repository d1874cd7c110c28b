"""Few-sample regime: us per ELBO evaluation of ONE minibatch (one hipGraph replayed back to back) for S = 1..16, with
the later layers' sampling riding on the first layer's launch (engine.PRESAMPLE_HIDDEN_MAX_SAMPLES >= S) or not."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO)
import torch
import bnn_hip
from bnn_hip import engine
import bench
dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
net, x, y = bench.build_net(bench.DIMS["mnist"], False, 128, dev, "classification", 1)
for S in (1, 2, 3, 4, 8, 16):
    row = []
    for thr in (0, 16):
        engine.PRESAMPLE_HIDDEN_MAX_SAMPLES = thr
        ev = engine.GraphedElbo(net, x[0], y[0], S)
        for _ in range(50):
            ev.replay()
        torch.cuda.synchronize()
        n = 400
        t0 = time.perf_counter()
        for _ in range(n):
            ev.replay()
        torch.cuda.synchronize()
        row.append((ev.pre_from, (time.perf_counter() - t0) / n * 1e6))
        del ev
    print(f"S={S:2d}: rider on layer {row[0][0]}: {row[0][1]:6.1f} us | on layer {row[1][0]}: {row[1][1]:6.1f} us", flush=True)
