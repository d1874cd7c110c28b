import os, sys, cProfile, pstats, io
sys.path.insert(0, "bayesian-neural-network_amd"); sys.path.insert(0, ".")
import torch, bnn_hip, networks
from bnn_hip import synth
dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification",
          mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=False)
net = networks.BayesianNetwork(mp).to(dev).eval()
x, _ = synth.synth_batch("classification", 128, 784, 10)
x = torch.from_numpy(x).to(dev)
with torch.no_grad():
    for _ in range(20): net(x, sample=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): net(x, sample=True)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
