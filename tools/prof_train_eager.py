import os, sys, cProfile, pstats, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO)
import torch, bnn_hip, networks
from bnn_hip import synth
dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification",
          mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=False)
net = networks.BayesianNetwork(mp).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
x, y = synth.synth_batch("classification", 128, 784, 10)
x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
def step():
    net.zero_grad()
    out = net.sample_elbo(x, y, 0.5, 2)
    out[0].backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:5000])
