"""Diagnostic: N graphed training steps of ONE configuration (for rocprofv3 --kernel-trace timelines).
usage: prof_train_graph.py [bbb|lr] [S] [steps]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO)
import torch, bnn_hip, networks
from bnn_hip import synth
from bnn_hip.optim import FusedAdam
from bnn_hip.train import GraphedTrainStep
dev = torch.device("cuda:0")
lr = (sys.argv[1] if len(sys.argv) > 1 else "bbb") == "lr"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
bnn_hip.set_math("bf16")
mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification",
          mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=lr)
net = networks.BayesianNetwork(mp).to(dev).train()
x, y = synth.synth_batch("classification", 128, 784, 10)
x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
opt = FusedAdam(net.parameters(), lr=1e-4, capturable=True)
g = GraphedTrainStep(net, opt, x, y, S)
for _ in range(20): g.step(x, y, 0.5)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): g.step(x, y, 0.5)
torch.cuda.synchronize()
print(f"{'LR' if lr else 'BBB'} S={S}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step")
