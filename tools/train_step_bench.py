"""Diagnostic: wall time of one training step (sample_elbo + backward + Adam) of the drop-in
BayesianNetwork at the MNIST config, HIP backward kernels vs the tensor-op backward."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO)
import torch, bnn_hip, networks
from bnn_hip import synth, functional as Fn
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for lr in (False, True):
    for math_mode in ("bf16", "f32"):
        for hip in (True, False):
            bnn_hip.set_math(math_mode); Fn.HIP_BACKWARD = hip
            mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification",
                      mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=lr)
            net = networks.BayesianNetwork(mp).to(dev).train()
            opt = torch.optim.Adam(net.parameters(), lr=1e-4)
            x, y = synth.synth_batch("classification", 128, 784, 10)
            x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
            def step():
                net.zero_grad()
                out = (net.sample_elbo_lr if lr else net.sample_elbo)(x, y, 0.5, S)
                out[0].backward(); opt.step()
            for _ in range(5): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 30
            for _ in range(n): step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print(f"{'LR ' if lr else 'BBB'} S={S} math={math_mode} backward={'HIP kernels' if hip else 'tensor ops '}: {dt*1e3:.3f} ms/step")
