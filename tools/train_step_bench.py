"""Wall time of one training step (zero_grad + sample_elbo + backward + Adam) of the drop-in
BayesianNetwork at the MNIST configuration: eager with torch.optim.Adam (what the reference's
trainer does unchanged), eager with FusedAdam, and the whole step as one
hipGraph (train.GraphedTrainStep)."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO)
import torch, bnn_hip, networks
from bnn_hip import synth, engine
from bnn_hip.optim import FusedAdam
from bnn_hip.train import GraphedTrainStep
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
only = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for lr in (False, True):
    for math_mode in ("bf16", "f32"):
        for form in ("graph", "graph(autograd)", "eager+FusedAdam", "eager+torch.Adam", "per-layer nodes+torch.Adam"):
            if only and only != form:
                continue
            bnn_hip.set_math(math_mode)
            engine.FUSED_ELBO_NODE = form != "per-layer nodes+torch.Adam"
            mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification",
                      mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=lr)
            net = networks.BayesianNetwork(mp).to(dev).train()
            x, y = synth.synth_batch("classification", 128, 784, 10)
            x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
            if form.startswith("graph"):
                opt = FusedAdam(net.parameters(), lr=1e-4, capturable=True)
                g = GraphedTrainStep(net, opt, x, y, S, autograd=form != "graph")
                step = lambda: g.step(x, y, 0.5)
            else:
                opt = (FusedAdam if form == "eager+FusedAdam" else torch.optim.Adam)(net.parameters(), lr=1e-4)
                def step():
                    net.zero_grad()
                    out = (net.sample_elbo_lr if lr else net.sample_elbo)(x, y, 0.5, S)
                    out[0].backward(); opt.step()
            for _ in range(5): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 200 if form.startswith("graph") else 30
            for _ in range(n): step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            rows.append(dict(variant="LR" if lr else "BBB", mc_samples=S, math=math_mode, form=form, ms_per_step=dt * 1e3))
            print(f"{'LR ' if lr else 'BBB'} S={S} math={math_mode} {form:28s}: {dt*1e3:.3f} ms/step", flush=True)
engine.FUSED_ELBO_NODE = True
print(json.dumps(rows))
