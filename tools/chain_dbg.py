import os, sys
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch, bnn_hip, networks
from bnn_hip import engine, synth
dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification", mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=False)
net = networks.BayesianNetwork(mp).to(dev).train()
x, y = synth.synth_batch("classification", 128, 784, 10)
x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
res = {}
for name, ms in (("sep", 0), ("chain", 3)):
    engine.CHAIN_MAX_SAMPLES = ms
    bnn_hip.manual_seed(11, counter=7)
    ev = engine.GraphedElbo(net, x, y, 1, capture=False)
    ev.replay(); torch.cuda.synchronize()
    res[name] = dict(sums=ev.sums.clone(), logits=ev.logits.clone(), h1=ev.bufs[0].float().clone(), h2=ev.bufs[1].float().clone(),
                     ws=[w.clone() for w in ev.ws], **{k: v.clone() for k, v in ev.out.items()})
    print(name, "chain" if ev.chain else "separate", ev.sums.tolist())
for k in ("sums", "logits", "h1", "h2", "log_prior", "log_q", "nll"):
    a, b = res["sep"][k], res["chain"][k]
    print(k, "max abs diff", float((a - b).abs().max()), "n diff", int((a != b).sum()), "of", a.numel())
for i in range(3):
    a, b = res["sep"]["ws"][i], res["chain"]["ws"][i]
    n = min(a.numel(), 4 * 400)
    print("ws", i, "n diff", int((a[:n] != b[:n]).sum()), "first", a[:8].tolist(), b[:8].tolist())
a, b = res["sep"]["ws"][0][4:4 + 4 * 150].view(-1, 4), res["chain"]["ws"][0][4:4 + 4 * 150].view(-1, 4)
print("ws0 per component n diff:", [(int((a[:, i] != b[:, i]).sum())) for i in range(4)])
idx = (a != b).any(1).nonzero().flatten()[:5].tolist()
for i in idx: print(i, a[i].tolist(), b[i].tolist())
