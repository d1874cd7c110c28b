"""Generate tests/golden/*.npz by RUNNING THE REAL REFERENCE in the build container.

TEST INFRASTRUCTURE.  Imports ``/root/reference/networks.py`` (CPU), loads numpy-seeded
parameters through ``load_state_dict``, injects epsilon through the reference's own
``.normal`` attribute seam (networks.py:35/42 and :100/123-124) and records what the
reference computes.  The reference never travels to the GPU box; these small fixtures do.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
Fixture set: G1..G9 of SURVEY.md §8(c).  Every stored output is what the reference
returned (fp32 values widened to float64 where they are scalars).
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import networks as ref  # noqa: E402  (the real reference)
sys.path.remove(REF)
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
for _m in ("networks", "config"):          # keep the reference's modules under private names only
    sys.modules.pop(_m, None)
from bnn_hip import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
torch.set_num_threads(1)       # reproducible reduction order for the recorded values


class Replay:
    """Stands in for ``torch.distributions.Normal(0,1)``: hands back pre-generated eps."""

    def __init__(self, arrays):
        self.q = [torch.from_numpy(np.ascontiguousarray(a)) for a in arrays]

    def sample(self, size):
        t = self.q.pop(0)
        assert tuple(t.shape) == tuple(size), (tuple(t.shape), tuple(size))
        return t


def f64(t):
    if isinstance(t, (int, float)):
        return np.float64(t)
    return t.detach().double().numpy().copy()


def rs_layer(rs, fin, fout, lr, mu=(-0.2, 0.2), rho=(-5, -4)):
    ws = (fin, fout) if lr else (fout, fin)
    return dict(weight_mu=rs.uniform(*mu, ws).astype(np.float32), weight_rho=rs.uniform(*rho, ws).astype(np.float32),
                bias_mu=rs.uniform(*mu, (fout,)).astype(np.float32), bias_rho=rs.uniform(*rho, (fout,)).astype(np.float32))


def load_layer(layer, p):
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})


# ------------------------------------------------------------------ G1/G2/G3/G7: BBB layer
def bbb_layer_case(fin, fout, B, prior_init, mixture, seed, train=True, sample=False, clp=False,
                   mu=(-0.2, 0.2), rho=(-5, -4), xr=(-1.0, 1.0)):
    rs = np.random.RandomState(seed)
    p = rs_layer(rs, fin, fout, False, mu, rho)
    x = rs.uniform(xr[0], xr[1], (B, fin)).astype(np.float32)
    ew = rs.standard_normal((fout, fin)).astype(np.float32)
    eb = rs.standard_normal((fout,)).astype(np.float32)
    layer = ref.BayesianLinear(fin, fout, [-0.2, 0.2], [-5, -4], prior_init, mixture)
    load_layer(layer, p)
    layer.train(train)
    layer.weight.normal = Replay([ew])
    layer.bias.normal = Replay([eb])
    with torch.no_grad():
        y = layer(torch.from_numpy(x), sample, clp)
    rec = dict(p)
    rec.update(x=x, eps_w=ew, eps_b=eb, y=f64(y), log_prior=f64(layer.log_prior),
               log_q=f64(layer.log_variational_posterior),
               log_prior_is_int=np.int64(isinstance(layer.log_prior, int)),
               prior_init=np.asarray(prior_init, np.float64), mixture=np.int64(mixture),
               train=np.int64(train), sample=np.int64(sample), clp=np.int64(clp))
    return rec


def lr_layer_case(fin, fout, B, sigma_p, seed, train=True, sample=False, clp=False, xr=(-1.0, 1.0)):
    rs = np.random.RandomState(seed)
    p = rs_layer(rs, fin, fout, True)
    x = rs.uniform(xr[0], xr[1], (B, fin)).astype(np.float32)
    ea = rs.standard_normal((B, fout)).astype(np.float32)
    eb = rs.standard_normal((fout,)).astype(np.float32)
    layer = ref.BayesianLinearLR(fin, fout, [-0.2, 0.2], [-5, -4], [sigma_p])
    load_layer(layer, p)
    layer.train(train)
    layer.normal = Replay([ea, eb])
    with torch.no_grad():
        y = layer(torch.from_numpy(x), sample, clp)
    rec = dict(p)
    rec.update(x=x, eps_act=ea, eps_b=eb, y=f64(y), weight_kl=f64(layer.weight_kl_cost),
               bias_kl=f64(layer.bias_kl_cost), kl=f64(layer.kl_cost), sigma_p=np.float64(sigma_p),
               train=np.int64(train), sample=np.int64(sample), clp=np.int64(clp))
    return rec


def flatten(prefix, rec, out):
    for k, v in rec.items():
        out[f"{prefix}/{k}"] = v


def g1_g2_g3_g4_g7_g8():
    out = {}
    # G1 Gaussian prior
    for i, sp in enumerate((1.0, 0.5)):
        flatten(f"G1/{i}", bbb_layer_case(7, 5, 3, [sp], False, 100 + i), out)
    # G2 mixture priors
    for i, pi3 in enumerate(([0.5, 0.0, -6.0], [0.5, 0.0, -8.0], [0.25, -1.0, -7.0])):
        flatten(f"G2/{i}", bbb_layer_case(7, 5, 3, pi3, True, 200 + i), out)
    # G3 mode truth table (BBB): {train,eval} x sample x calculate_log_probs
    i = 0
    for train in (True, False):
        for sample in (False, True):
            for clp in (False, True):
                flatten(f"G3/{i}", bbb_layer_case(7, 5, 3, [1.0], False, 300, train, sample, clp), out)
                i += 1
    # G4 LR layer: train; eval+sample (KL computed with clp, stale without)
    flatten("G4/0", lr_layer_case(7, 5, 3, 1.0, 400, True, False, False), out)
    flatten("G4/1", lr_layer_case(7, 5, 3, 0.7, 401, True, True, False), out)
    flatten("G4/2", lr_layer_case(7, 5, 3, 1.0, 402, False, True, True), out)
    rec = lr_layer_case(7, 5, 3, 1.0, 403, False, True, False)   # stale: kl attrs stay at their init 0
    flatten("G4/3", rec, out)
    # G7 odd shapes, asymmetric values (wide mu range so a transposed tile cannot pass)
    shapes = [(33, 65, 5), (1, 50, 8), (50, 1, 8), (1200, 10, 4), (37, 16, 17), (64, 48, 128), (40, 24, 130)]
    for i, (fin, fout, B) in enumerate(shapes):
        flatten(f"G7/bbb{i}", bbb_layer_case(fin, fout, B, [1.0], False, 700 + i, mu=(-1.0, 2.0), rho=(-3, 0.5)), out)
        flatten(f"G7/mix{i}", bbb_layer_case(fin, fout, B, [0.5, 0.0, -6.0], True, 720 + i, mu=(-0.3, 0.6)), out)
        flatten(f"G7/lr{i}", lr_layer_case(fin, fout, B, 1.0, 740 + i), out)
    # G8 rho extremes: sigma, log q at a fixed eps, incl. +inf at rho=89
    rho = np.asarray([-20, -5, 0, 5, 20, 60, 89], np.float32)
    mu = np.linspace(-0.3, 0.3, rho.size).astype(np.float32)
    eps = np.asarray([0.5, -1.0, 0.25, 2.0, -0.125, 1.5, 0.75], np.float32)
    node = ref.GaussianNode(torch.from_numpy(mu), torch.from_numpy(rho))
    node.normal = Replay([eps])
    w = node.sample()
    sig = node.sigma
    lq_terms = (-math.log(math.sqrt(2 * math.pi)) - torch.log(sig) - ((w - node.mu) ** 2) / (2 * sig ** 2))
    out["G8/rho"], out["G8/mu"], out["G8/eps"] = rho, mu, eps
    out["G8/sigma"], out["G8/w"], out["G8/log_q_terms"] = f64(sig), f64(w), f64(lq_terms)
    out["G8/log_q_sum_finite"] = f64(lq_terms[:6].sum())
    np.savez_compressed(os.path.join(OUT, "layers.npz"), **out)
    print("layers.npz", len(out), "arrays")


# ------------------------------------------------------------------ G5/G6: whole network
def make_net(input_shape, hidden, classes, mode, prior_init, mixture, lr, B):
    mp = dict(input_shape=input_shape, classes=classes, batch_size=B, hidden_units=hidden, mode=mode,
              mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=prior_init, mixture_prior=mixture,
              local_reparam=lr)
    net = ref.BayesianNetwork(mp)
    sd = synth.synth_state_dict(input_shape, hidden, classes, lr)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.train()
    return net


def eps_shapes(net, B, lr):
    shapes = []
    for l in (net.l1, net.l2, net.l3):
        shapes += [(B, l.weight_mu.shape[1]) if lr else tuple(l.weight_mu.shape), tuple(l.bias_mu.shape)]
    return shapes


def install_eps(net, B, S, lr):
    shapes = eps_shapes(net, B, lr)
    per_sample = [synth.synth_eps(shapes, s) for s in range(S)]
    for li, l in enumerate((net.l1, net.l2, net.l3)):
        if lr:
            l.normal = Replay([a for s in range(S) for a in (per_sample[s][2 * li], per_sample[s][2 * li + 1])])
        else:
            l.weight.normal = Replay([per_sample[s][2 * li] for s in range(S)])
            l.bias.normal = Replay([per_sample[s][2 * li + 1] for s in range(S)])


def run_elbo(net, x, y, beta, S, sigma, lr, want_grads):
    net.zero_grad()
    tup = net.sample_elbo_lr(x, y, beta, S, sigma) if lr else net.sample_elbo(x, y, beta, S, sigma)
    rec = {f"t{i}": f64(t) for i, t in enumerate(tup)}
    rec["t_shapes"] = np.asarray([t.dim() for t in tup], np.int64)
    if want_grads:
        tup[0].backward()
        for n, p in net.named_parameters():
            rec[f"grad/{n}"] = p.grad.detach().numpy().astype(np.float32).copy()
    return rec


def g5():
    out = {}
    for lr in (False, True):
        for B in (8, 128):
            for S in (1, 5):
                for bi, beta in enumerate((0.5, 0.25, 1e-3, 0.0)):
                    net = make_net(1, 50, 1, "regression", [1.0], False, lr, B)
                    x, y = synth.synth_batch("regression", B, 1, 1)
                    install_eps(net, B, S, lr)
                    want_grads = (B, S) in ((8, 5), (128, 1)) and bi in (0, 2)
                    rec = run_elbo(net, torch.from_numpy(x), torch.from_numpy(y), beta, S, 0.1, lr, want_grads)
                    rec["beta"] = np.float64(beta)
                    flatten(f"G5/{'lr' if lr else 'bbb'}/B{B}/S{S}/b{bi}", rec, out)
    # mixture prior through the whole (BBB) network, incl. grads (RLConfig default prior)
    for S in (1, 3):
        net = make_net(1, 50, 1, "regression", [0.5, 0.0, -6.0], True, False, 8)
        x, y = synth.synth_batch("regression", 8, 1, 1)
        install_eps(net, 8, S, False)
        rec = run_elbo(net, torch.from_numpy(x), torch.from_numpy(y), 0.5, S, 0.1, False, True)
        rec["beta"] = np.float64(0.5)
        flatten(f"G5/mix/B8/S{S}/b0", rec, out)
    np.savez_compressed(os.path.join(OUT, "net_c1.npz"), **out)
    print("net_c1.npz", len(out), "arrays")


def g6():
    out = {}
    torch.set_num_threads(8)
    cases = [("bbb", [1.0], False, False), ("mix", [0.5, 0.0, -6.0], True, False), ("lr", [1.0], False, True)]
    for name, prior_init, mixture, lr in cases:
        for S in (1, 2):
            B = 128
            net = make_net(784, 1200, 10, "classification", prior_init, mixture, lr, B)
            x, y = synth.synth_batch("classification", B, 784, 10)
            xt, yt = torch.from_numpy(x), torch.from_numpy(y)
            # per-layer scalars and logits of MC sample 0
            install_eps(net, B, 1, lr)
            with torch.no_grad():
                logits = net(xt, sample=True)
            rec = {"logits_s0_rows01": f64(logits[:2]), "logits_s0_absmax": f64(logits.abs().max()),
                   "nll_s0": f64(net.get_nll(logits, yt))}
            for li, l in enumerate((net.l1, net.l2, net.l3)):
                if lr:
                    rec[f"l{li+1}/kl"] = f64(l.kl_cost)
                else:
                    rec[f"l{li+1}/log_prior"] = f64(l.log_prior)
                    rec[f"l{li+1}/log_q"] = f64(l.log_variational_posterior)
            install_eps(net, B, S, lr)
            with torch.no_grad():
                tup = net.sample_elbo_lr(xt, yt, 0.5, S) if lr else net.sample_elbo(xt, yt, 0.5, S)
            for i, t in enumerate(tup):
                rec[f"t{i}"] = f64(t)
            if S == 2:
                # gradient evidence at full C2 size (the reference's loss.backward(), class_task.py:78): every 997th
                # element (997 is prime: the stride walks all rows and columns) and the L2 norm of all 12 .grad tensors
                install_eps(net, B, S, lr)
                net.zero_grad()
                tup = net.sample_elbo_lr(xt, yt, 0.5, S) if lr else net.sample_elbo(xt, yt, 0.5, S)
                tup[0].backward()
                for n, p_ in net.named_parameters():
                    g = p_.grad.detach().double().flatten()
                    rec[f"grad_sub/{n}"] = g[::997].numpy().astype(np.float64)
                    rec[f"grad_l2/{n}"] = np.float64(float(g.pow(2).sum().sqrt()))
            flatten(f"G6/{name}/S{S}", rec, out)
    torch.set_num_threads(1)
    np.savez_compressed(os.path.join(OUT, "net_c2.npz"), **out)
    print("net_c2.npz", len(out), "arrays")


def g9():
    out = {}
    M = 468
    idx = np.asarray([0, 1, 10, 148, 149, 467], np.int64)
    out["G9/M"] = np.int64(M)
    out["G9/idx"] = idx
    out["G9/beta"] = np.asarray([2 ** (M - (int(i) + 1)) / (2 ** M - 1) for i in idx], np.float64)
    # what an fp32 tensor times that python float gives: beta * 7.4e6
    out["G9/beta_times_7p4e6_f32"] = np.asarray(
        [float((b * torch.tensor(7.4e6, dtype=torch.float32)).item()) for b in out["G9/beta"]], np.float64)
    np.savez_compressed(os.path.join(OUT, "beta.npz"), **out)
    print("beta.npz", len(out), "arrays")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    g1_g2_g3_g4_g7_g8()
    g5()
    g6()
    g9()
