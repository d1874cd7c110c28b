"""GPU parity: the HIP path (through the drop-in networks.py and the C ABI) against
(i) the golden vectors recorded from the real reference and (ii) the CPU oracle on
identical epsilon.  Tolerances (fp32 path): outputs and scalars rtol 2e-5 — the target
the north star states is rtol 1e-4 for KL/ELBO.  bf16-operand math: logits and NLL to 3x the error
measured at C2 (BF16_LOGIT_TOL, BF16_NLL_RTOL below), complexity scalars (always fp32) 2e-5."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bnn_hip
from bnn_hip import _lib as L
from bnn_hip import ops, synth
from oracle import bnn_oracle as O

F32_RTOL = 2e-5
# bf16 math mode against the FP32 oracle (= the reference's arithmetic): the deviation is a property of rounding the
# matmul OPERANDS (inputs, hidden activations, sampled weights) to bf16, not of the kernels -- the oracle's own CPU
# restatement with the same rounding points (O.network_forward_bf16) moves the NLL of one (minibatch, MC sample) pair of
# the C2 network by 3e-5 .. 2.2e-3 relative to O.network_forward (tests/test_oracle_golden.py::
# test_bf16_rounding_points_deviation_on_cpu; SURVEY 7.3's 5.5e-5 was a single draw at the low end) and the logits by
# ~5e-3 of their scale.  These two bounds hold that precision choice; the KERNELS are pinned separately and tightly
# against the rounding-point oracle (BF16_POINTS_* below, 1e-4).
BF16_NLL_RTOL = 5.5e-3
BF16_LOGIT_TOL = 1.8e-2
BF16_NLL_RTOL_GOLDEN = 2e-3      # the bound the golden C2 vectors (one fixed epsilon draw) have been held to since round 1


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _f32_math():
    bnn_hip.set_math("f32")
    yield
    bnn_hip.set_math("bf16")


class Replay:
    def __init__(self, arrays):
        self.q = [torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a for a in arrays]

    def sample(self, size):
        t = self.q.pop(0)
        assert tuple(t.shape) == tuple(size)
        return t


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, rtol=F32_RTOL, atol=0.0):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    np.testing.assert_allclose(a, np.asarray(b, np.float64), rtol=rtol, atol=atol)


def make_bbb(c, dev):
    import networks
    fout, fin = c["weight_mu"].shape
    layer = networks.BayesianLinear(fin, fout, [-0.2, 0.2], [-5, -4], list(c["prior_init"]), bool(c["mixture"]))
    layer.load_state_dict({k: t(c[k]) for k in synth.PARAM_NAMES})
    layer.to(dev)
    layer.train(bool(c["train"]))
    layer.weight.normal = Replay([c["eps_w"]])
    layer.bias.normal = Replay([c["eps_b"]])
    return layer


def make_lr(c, dev):
    import networks
    fin, fout = c["weight_mu"].shape
    layer = networks.BayesianLinearLR(fin, fout, [-0.2, 0.2], [-5, -4], [float(c["sigma_p"])])
    layer.load_state_dict({k: t(c[k]) for k in synth.PARAM_NAMES})
    layer.to(dev)
    layer.train(bool(c["train"]))
    layer.normal = Replay([c["eps_act"], c["eps_b"]])
    return layer


def run_bbb_case(c, dev, y_rtol=F32_RTOL, y_atol=2e-5, s_rtol=F32_RTOL):
    layer = make_bbb(c, dev)
    with torch.no_grad():
        y = layer(t(c["x"]).to(dev), bool(c["sample"]), bool(c["clp"]))
    close(y, c["y"], rtol=y_rtol, atol=y_atol)
    if int(c["log_prior_is_int"]):
        assert isinstance(layer.log_prior, int) and layer.log_prior == 0
        assert isinstance(layer.log_variational_posterior, int) and layer.log_variational_posterior == 0
    else:
        assert layer.log_prior.dim() == 0 and layer.log_variational_posterior.dim() == 0
        close(layer.log_prior, c["log_prior"], rtol=s_rtol)
        close(layer.log_variational_posterior, c["log_q"], rtol=s_rtol)


# ------------------------------------------------------------------ epsilon generator
def test_philox_matches_cpu_restatement(dev):
    for (tid, s0, S, rows, cols) in [(0, 0, 2, 5, 37), (9, 1000, 1, 16, 8), (2, 7, 3, 1, 10), (4, 0, 1, 130, 1201)]:
        e = ops.philox_normal(2026, tid, s0, S, rows, cols, dev).cpu().numpy()
        for s in range(S):
            np.testing.assert_allclose(e[s], O.philox_normal(2026, tid, s0 + s, rows, cols), atol=5e-5, rtol=0)
    big = ops.philox_normal(12345678901234567, 5, 0, 1, 1200, 1200, dev)
    assert abs(float(big.mean())) < 3e-3 and abs(float(big.std()) - 1.0) < 3e-3
    assert torch.isfinite(big).all()


# ------------------------------------------------------------------ K1 against the reference's vectors
def test_k1_g1_g2_gaussian_and_mixture(g_layers, dev):
    for g in ("G1", "G2"):
        for name in g_layers.cases(g):
            run_bbb_case(g_layers.case(name), dev)


def test_k1_g3_mode_truth_table(g_layers, dev):
    for name in g_layers.cases("G3"):
        run_bbb_case(g_layers.case(name), dev)


def test_k1_g7_odd_shapes(g_layers, dev):
    names = [n for n in g_layers.cases("G7") if "lr" not in n.split("/")[-1]]
    assert len(names) == 14
    for name in names:
        run_bbb_case(g_layers.case(name), dev, y_rtol=1e-4, y_atol=2e-4)


def test_k1_bf16_math_tolerance(g_layers, dev):
    bnn_hip.set_math("bf16")
    for name in g_layers.cases("G7"):
        if "lr" in name.split("/")[-1]:
            continue
        c = g_layers.case(name)
        scale = float(np.abs(c["y"]).max())
        layer = make_bbb(c, dev)
        with torch.no_grad():
            y = layer(t(c["x"]).to(dev))
        close(y, c["y"], rtol=0, atol=3e-2 * scale)
        close(layer.log_prior, c["log_prior"])                       # statistics stay fp32
        close(layer.log_variational_posterior, c["log_q"])


def test_k1_identity_input_catches_transposes(dev):
    """x = I (asymmetric W): y must equal W^T + b exactly in fp32 math."""
    import networks
    fin, fout = 48, 40
    layer = networks.BayesianLinear(fin, fout, [-1.0, 2.0], [-5, -4], [1.0], False).to(dev).eval()
    with torch.no_grad():
        y = layer(torch.eye(fin, device=dev))                         # eval & sample=False: w = mu
    close(y, (layer.weight_mu.t() + layer.bias_mu).detach().cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_k1_philox_mode_dump_and_oracle(dev):
    """On-chip eps: the kernel's dumped eps equals the Philox map, and the outputs equal
    the oracle evaluated on that dumped eps."""
    rs = np.random.RandomState(3)
    S, B, K, N = 3, 20, 72, 37
    wmu, wrho = rs.uniform(-0.5, 0.5, (N, K)).astype(np.float32), rs.uniform(-5, -2, (N, K)).astype(np.float32)
    bmu, brho = rs.uniform(-0.5, 0.5, N).astype(np.float32), rs.uniform(-5, -2, N).astype(np.float32)
    x = rs.uniform(-1, 1, (B, K)).astype(np.float32)
    prior = ops.PriorSpec(False, 0.8)
    out = ops.bbb_linear_fwd(t(x).to(dev), t(wmu).to(dev), t(wrho).to(dev), t(bmu).to(dev), t(brho).to(dev),
                             n_samples=S, prior=prior, math_mode=L.MATH_F32, relu=True, y_dtype=torch.float32,
                             eps_mode=L.EPS_PHILOX, seed=99, layer_id=1, sample_offset=5, want_stats=True,
                             want_scalars=True, dump_eps=True)
    ew, eb = out["eps_w"].cpu().numpy(), out["eps_b"].cpu().numpy()
    for s in range(S):
        np.testing.assert_allclose(ew[s], O.philox_normal(99, O.tensor_id(1, 0), 5 + s, N, K), atol=5e-5, rtol=0)
        np.testing.assert_allclose(eb[s], O.philox_normal(99, O.tensor_id(1, 1), 5 + s, 1, N)[0], atol=5e-5, rtol=0)
        y, lp, lq = O.bbb_linear(t(x), t(wmu), t(wrho), t(bmu), t(brho), t(ew[s]), t(eb[s]), O.Prior(False, 0.8))
        close(out["y"][s], torch.relu(y).numpy(), atol=2e-5)
        close(out["log_prior"][s], float(lp))
        close(out["log_q"][s], float(lq))
    fill = ops.philox_normal(99, O.tensor_id(1, 0), 5, S, N, K, dev)
    assert torch.equal(fill, out["eps_w"])                            # same code path, bitwise


# ------------------------------------------------------------------ K3 / K2
def test_k3_g4_lr_layer(g_layers, dev):
    for name in g_layers.cases("G4"):
        c = g_layers.case(name)
        layer = make_lr(c, dev)
        with torch.no_grad():
            y = layer(t(c["x"]).to(dev), bool(c["sample"]), bool(c["clp"]))
        close(y, c["y"], atol=2e-6)
        if bool(c["train"]) or bool(c["clp"]):
            close(layer.kl_cost, c["kl"])
            close(layer.weight_kl_cost, c["weight_kl"])
            close(layer.bias_kl_cost, c["bias_kl"], rtol=1e-4)
        else:
            assert layer.kl_cost == 0                                  # stale, as in the reference


def test_k3_g7_odd_shapes(g_layers, dev):
    names = [n for n in g_layers.cases("G7") if "lr" in n.split("/")[-1]]
    assert len(names) == 7
    for name in names:
        c = g_layers.case(name)
        layer = make_lr(c, dev)
        with torch.no_grad():
            y = layer(t(c["x"]).to(dev))
        close(y, c["y"], rtol=1e-4, atol=1e-4)
        close(layer.kl_cost, c["kl"])


def test_k3_eval_mean_path(dev):
    import networks
    layer = networks.BayesianLinearLR(33, 21, [-1.0, 2.0], [-5, -4], [1.0]).to(dev).eval()
    x = torch.rand(9, 33, device=dev)
    with torch.no_grad():
        y = layer(x)
    close(y, (x @ layer.weight_mu + layer.bias_mu).detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_k2_gauss_kl(dev):
    rs = np.random.RandomState(11)
    for n, sp in ((1, 1.0), (1027, 0.5), (1200 * 784, 1.0), (4096 * 4096 + 3, 2.0)):
        mu = rs.uniform(-0.2, 0.2, n).astype(np.float32)
        rho = rs.uniform(-5, -4, n).astype(np.float32)
        out = ops.gauss_kl(t(mu).to(dev), t(rho).to(dev), sp).cpu().numpy()
        sig = np.log1p(np.exp(rho.astype(np.float64)))
        kl = 0.5 * np.sum(2 * np.log(sp / sig) - 1 + (sig / sp) ** 2 + (mu.astype(np.float64) / sp) ** 2)
        np.testing.assert_allclose(out[0], kl, rtol=2e-6)
        np.testing.assert_allclose(out[1], np.log(sig).sum(), rtol=2e-6)
        np.testing.assert_allclose(out[2], (sig ** 2).sum(), rtol=2e-6)
        np.testing.assert_allclose(out[3], (mu.astype(np.float64) ** 2).sum(), rtol=2e-6)
    # against the fp32 oracle formula too (networks.py:113)
    mu, rho = rs.uniform(-0.2, 0.2, 5000).astype(np.float32), rs.uniform(-5, -4, 5000).astype(np.float32)
    ref = float(O.kl_closed_form(t(mu), O.softplus_naive(t(rho)), 0.0, 1.0))
    np.testing.assert_allclose(ops.gauss_kl(t(mu).to(dev), t(rho).to(dev), 1.0).cpu().numpy()[0], ref, rtol=2e-6)


def test_rho_extremes_g8(g_layers, dev):
    c = g_layers.case("G8")
    rho, mu, eps = c["rho"], c["mu"], c["eps"]
    n = rho.size
    # one output feature per rho value, in_features = 1, x = 1: y = w + b with b = 0 -> y = w
    out = ops.bbb_linear_fwd(torch.ones(1, 1, device=dev), t(mu).view(n, 1).to(dev), t(rho).view(n, 1).to(dev),
                             torch.zeros(n, device=dev), torch.full((n,), -200.0, device=dev), n_samples=1,
                             prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_F32, relu=False,
                             y_dtype=torch.float32, eps_mode=L.EPS_MEMORY, eps_w=t(eps).view(1, n, 1).to(dev),
                             eps_b=torch.zeros(1, n, device=dev), want_stats=False)
    w = out["y"][0, 0].cpu().numpy()
    np.testing.assert_allclose(w[:6], c["w"][:6], rtol=2e-6)
    assert np.isinf(w[6]) and w[6] > 0                                # rho = 89: naive softplus overflows


# ------------------------------------------------------------------ whole network, C1 (golden G5)
def build_net(dev, lr, dims, mode, prior_init=(1.0,), mixture=False, B=128):
    import networks
    mp = dict(input_shape=dims[0], classes=dims[2], batch_size=B, hidden_units=dims[1], mode=mode,
              mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=list(prior_init), mixture_prior=mixture,
              local_reparam=lr)
    net = networks.BayesianNetwork(mp)
    sd = synth.synth_state_dict(dims[0], dims[1], dims[2], lr)
    net.load_state_dict({k: t(v) for k, v in sd.items()})
    return net.to(dev).train(), sd


def install_eps(net, B, S, lr):
    shapes = []
    for l in (net.l1, net.l2, net.l3):
        shapes += [(B, l.weight_mu.shape[1]) if lr else tuple(l.weight_mu.shape), tuple(l.bias_mu.shape)]
    per = [synth.synth_eps(shapes, s) for s in range(S)]
    for li, l in enumerate((net.l1, net.l2, net.l3)):
        if lr:
            l.normal = Replay([a for s in range(S) for a in (per[s][2 * li], per[s][2 * li + 1])])
        else:
            l.weight.normal = Replay([per[s][2 * li] for s in range(S)])
            l.bias.normal = Replay([per[s][2 * li + 1] for s in range(S)])


@pytest.mark.parametrize("variant", ["bbb", "lr", "mix"])
def test_c1_regression_elbo_and_grads(g_c1, dev, variant):
    lr = variant == "lr"
    n_grad = 0
    for Bname in g_c1.cases(f"G5/{variant}"):
        B = int(Bname.split("/")[-1][1:])
        for Sname in g_c1.cases(Bname):
            S = int(Sname.split("/")[-1][1:])
            for bname in g_c1.cases(Sname):
                c = g_c1.case(bname)
                net, _ = build_net(dev, lr, (1, 50, 1), "regression",
                                   (0.5, 0.0, -6.0) if variant == "mix" else (1.0,), variant == "mix", B)
                x, y = synth.synth_batch("regression", B, 1, 1)
                install_eps(net, B, S, lr)
                want_grad = "grad/l1.weight_mu" in c
                net.zero_grad()
                with torch.set_grad_enabled(want_grad):
                    fn = net.sample_elbo_lr if lr else net.sample_elbo
                    tup = fn(t(x).to(dev), t(y).to(dev), float(c["beta"]), S, 0.1)
                assert [v.dim() for v in tup] == list(c["t_shapes"])
                for i, v in enumerate(tup):
                    close(v, c[f"t{i}"], rtol=3e-5)
                if want_grad:
                    tup[0].backward()
                    for n, p in net.named_parameters():
                        gref = c[f"grad/{n}"]
                        np.testing.assert_allclose(p.grad.cpu().numpy(), gref, rtol=2e-4,
                                                   atol=2e-5 * float(np.abs(gref).max()))
                    n_grad += 1
    assert n_grad >= 2


def test_lr_network_six_samples_on_one_minibatch_with_stubbed_normal(dev):
    """sample_elbo_lr with 6 samples and the layers' `.normal` stubbed (the reference's seam, networks.py:100): in bf16 math the
    first layer runs K3s over the one minibatch for all samples (products once, epilogues spread) on INJECTED epsilon; against
    the exact-fp32 math mode on the same epsilon (itself pinned by the golden vectors): KL to fp32 rounding, NLL within the
    operand-rounding bound of DESIGN 2, and against the oracle's fp32 arithmetic."""
    S, B = 6, 128
    x, y = synth.synth_batch("classification", B, 784, 10)
    xd, yd = t(x).to(dev), t(y).to(dev)
    got = {}
    for math_mode in ("f32", "bf16"):
        bnn_hip.set_math(math_mode)
        net, sd = build_net(dev, True, (784, 1200, 10), "classification")
        install_eps(net, B, S, True)
        with torch.no_grad():
            got[math_mode] = [v.double().cpu().numpy() for v in net.sample_elbo_lr(xd, yd, 0.25, S)]
    bnn_hip.set_math("bf16")
    p = O.NetParams.from_state_dict(sd, "classification", 784, True, O.Prior.from_init([1.0], False))
    eps = [[t(a) for a in synth.synth_eps(p.eps_shapes(B), s_)] for s_ in range(S)]
    torch.set_num_threads(8)
    ref = [v.double().numpy() for v in O.sample_elbo_lr(p, t(x), t(y), 0.25, S, eps=eps)]
    torch.set_num_threads(1)
    for i in range(3):
        close(got["f32"][i], ref[i], rtol=3e-5)
    close(got["bf16"][1], got["f32"][1], rtol=2e-5)                        # KL: fp32 statistics in both modes
    close(got["bf16"][2], got["f32"][2], rtol=BF16_NLL_RTOL_GOLDEN)        # NLL: bf16 operands


# ------------------------------------------------------------------ whole network, C2 (golden G6)
@pytest.mark.parametrize("variant", ["bbb", "mix", "lr"])
@pytest.mark.parametrize("math_mode", ["f32", "bf16"])
def test_c2_mnist_shape_elbo(g_c2, dev, variant, math_mode):
    bnn_hip.set_math(math_mode)
    lr = variant == "lr"
    net, _ = build_net(dev, lr, (784, 1200, 10), "classification",
                       (0.5, 0.0, -6.0) if variant == "mix" else (1.0,), variant == "mix")
    x, y = synth.synth_batch("classification", 128, 784, 10)
    xd, yd = t(x).to(dev), t(y).to(dev)
    c1 = g_c2.case(f"G6/{variant}/S1")
    f32 = math_mode == "f32"
    # MC sample 0: logits, per-layer scalars, NLL
    install_eps(net, 128, 1, lr)
    with torch.no_grad():
        logits = net(xd, sample=True)
        nll0 = net.get_nll(logits, yd)
    scale = float(c1["logits_s0_absmax"])
    close(logits[:2], c1["logits_s0_rows01"], rtol=0, atol=(2e-5 if f32 else 3e-2) * scale)
    close(nll0, c1["nll_s0"], rtol=2e-5 if f32 else BF16_NLL_RTOL_GOLDEN)
    for li, l in enumerate((net.l1, net.l2, net.l3)):
        if lr:
            close(l.kl_cost, c1[f"l{li+1}/kl"])
        else:
            close(l.log_prior, c1[f"l{li+1}/log_prior"])
            close(l.log_variational_posterior, c1[f"l{li+1}/log_q"])
    for S in (1, 2):
        c = g_c2.case(f"G6/{variant}/S{S}")
        install_eps(net, 128, S, lr)
        with torch.no_grad():
            tup = (net.sample_elbo_lr if lr else net.sample_elbo)(xd, yd, 0.5, S)
        nll_tol = 2e-5 if f32 else BF16_NLL_RTOL_GOLDEN
        tols = [1e-4, F32_RTOL, nll_tol] if lr else [1e-4, F32_RTOL, F32_RTOL, nll_tol]   # ELBO: rtol 1e-4 (north star)
        for i, v in enumerate(tup):
            close(v, c[f"t{i}"], rtol=tols[i])


@pytest.mark.parametrize("variant", ["bbb", "mix", "lr"])
def test_c2_gradients_through_the_hip_backward(g_c2, dev, variant):
    """loss.backward() at full C2 size (S = 2, beta = 0.5) through the HIP backward kernels, fp32 math, on the
    reference's epsilon, against gradient vectors recorded from the REAL reference (golden G6: every 997th element
    and the L2 norm of all 12 .grad tensors, classification/class_task.py:78)."""
    lr = variant == "lr"
    net, _ = build_net(dev, lr, (784, 1200, 10), "classification",
                       (0.5, 0.0, -6.0) if variant == "mix" else (1.0,), variant == "mix")
    x, y = synth.synth_batch("classification", 128, 784, 10)
    c = g_c2.case(f"G6/{variant}/S2")
    install_eps(net, 128, 2, lr)
    net.zero_grad()
    tup = (net.sample_elbo_lr if lr else net.sample_elbo)(t(x).to(dev), t(y).to(dev), 0.5, 2)
    close(tup[0], c["t0"], rtol=1e-4)
    tup[0].backward()
    for n, p_ in net.named_parameters():
        g = p_.grad.double().flatten().cpu()
        sub = c[f"grad_sub/{n}"]
        np.testing.assert_allclose(g[::997].numpy(), sub, rtol=2e-4, atol=2e-5 * float(np.abs(sub).max()), err_msg=n)
        close(float(g.pow(2).sum().sqrt()), c[f"grad_l2/{n}"], rtol=2e-5)


# ------------------------------------------------------------------ size-independent properties
def test_sample_offset_invariance_and_sharding(dev):
    """MC sample g gives the same result whichever launch / shard computes it."""
    net, _ = build_net(dev, False, (784, 1200, 10), "classification")
    bnn_hip.set_math("f32")        # (with bf16 hidden activations a last-bit change flips bf16 roundings)
    x, y = synth.synth_batch("classification", 128, 784, 10)
    xd = t(x).to(dev).view(128, 784)
    specs = net._specs()
    from bnn_hip import engine
    with torch.no_grad():
        full, st_full = engine.run_layers(specs, xd, 8, 100, want_stats=True, sample=True, differentiable=False)
        parts = [engine.run_layers(specs, xd, n, 100 + lo, want_stats=True, sample=True, differentiable=False)
                 for lo, n in ((0, 3), (3, 1), (4, 4))]
    got = torch.cat([p[0] for p in parts])
    # identical eps and identical products; only the fp32 accumulation ORDER may differ
    # between launch shapes (the tile geometry is a function of the shape, incl. n_samples)
    scale = float(full.abs().max())
    assert float((full - got).abs().max()) <= 5e-6 * scale
    fin = lambda st, n: ops.elbo_finalize(workspaces=st, layer_in=[784, 1200, 1200], layer_out=[1200, 1200, 10],
                                          local_reparam=False, prior=ops.PriorSpec(False, 1.0), n_samples=n,
                                          logits=None, target=None, mode=None)
    a = fin(st_full, 8)
    b = [fin(p[1], n) for p, n in zip(parts, (3, 1, 4))]
    close(torch.cat([q["log_prior"] for q in b]), a["log_prior"].cpu().numpy(), rtol=1e-6)
    close(torch.cat([q["log_q"] for q in b]), a["log_q"].cpu().numpy(), rtol=1e-6)
    # the same launch twice is bitwise reproducible (no atomics anywhere)
    with torch.no_grad():
        again, st_again = engine.run_layers(specs, xd, 8, 100, want_stats=True, sample=True, differentiable=False)
    assert torch.equal(full, again)
    a2 = fin(st_again, 8)
    assert torch.equal(a["log_prior"], a2["log_prior"]) and torch.equal(a["log_q"], a2["log_q"])


def test_mc_mean_converges_to_mean_weights(dev):
    """E_eps[y] = x.mu^T + b_mu: averaging many on-chip samples approaches the eval output."""
    import networks
    layer = networks.BayesianLinear(64, 32, [-0.2, 0.2], [-3, -2], [1.0], False).to(dev)
    x = torch.rand(16, 64, device=dev)
    out = ops.bbb_linear_fwd(x, layer.weight_mu.detach(), layer.weight_rho.detach(), layer.bias_mu.detach(),
                             layer.bias_rho.detach(), n_samples=4096, prior=layer._prior_spec, math_mode=L.MATH_F32,
                             relu=False, y_dtype=torch.float32, eps_mode=L.EPS_PHILOX, seed=1, want_stats=True,
                             want_scalars=True)
    mean = out["y"].mean(0)
    ref = x @ layer.weight_mu.detach().t() + layer.bias_mu.detach()
    sd = out["y"].std(0).max().item()
    assert (mean - ref).abs().max().item() < 6 * sd / math.sqrt(4096)
    # E[sum eps^2] = count: log q mean is near its expectation
    cnt = 64 * 32 + 32
    sig_w = torch.log1p(torch.exp(layer.weight_rho.detach()))
    sig_b = torch.log1p(torch.exp(layer.bias_rho.detach()))
    exp_lq = cnt * O.C0 - float(torch.log(sig_w).sum() + torch.log(sig_b).sum()) - 0.5 * cnt
    assert abs(out["log_q"].mean().item() - exp_lq) < 6 * math.sqrt(cnt / 2) / math.sqrt(4096) + 1e-3 * abs(exp_lq)


def test_cpu_tensors_raise_no_fallback():
    import networks
    layer = networks.BayesianLinear(4, 3, [-0.2, 0.2], [-5, -4], [1.0], False)
    with pytest.raises(bnn_hip.BnnHipError):
        layer(torch.rand(2, 4))


def test_fused_final_layer_equals_two_launches(dev, monkeypatch):
    """bnn_bbb_final_fwd (last layer + finalize in one launch, incl. the multi-sample ticket
    path and the device sample counter) gives the same scalars as the two-launch form (which the
    tile-form preference selects: the finalize is fused only under BNN_FORM_AUTO)."""
    from bnn_hip import engine
    for math_mode in ("f32", "bf16"):
        bnn_hip.set_math(math_mode)
        for S in (1, 5):
            net, _ = build_net(dev, False, (784, 1200, 10), "classification")
            x, y = synth.synth_batch("classification", 128, 784, 10)
            xd, yd = t(x).to(dev), t(y).to(dev)
            outs = []
            for form in (L.FORM_AUTO, L.FORM_TILE):
                monkeypatch.setattr(bnn_hip.runtime.state, "form", form)
                bnn_hip.manual_seed(77, counter=1000)
                ev = engine.GraphedElbo(net, xd, yd, S, capture=False)
                sums1 = ev.replay().clone()
                per1 = {k: v.clone() for k, v in ev.out.items()}
                sums2 = ev.replay().clone()                 # counter advanced on device: fresh eps
                outs.append((sums1, per1, sums2, ev.logits.clone(), int(ev.counter.item())))
            (a1, p1, a2, lg_a, c_a), (b1, p2, b2, lg_b, c_b) = outs
            assert c_a == c_b == 1000 + 2 * S                # 2 replays
            scale = float(lg_a.abs().max())                  # different tile geometry: fp32 sum order only
            assert float((lg_a - lg_b).abs().max()) <= (5e-6 if math_mode == "f32" else 2e-3) * scale
            tol = 1e-6 if math_mode == "f32" else 1e-3       # nll follows the logits
            for k in p1:
                close(p1[k], p2[k].cpu().numpy(), rtol=tol if k == "nll" else 1e-6)
            close(a1, b1.cpu().numpy(), rtol=tol)
            close(a2, b2.cpu().numpy(), rtol=tol)
            assert not torch.equal(a1, a2)                   # second replay drew different eps
            close(a1[:3], torch.stack([p1["log_prior"].sum(), p1["log_q"].sum(), p1["nll"].sum()]).cpu().numpy(),
                  rtol=1e-6)


@pytest.mark.parametrize("force_gemm,S", [(L.FORM_TILE, 8), (L.FORM_GEMM, 8), (L.FORM_GEMM, 24)])
def test_lr_throughput_path_against_oracle(dev, monkeypatch, force_gemm, S):
    """LR network, S MC samples in one evaluation, bf16 math with the bf16 x / x^2 activation pair:
    the K-split kernel ("0"), the LDS-DMA block-GEMM kernel ("1") and, at S=24, the prepared-
    fragment form of it (bnn_lr_prepare), against the oracle on injected eps."""
    monkeypatch.setattr(bnn_hip.runtime.state, "form", force_gemm)
    bnn_hip.set_math("bf16")
    B = 128
    net, sd = build_net(dev, True, (784, 1200, 10), "classification")
    x, y = synth.synth_batch("classification", B, 784, 10)
    p = O.NetParams.from_state_dict(sd, "classification", 784, True, O.Prior.from_init([1.0], False))
    eps = [[t(a) for a in synth.synth_eps(p.eps_shapes(B), s)] for s in range(S)]
    torch.set_num_threads(8)
    ref = O.sample_elbo_lr(p, t(x), t(y), 0.5, S, eps=eps)
    torch.set_num_threads(1)
    install_eps(net, B, S, True)
    with torch.no_grad():
        got = net.sample_elbo_lr(t(x).to(dev), t(y).to(dev), 0.5, S)
    close(got[1], ref[1].numpy())                       # KL: fp32 statistics
    close(got[2], ref[2].numpy(), rtol=BF16_NLL_RTOL)   # NLL follows bf16 logits
    close(got[0], ref[0].numpy(), rtol=1e-4)            # ELBO


@pytest.mark.parametrize("eps_mode", ["philox", "memory"])
@pytest.mark.parametrize("prior", [ops.PriorSpec(False, 0.8), ops.PriorSpec(True, 1.0, 0.5, 1.0, math.exp(-6.0))])
@pytest.mark.parametrize("shape", [(3, 20, 72, 37, True), (2, 128, 1200, 1200, True), (1, 8, 50, 1, False), (2, 5, 33, 65, False),
                                   (3, 20, 72, 36, True), (2, 100, 200, 40, False), (1, 130, 784, 1200, True)])
def test_bbb_backward_kernels_match_tensor_op_gradients(dev, prior, shape, eps_mode):
    """F1: bnn_bbb_linear_bwd (eps regenerated on chip, fp32 matrix core) against the closed-form
    gradients evaluated with tensor ops on the same Philox eps — and through them against
    autograd of the oracle (the tensor-op path is what golden G5 pins)."""
    from bnn_hip import functional as Fn
    S, B, K, N, relu = shape
    rs = np.random.RandomState(5)
    mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
    x = mk(S, B, K, lo=-1, hi=1)
    w_mu, w_rho, b_mu, b_rho = mk(N, K), mk(N, K, lo=-5, hi=-2), mk(N), mk(N, lo=-5, hi=-2)
    mem = eps_mode == "memory"
    eps_w = torch.from_numpy(rs.standard_normal((S, N, K)).astype(np.float32)).to(dev) if mem else None
    eps_b = torch.from_numpy(rs.standard_normal((S, N)).astype(np.float32)).to(dev) if mem else None
    call = Fn.LayerCall(n_samples=S, prior=prior, math_mode=L.MATH_F32, relu=relu,
                        eps_mode=L.EPS_MEMORY if mem else L.EPS_PHILOX, seed=11, layer_id=2, sample_offset=40, want_stats=True)
    gy, glp, glq = mk(S, B, N, lo=-1, hi=1), mk(S, lo=-0.5, hi=0.5), mk(S, lo=-0.5, hi=0.5)
    import tensor_op_grads as T
    leaves = [t_.clone().requires_grad_(True) for t_ in (x, w_mu, w_rho, b_mu, b_rho)]
    y, lp, lq = Fn.BBBLinearFn.apply(*leaves, eps_w, eps_b, call)
    ((y * gy).sum() + (lp * glp).sum() + (lq * glq).sum()).backward()
    got = [l.grad.clone() for l in leaves]
    with torch.no_grad():
        want = T.bbb_grads(x, gy, glp, glq, y.detach(), w_mu, w_rho, b_mu, b_rho, eps_w, eps_b, call)
    for name, a, b in zip(("x", "w_mu", "w_rho", "b_mu", "b_rho"), got, want):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max())
        assert err <= 3e-5 * scale + 1e-6, (name, err, scale)


@pytest.mark.parametrize("eps_mode", ["philox", "memory"])
@pytest.mark.parametrize("shape", [(3, 20, 72, 38, True), (2, 128, 1200, 1200, True), (2, 128, 784, 1200, True),
                                   (1, 8, 50, 1, False), (2, 5, 33, 65, False), (1, 7, 1, 50, True),
                                   # narrow output layers (on-chip eps: one launch, lr_out_layer_bwd_kernel)
                                   (2, 128, 1200, 10, False), (3, 100, 72, 16, True), (5, 20, 33, 3, True)])
def test_lr_backward_kernels_match_tensor_op_gradients(dev, shape, eps_mode):
    """F1: bnn_lr_linear_bwd (eps regenerated on chip, v saved by the forward, fp32 matrix core)
    against the closed-form gradients evaluated with tensor ops on the same eps (the path golden
    G5 pins against autograd of the oracle)."""
    from bnn_hip import functional as Fn
    S, B, K, N, relu = shape
    rs = np.random.RandomState(6)
    mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
    x = mk(S, B, K, lo=-1, hi=1)
    w_mu, w_rho, b_mu, b_rho = mk(K, N), mk(K, N, lo=-5, hi=-2), mk(N), mk(N, lo=-5, hi=-2)
    mem = eps_mode == "memory"
    eps_act = torch.from_numpy(rs.standard_normal((S, B, N)).astype(np.float32)).to(dev) if mem else None
    eps_b = torch.from_numpy(rs.standard_normal((S, N)).astype(np.float32)).to(dev) if mem else None
    call = Fn.LayerCall(n_samples=S, prior=ops.PriorSpec(False, 0.8), math_mode=L.MATH_F32, relu=relu,
                        eps_mode=L.EPS_MEMORY if mem else L.EPS_PHILOX, seed=13, layer_id=1, sample_offset=70,
                        want_stats=True)
    gy, gkl = mk(S, B, N, lo=-1, hi=1), mk(3, lo=0.1, hi=0.5)
    import tensor_op_grads as T
    leaves = [t_.clone().requires_grad_(True) for t_ in (x, w_mu, w_rho, b_mu, b_rho)]
    y, kl3 = Fn.LRLinearFn.apply(*leaves, eps_act, eps_b, call)
    ((y * gy).sum() + (kl3 * gkl).sum()).backward()
    got = [l.grad.clone() for l in leaves]
    with torch.no_grad():
        want = T.lr_grads(x, gy, gkl, y.detach(), w_mu, w_rho, b_mu, b_rho, eps_act, eps_b, call)
    for name, a, b in zip(("x", "w_mu", "w_rho", "b_mu", "b_rho"), got, want):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max())
        assert err <= 5e-5 * scale + 1e-6, (name, err, scale)


def test_bbb_throughput_forms_agree(dev, monkeypatch):
    """24 MC samples per evaluation, on-chip eps: the LDS-DMA block-GEMM form with hoisted sigma
    (and the tail-kernel sums) against the K-split kernel on the same Philox sample indices, and
    both against the fp32-math K-split result."""
    from bnn_hip import engine
    S = 24
    net, _ = build_net(dev, False, (784, 1200, 10), "classification")
    x, y = synth.synth_batch("classification", 128, 784, 10)
    xd, yd = t(x).to(dev), t(y).to(dev)
    res = {}
    for name, math_mode, gemm in (("gemm", "bf16", L.FORM_GEMM), ("ksplit", "bf16", L.FORM_TILE), ("f32", "f32", L.FORM_TILE)):
        monkeypatch.setattr(bnn_hip.runtime.state, "form", gemm)
        bnn_hip.set_math(math_mode)
        bnn_hip.manual_seed(5, counter=300)
        ev = engine.GraphedElbo(net, xd, yd, S, capture=False)
        ev.replay()
        res[name] = ({k: v.clone() for k, v in ev.out.items()}, ev.sums.clone(), int(ev.counter.item()))
    for name in ("gemm", "ksplit"):
        out, sums, ctr = res[name]
        ref, rsums, rctr = res["f32"]
        assert ctr == rctr == 300 + S
        close(out["log_prior"], ref["log_prior"].cpu().numpy(), rtol=2e-6)      # statistics are fp32 in every form
        close(out["log_q"], ref["log_q"].cpu().numpy(), rtol=2e-6)
        close(out["nll"], ref["nll"].cpu().numpy(), rtol=BF16_NLL_RTOL)         # bf16 logits
        close(sums[:2], rsums[:2].cpu().numpy(), rtol=2e-6)
        close(sums[2], float(out["nll"].double().sum()), rtol=1e-6)
        assert float(sums[3]) == S


@pytest.mark.parametrize("lr", [False, True])
@pytest.mark.parametrize("S", [1, 8])
def test_recorded_launch_list_replays_what_the_graph_replays(dev, lr, S):
    """engine.GraphedElbo(capture="calls"): the evaluation as a recorded list of C-ABI launches called again -- bit for bit the
    replay of the captured hipGraph at the same Philox sample indices (sums, per-sample scalars, logits), replay after replay,
    and the device-resident sample counter advances the same way."""
    from bnn_hip import engine
    B, dims, seed, c = 128, (784, 1200, 10), 777, 9000
    bnn_hip.set_math("bf16")
    net, sd = build_net(dev, lr, dims, "classification")
    x, y = synth.synth_batch("classification", B, dims[0], dims[2], seed=5)
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    bnn_hip.manual_seed(seed, counter=c)
    ev_c = engine.GraphedElbo(net, xd, yd, S, capture="calls")          # warm-up + the recording pass: [c, c + 2 S)
    assert ev_c.graph is None and len(ev_c.calls) >= 3
    bnn_hip.manual_seed(seed, counter=c + S)
    ev_g = engine.GraphedElbo(net, xd, yd, S)                           # warm-up: [c + S, c + 2 S)
    for r in range(3):
        sc = ev_c.replay().clone()
        sg = ev_g.replay().clone()
        torch.cuda.synchronize()
        assert torch.equal(sc, sg), (r, sc, sg)
        assert torch.equal(ev_c.logits, ev_g.logits) and all(torch.equal(ev_c.out[k], ev_g.out[k]) for k in ev_c.out)
        assert int(ev_c.counter.item()) == int(ev_g.counter.item()) == c + (3 + r) * S


@pytest.mark.parametrize("x3", [False, True])
def test_lr_prepare_many_equals_the_per_layer_launches(dev, x3):
    """bnn_lr_prepare_many (the prepared operands of several layers in one launch) against one bnn_lr_prepare[_x3] call per layer:
    fragments and KL workspace entries bit for bit -- the three layers of the MNIST network, an odd-shaped layer (partial tiles,
    K and N not multiples of the block) and a layer wide enough for several k-step ranges per group."""
    rs = np.random.RandomState(7)
    shapes = [(784, 1200), (1200, 1200), (1200, 10), (100, 72), (37, 5), (4096, 256)]
    layers = [[t(rs.uniform(-0.3, 0.3, sh).astype(np.float32)).to(dev), t(rs.uniform(-5, -4, sh).astype(np.float32)).to(dev),
               t(rs.uniform(-0.3, 0.3, sh[1]).astype(np.float32)).to(dev), t(rs.uniform(-5, -4, sh[1]).astype(np.float32)).to(dev)]
              for sh in shapes]
    ref = [ops.lr_prepare(*w, x3=x3) for w in layers]
    got = ops.lr_prepare_many([dict(w_mu=w[0], w_rho=w[1], b_mu=w[2], b_rho=w[3]) for w in layers], x3=x3)
    torch.cuda.synchronize()
    for sh, (rf, rw), (gf, gw) in zip(shapes, ref, got):
        assert torch.equal(rf.view(torch.int32), gf.view(torch.int32)), sh
        n = int(rw.view(torch.int32)[0])
        assert n == int(gw.view(torch.int32)[0]) and torch.equal(rw.view(-1, 4)[1:1 + n], gw.view(-1, 4)[1:1 + n]), sh
    with pytest.raises(ops.BnnHipError):
        ops.lr_prepare_many([dict(w_mu=w[0], w_rho=w[1], b_mu=w[2], b_rho=w[3]) for w in layers] * 2, x3=x3)      # > 8 layers


@pytest.mark.parametrize("lr", [False, True])
def test_graphed_predict_replays_predict_mc_with_fresh_epsilon(dev, lr):
    """engine.GraphedPredict (`net.predictor`): predict_mc as one captured evaluation -- every replay equals predict_mc at the
    same Philox sample indices (exact-fp32 math: to sum-order noise; bf16 math: the two chains take different kernel forms,
    rounding-level agreement), consecutive replays draw different epsilon, and a new minibatch copied into the static input
    changes the answer."""
    B, S, seed, c = 128, 10, 20260, 500
    net, sd = build_net(dev, lr, (784, 1200, 10), "classification")
    net.eval()
    xs = [torch.from_numpy(synth.synth_batch("classification", B, 784, 10, seed=40 + m)[0]).to(dev) for m in range(2)]
    for math, tol in (("f32", 2e-5), ("bf16", 6e-3)):
        bnn_hip.set_math(math)
        bnn_hip.manual_seed(seed, counter=c)
        p = net.predictor(xs[0].clone(), S)                   # its warm-up pass drew the samples [c, c + S)
        warm = p.probs.clone()
        if math == "bf16":                                    # the recorded-launch replay of the same prediction: the same bits
            bnn_hip.manual_seed(seed, counter=c - S)          # (its warm-up + recording pass: [c - S, c + S))
            pc = net.predictor(xs[0].clone(), S, capture="calls")
            bnn_hip.manual_seed(seed, counter=c + S)
            first = pc.replay()[1].clone()
        got = []
        for r in range(2):                                    # replay r: samples [c + (r + 1) S, c + (r + 2) S)
            preds, probs = p.replay()
            torch.cuda.synchronize()
            assert torch.equal(preds, probs.argmax(1))
            got.append(probs.clone())
        assert int(p.counter.item()) == c + 3 * S
        if math == "bf16":
            assert torch.equal(first, got[0])
        p.x.copy_(net._flat(xs[1]))
        other = p.replay()[1].clone()
        for k, (first, want_of) in enumerate(((c, warm), (c + S, got[0]), (c + 2 * S, got[1]))):
            bnn_hip.manual_seed(seed, counter=first)
            ref = net.predict_mc(xs[0], S)[1]
            err = float((ref - want_of).abs().max())
            assert err <= tol, (math, lr, k, err)
        assert float((got[0] - got[1]).abs().max()) > 1e-4     # fresh epsilon
        bnn_hip.manual_seed(seed, counter=c + 3 * S)
        assert float((net.predict_mc(xs[1], S)[1] - other).abs().max()) <= tol
        close(other.sum(1).cpu().numpy(), np.ones(B), rtol=1e-5)


@pytest.mark.parametrize("lr", [False, True])
def test_mc_predict_equals_the_reference_loop(dev, lr):
    """F3: predict_mc (samples batched per launch + bnn_mc_softmax_mean) against the loop of
    classification/class_task.py:81-87 run through the oracle on the same injected eps, and
    against the drop-in's own per-call loop."""
    bnn_hip.set_math("f32")
    B, S = 128, 6
    net, sd = build_net(dev, lr, (784, 1200, 10), "classification")
    net.eval()
    x, _ = synth.synth_batch("classification", B, 784, 10)
    p = O.NetParams.from_state_dict(sd, "classification", 784, lr, O.Prior.from_init([1.0], False))
    probs_ref = np.zeros((B, 10), dtype=np.float32)
    for s in range(S):
        eps = [t(a) for a in synth.synth_eps(p.eps_shapes(B), s)]
        out = O.network_forward(p, t(x).view(B, -1), eps)[0]
        probs_ref = probs_ref + (torch.softmax(out, dim=1) / S).numpy()
    install_eps(net, B, S, lr)
    preds, probs = net.predict_mc(t(x).to(dev), S)
    close(probs, probs_ref, rtol=1e-4, atol=1e-7)            # probabilities: logits carry fp32 sum-order noise
    assert np.array_equal(preds.cpu().numpy(), probs.cpu().numpy().argmax(1))
    install_eps(net, B, S, lr)
    loop = torch.zeros((B, 10), device=dev)
    with torch.no_grad():
        for _ in range(S):
            loop = loop + torch.softmax(net(t(x).to(dev), sample=True), dim=1) / S
    close(probs, loop.cpu().numpy(), rtol=1e-4, atol=1e-7)
    install_eps(net, B, S, lr)
    outs = net.forward_mc(t(x).to(dev), S)                    # reg_task.py:76-83 style collection
    assert tuple(outs.shape) == (S, B, 10)
    close(torch.softmax(outs, dim=2).mean(0), probs_ref, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("shape", [(1, 50, 128, 1), (4, 50, 128, 2), (8, 64, 128, 1), (64, 1200, 128, 1), (33, 7, 100, 3), (1, 1, 128, 1),
                                   (50, 1, 128, 1), (1200, 10, 128, 1), (784, 1200, 300, 1), (16, 16, 513, 1), (128, 600, 64, 2),
                                   (96, 1200, 16, 1), (200, 40, 129, 1)])
def test_launch_geometry_over_odd_shapes(dev, shape):
    """Tile / wave / block decomposition of K1 and K3 over shapes far from the benchmark's: with eps
    switched off y = x W^T + b_mu (BBB) or x M + b_mu (LR) exactly, so any output item a launch
    geometry fails to cover shows up (a short k range once left the LR block with fewer threads than
    output items)."""
    K, N, B, S = shape
    gen = torch.Generator(device="cpu").manual_seed(K * 7919 + N)
    x = torch.randn(B, K, generator=gen).to(dev)
    for lr in (False, True):
        for mm, tol in ((L.MATH_F32, 2e-5), (L.MATH_BF16, 2e-2)):
            wm = (torch.randn((K, N) if lr else (N, K), generator=gen) * 0.3).to(dev)
            wr = torch.full_like(wm, -3.0)
            bm = torch.randn(N, generator=gen).to(dev)
            br = torch.full((N,), -3.0, device=dev)
            sig_w, sig_b = torch.log1p(torch.exp(wr.double())), torch.log1p(torch.exp(br.double()))
            cnt = wm.numel() + bm.numel()
            c0 = -0.9189385332046727
            if lr:
                out = ops.lr_linear_fwd(x, wm, wr, bm, br, n_samples=S, sigma_p=0.9, math_mode=mm, relu=False,
                                        y_dtype=torch.float32, eps_mode=L.EPS_ZERO, want_kl=True, want_scalars=True)
                ref = x @ wm + bm
                sp = 0.9                                                   # closed-form KL (networks.py:109-114)
                kl = sum(0.5 * (2 * torch.log(torch.tensor(sp, dtype=torch.float64) / sg) - 1 + (sg / sp) ** 2 + (mu.double() / sp) ** 2).sum()
                         for mu, sg in ((wm, sig_w), (bm, sig_b)))
                close(out["kl3"][0], float(kl), rtol=2e-5)
            else:
                out = ops.bbb_linear_fwd(x, wm, wr, bm, br, n_samples=S, prior=ops.PriorSpec(False, 0.9), math_mode=mm,
                                         relu=False, y_dtype=torch.float32, eps_mode=L.EPS_ZERO, want_stats=True, want_scalars=True)
                ref = x @ wm.t() + bm
                # eps = 0: w = mu, so log q = count * c0 - sum log sigma and log p is the Gaussian at mu
                lq = cnt * c0 - float(torch.log(sig_w).sum() + torch.log(sig_b).sum())
                lp = cnt * (c0 - math.log(0.9)) - float((wm.double() ** 2).sum() + (bm.double() ** 2).sum()) / (2 * 0.81)
                close(out["log_q"], np.full(S, lq), rtol=2e-5)
                close(out["log_prior"], np.full(S, lp), rtol=2e-5)
            err = float((out["y"] - ref.unsqueeze(0)).abs().max())
            assert err <= tol * (float(ref.abs().max()) + 1e-6), (shape, "LR" if lr else "BBB", mm, err)


@pytest.mark.parametrize("shape", [(64, 100, 128, 12), (200, 1000, 100, 8), (784, 1200, 128, 24), (1200, 72, 130, 9), (8, 16, 16, 40),
                                   (96, 200, 257, 5)])
def test_throughput_forms_over_odd_shapes(dev, monkeypatch, shape):
    """The LDS-DMA block-GEMM forms (K1b, K3b with and without prepared fragments), forced on at
    shapes that are not multiples of their tiles: with eps off y must equal the plain product."""
    K, N, B, S = shape
    gen = torch.Generator(device="cpu").manual_seed(K * 31 + N)
    x = torch.randn(S, B, K, generator=gen).to(dev)
    x16 = x.to(torch.bfloat16)
    xf = x16.float()
    for lr in (False, True):
        wm = (torch.randn((K, N) if lr else (N, K), generator=gen) * 0.3).to(dev)
        wr = torch.full_like(wm, -3.0)
        bm = torch.randn(N, generator=gen).to(dev)
        br = torch.full((N,), -3.0, device=dev)
        if lr:
            ref = torch.matmul(xf, wm.to(torch.bfloat16).float()) + bm
            for prep in (False, True):
                wfrag, ws = ops.lr_prepare(wm, wr, bm, br) if prep else (None, None)
                out = ops.lr_linear_fwd(x16, wm, wr, bm, br, n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=False,
                                        y_dtype=torch.float32, eps_mode=L.EPS_ZERO, want_kl=True, x_sq=(x16 * x16),
                                        w_frag=wfrag, workspace=ws, form=L.FORM_GEMM)
                err = float((out["y"] - ref).abs().max())
                assert err <= 3e-3 * (float(ref.abs().max()) + 1e-6), (shape, "LR", prep, err)
        else:
            ref = torch.matmul(xf, wm.to(torch.bfloat16).float().t()) + bm
            for hoist in (False, True):
                out = ops.bbb_linear_fwd(x16, wm, wr, bm, br, n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16,
                                         relu=False, y_dtype=torch.float32, eps_mode=L.EPS_ZERO, want_stats=True,
                                         w_sigma=ops.softplus(wr) if hoist else None, form=L.FORM_GEMM)
                err = float((out["y"] - ref).abs().max())
                assert err <= 3e-3 * (float(ref.abs().max()) + 1e-6), (shape, "BBB", hoist, err)


@pytest.mark.parametrize("shape", [(1200, 1200, 128, 5), (784, 1200, 128, 7), (200, 1000, 100, 9), (256, 200, 257, 3), (72, 40, 300, 5),
                                   (1200, 1200, 128, 64)])
@pytest.mark.parametrize("prior", [ops.PriorSpec(False, 0.9), ops.PriorSpec(True, 1.0, 0.4, 1.1, 0.05)])
def test_pair_sharing_form_equals_the_register_form(dev, shape, prior):
    """K1b2 (block GEMM with the parameters shared through LDS by two (sample, batch block) pairs per block: taken with a
    hoisted sigma) against K1b (parameters in registers: taken without one) on the same on-chip epsilon: outputs bit for
    bit, per-sample statistics to fp32 summation order -- an ODD number of pairs (the last block's second pair slot is
    idle), ragged feature groups (fewer than four tiles in the last group), K % 32 != 0, several batch blocks per sample,
    Gaussian and mixture prior."""
    K, N, B, S = shape
    gen = torch.Generator(device="cpu").manual_seed(K * 13 + N + S)
    x16 = torch.rand(S, B, K, generator=gen).to(dev).to(torch.bfloat16)
    wm = ((torch.rand((N, K), generator=gen) - 0.5) * 0.4).to(dev)
    wr = (torch.rand((N, K), generator=gen) - 5.0).to(dev)
    bm = ((torch.rand(N, generator=gen) - 0.5) * 0.4).to(dev)
    br = (torch.rand(N, generator=gen) - 5.0).to(dev)
    kw = dict(n_samples=S, prior=prior, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.float32, eps_mode=L.EPS_PHILOX, seed=78,
              layer_id=1, sample_offset=11, want_stats=True, want_scalars=True, form=L.FORM_GEMM)
    sig = ops.softplus(wr)
    p_reg = ops.bbb_plan(x16, wm, wr, bm, br, **kw)
    p_lds = ops.bbb_plan(x16, wm, wr, bm, br, w_sigma=sig, **kw)
    assert p_reg["form"] == p_lds["form"] == L.FORM_GEMM and p_reg["waves"] == 4 and p_lds["waves"] == 8
    pairs = S * ((B + 127) // 128)
    assert p_lds["blocks"] == ((N + 63) // 64) * ((pairs + 1) // 2)
    a = ops.bbb_linear_fwd(x16, wm, wr, bm, br, **kw)
    b = ops.bbb_linear_fwd(x16, wm, wr, bm, br, w_sigma=sig, **kw)
    assert torch.equal(a["y"], b["y"])
    close(b["log_prior"], a["log_prior"].cpu().numpy(), rtol=2e-6)
    close(b["log_q"], a["log_q"].cpu().numpy(), rtol=2e-6)
    b2 = ops.bbb_linear_fwd(x16, wm, wr, bm, br, w_sigma=sig, **kw)            # and repeatable bit for bit
    assert torch.equal(b["y"], b2["y"]) and torch.equal(b["log_prior"], b2["log_prior"]) and torch.equal(b["log_q"], b2["log_q"])


@pytest.mark.parametrize("shape", [(1200, 1200, 128, 8), (784, 1200, 128, 5), (200, 1000, 100, 4), (256, 200, 257, 7), (1200, 72, 130, 9)])
def test_kslice_form_equals_the_whole_k_form(dev, shape):
    """K-sliced block GEMM with the fused last-arriver reduce (the form of 4 .. ~100 samples per launch): against the
    whole-K block GEMM on the same on-chip epsilon -- outputs to fp32 summation order (1e-5 of scale: the slices'
    partial sums are added in slice order instead of along one accumulator), per-sample statistics to 2e-6 -- ragged
    feature groups / batch blocks included; and bitwise repeatable whichever slice block arrives last."""
    K, N, B, S = shape
    gen = torch.Generator(device="cpu").manual_seed(K * 7 + N)
    x16 = torch.rand(S, B, K, generator=gen).to(dev).to(torch.bfloat16)
    wm = ((torch.rand((N, K), generator=gen) - 0.5) * 0.4).to(dev)
    wr = (torch.rand((N, K), generator=gen) - 5.0).to(dev)
    bm = ((torch.rand(N, generator=gen) - 0.5) * 0.4).to(dev)
    br = (torch.rand(N, generator=gen) - 5.0).to(dev)
    kw = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16, relu=True, y_dtype=torch.float32,
              eps_mode=L.EPS_PHILOX, seed=77, layer_id=2, sample_offset=9, want_stats=True, want_scalars=True)
    ref = ops.bbb_linear_fwd(x16, wm, wr, bm, br, form=L.FORM_GEMM, **kw)
    scratch = ops.split_scratch(S, B, N, dev)
    plan = ops.bbb_plan(x16, wm, wr, bm, br, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, **kw)
    assert plan["form"] == L.FORM_GEMM_KSLICE and plan["k_slices"] >= 2, plan
    outs = []
    for hoist in (False, True, True):
        out = ops.bbb_linear_fwd(x16, wm, wr, bm, br, form=L.FORM_GEMM_KSLICE, split_scratch=scratch,
                                 w_sigma=ops.softplus(wr) if hoist else None, **kw)
        scale = float(ref["y"].abs().max())
        assert float((out["y"] - ref["y"]).abs().max()) <= 1e-5 * scale, (shape, hoist)
        close(out["log_prior"], ref["log_prior"].cpu().numpy(), rtol=2e-6)
        close(out["log_q"], ref["log_q"].cpu().numpy(), rtol=2e-6)
        outs.append(out)
    assert torch.equal(outs[1]["y"], outs[2]["y"]) and torch.equal(outs[1]["log_q"], outs[2]["log_q"])
    n_zero = L.load().bnn_bbb_split_scratch_zero_bytes(S, B, N) // 4
    assert int(scratch[:n_zero].abs().sum()) == 0            # the arrival counters are left at zero


def _sq_of_rounded(v: torch.Tensor) -> torch.Tensor:
    """bf16(bf16(v)^2): the x^2 operand of the LR variance product in bf16 math, whoever forms it."""
    r = v.to(torch.bfloat16).float()
    return (r * r).to(torch.bfloat16)


def test_eval_prepare_equals_the_separate_passes(dev):
    """bnn_eval_prepare (softplus of several tensors + the input cast, one launch) against bnn_softplus / bnn_cast_bf16:
    bitwise, sizes that are not multiples of a block's 4096 elements or of the vector width, rho extremes included."""
    gen = torch.Generator(device="cpu").manual_seed(3)
    rhos = [(torch.rand(n, generator=gen) * 12 - 8).to(dev) for n in (1200 * 1200, 4097, 5, 784 * 1200)]
    rhos[2][:3] = torch.tensor([-110.0, 60.0, 89.0], device=dev)
    x = torch.rand(3, 50, 37, generator=gen).to(dev)
    sig, x16, xsq = ops.eval_prepare(rhos, cast=x, want_sq=True)
    for r, sg in zip(rhos, sig):
        assert torch.equal(sg, ops.softplus(r))
    r16, rsq = ops.cast_bf16(x, want_sq=True)
    assert torch.equal(x16, r16) and torch.equal(xsq, rsq)
    # the squares of bf16 math are ONE function of the rounded value (include/bnn_hip.h, bnn_math): bf16(bf16(x)^2)
    assert torch.equal(xsq, _sq_of_rounded(x))
    sig2, none16, _ = ops.eval_prepare(rhos[:1])
    assert none16 is None and torch.equal(sig2[0], sig[0])
    _, only16, nosq = ops.eval_prepare([], cast=x[:, :, :36].contiguous())
    assert nosq is None and torch.equal(only16, ops.cast_bf16(x[:, :, :36].contiguous()))


@pytest.mark.parametrize("shape", [(3, 128, 4096), (2, 37, 37), (20, 16, 100), (1, 128, 1), (4, 100, 1000), (2, 300, 1000)])
def test_regression_nll_wide_outputs(dev, shape):
    """K4's Gaussian NLL (networks.py:185-187) for output widths the 1-output regression net never
    reaches (the 4096-wide stack of BASELINE configs[4]): wave-per-row / 16-byte path and the
    block-per-sample grid against torch.distributions."""
    S, B, Cc = shape
    gen = torch.Generator(device="cpu").manual_seed(S * 131 + Cc)
    logits = torch.randn(S, B, Cc, generator=gen).to(dev)
    target = torch.randn(B, Cc, generator=gen).to(dev)
    sig = 0.7
    out = ops.elbo_finalize(workspaces=[], layer_in=[], layer_out=[], local_reparam=False, prior=ops.PriorSpec(), n_samples=S,
                            logits=logits, target=target, mode="regression", nll_sigma=sig)
    ref = torch.stack([-torch.distributions.Normal(logits[s].double(), sig).log_prob(target.double()).sum() for s in range(S)])
    close(out["nll"], ref.cpu().numpy(), rtol=2e-6)
    # with a scratch the rows' NLL is reduced by row blocks spread over the chip first (wide outputs only)
    out2 = ops.elbo_finalize(workspaces=[], layer_in=[], layer_out=[], local_reparam=False, prior=ops.PriorSpec(), n_samples=S,
                             logits=logits, target=target, mode="regression", nll_sigma=sig, scratch=ops.final_scratch(S, dev))
    close(out2["nll"], ref.cpu().numpy(), rtol=2e-6)
    if Cc >= 37:
        labels = torch.randint(0, Cc, (B,), generator=gen).to(dev)
        refc = torch.stack([torch.nn.functional.cross_entropy(logits[s].double(), labels, reduction="sum") for s in range(S)])
        for scratch in (None, ops.final_scratch(S, dev)):
            oc = ops.elbo_finalize(workspaces=[], layer_in=[], layer_out=[], local_reparam=False, prior=ops.PriorSpec(), n_samples=S,
                                   logits=logits, target=labels, mode="classification", scratch=scratch)
            close(oc["nll"], refc.cpu().numpy(), rtol=2e-6)


def test_misaligned_operands_are_rejected_or_handled(dev):
    """A tensor view that starts 4 bytes into an allocation breaks the 16-byte accesses of the vector
    paths: every entry point must either take its scalar path and be right, or answer with
    BNN_ERR_ALIGN (a Python exception) — never read the wrong bytes or fault."""
    from bnn_hip import BnnHipError
    K, N, B = 64, 32, 16
    x_bad = torch.randn(B * K + 1, device=dev)[1:].view(B, K)      # 4-byte offset
    assert x_bad.data_ptr() % 16 != 0
    wm, wr = torch.randn(N, K, device=dev) * 0.1, torch.full((N, K), -3.0, device=dev)
    bm, br = torch.randn(N, device=dev), torch.full((N,), -3.0, device=dev)

    def right_or_refused(fn, ref):
        try:
            out = fn()
        except BnnHipError as e:
            assert "align" in str(e).lower()
            return "refused"
        close(out, ref.cpu().numpy(), rtol=2e-5, atol=1e-5)
        return "handled"

    seen = set()
    seen.add(right_or_refused(lambda: ops.bbb_linear_fwd(x_bad, wm, wr, bm, br, n_samples=1, prior=ops.PriorSpec(False, 1.0),
                                                         math_mode=L.MATH_F32, relu=False, y_dtype=torch.float32,
                                                         eps_mode=L.EPS_ZERO, want_stats=False)["y"][0], x_bad @ wm.t() + bm))
    wml, wrl = wm.t().contiguous(), wr.t().contiguous()
    seen.add(right_or_refused(lambda: ops.lr_linear_fwd(x_bad, wml, wrl, bm, br, n_samples=1, sigma_p=1.0, math_mode=L.MATH_F32,
                                                        relu=False, y_dtype=torch.float32, eps_mode=L.EPS_ZERO,
                                                        want_kl=False)["y"][0], x_bad @ wml + bm))
    w_bad = (torch.randn(N * K + 1, device=dev) * 0.1)[1:].view(N, K)
    seen.add(right_or_refused(lambda: ops.bbb_linear_fwd(x_bad.contiguous().clone(), w_bad, wr, bm, br, n_samples=1,
                                                         prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_F32, relu=False,
                                                         y_dtype=torch.float32, eps_mode=L.EPS_ZERO, want_stats=False)["y"][0],
                              x_bad @ w_bad.t() + bm))
    x3, gy = torch.randn(1, B, K, device=dev), torch.randn(1, B, N, device=dev)
    with pytest.raises(BnnHipError):                              # the backward kernels are vector-only on their parameters
        ops.bbb_linear_bwd(x3, gy, None, w_bad, wr, bm, br, n_samples=1, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_F32,
                           relu=False, eps_mode=L.EPS_PHILOX, seed=1, layer_id=0, want_gx=False)
    assert seen <= {"refused", "handled"} and seen


@pytest.mark.parametrize("prior", [ops.PriorSpec(False, 0.9), ops.PriorSpec(True, 1.0, 0.4, 1.1, 0.05)])
@pytest.mark.parametrize("shape", [(1, 128, 784, 1200), (1, 128, 1200, 1200), (3, 20, 72, 38), (2, 7, 8, 5), (1, 130, 64, 17)])
def test_split_sampling_then_matmul_equals_the_fused_layer(dev, prior, shape):
    """K1s + matmul-only K1 against fused K1a on the same Philox elements: the sampled bf16 weights are the ones
    K1a feeds its matrix core (recomputed here from K1a's eps dump), the outputs agree to fp32 summation order,
    the statistics to fp32 summation order, the sampled biases exactly."""
    S, B, K, N = shape
    rs = np.random.RandomState(21)
    mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
    x = mk(B, K, lo=-1, hi=1).to(torch.bfloat16)
    w_mu, w_rho, b_mu, b_rho = mk(N, K), mk(N, K, lo=-5, hi=-2), mk(N), mk(N, lo=-5, hi=-2)
    fused = ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, n_samples=S, prior=prior, math_mode=L.MATH_BF16, relu=True,
                               y_dtype=torch.float32, eps_mode=L.EPS_PHILOX, seed=77, layer_id=3, sample_offset=9,
                               want_stats=True, dump_eps=True)
    sm = ops.bbb_sample_weights([dict(w_mu=w_mu, w_rho=w_rho, b_mu=b_mu, b_rho=b_rho, prior=prior, layer_id=3)],
                                n_samples=S, seed=77, sample_offset=9)[0]
    y = ops.bbb_sampled_matmul(x, sm["w"], sm["b"], n_samples=S, relu=True, y_dtype=torch.float32)
    # sampled parameters: same eps elements, same arithmetic up to the last ulp of softplus
    sig = torch.log1p(torch.exp(w_rho.double()))
    w_ref = (w_mu.double() + sig * fused["eps_w"].double())
    err = (sm["w"].double() - w_ref).abs()
    assert float((err / (w_ref.abs() + 1e-3)).max()) <= 2.0 ** -8              # bf16 rounding of the fp32 value
    b_ref = b_mu.double() + torch.log1p(torch.exp(b_rho.double())) * fused["eps_b"].double()
    close(sm["b"], b_ref.cpu().numpy(), rtol=2e-6, atol=1e-7)
    # outputs: both feed bf16(w) and bf16 x to the matrix core, fp32 accumulate
    scale = float(fused["y"].abs().max()) + 1e-6
    assert float((y - fused["y"]).abs().max()) <= 2e-5 * scale
    # statistics per sample
    def sums(ws):
        T = int(ws[:1].view(torch.int32)[0])
        return ws[4:4 + 4 * S * T].view(S, T, 4).double().sum(1), ws[4:4 + 4 * T].view(T, 4).double().sum(0)[2]
    (fa, fls), (sa, sls) = sums(fused["workspace"]), sums(sm["workspace"])
    close(sa[:, :2], fa[:, :2].cpu().numpy(), rtol=2e-5, atol=1e-3)
    close(sls, float(fls), rtol=2e-5)


@pytest.mark.parametrize("prior", [ops.PriorSpec(False, 0.9), ops.PriorSpec(True, 1.0, 0.4, 1.1, 0.05)])
@pytest.mark.parametrize("S", [2, 4, 5, 7, 9])
def test_sampling_in_sample_groups_equals_one_sample_launches(dev, prior, S):
    """K1s serves a group of four MC samples from one read of (mu, rho) and one softplus (a last group of 1..3 where S % 4 != 0):
    every sample's bf16 weights, fp32 biases and statistics entries are bit for bit those of its own one-sample launch at
    its own Philox offset -- two layers in one launch (odd shapes: a chunk that ends inside the matrix, biases spread over
    the chunks)."""
    rs = np.random.RandomState(100 + S)
    mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
    shapes = [(200, 136), (24, 1000)]                           # (out, in): in % 8 == 0
    layers = [dict(w_mu=mk(n, k), w_rho=mk(n, k, lo=-5, hi=-2), b_mu=mk(n), b_rho=mk(n, lo=-5, hi=-2), prior=prior, layer_id=1 + i)
              for i, (n, k) in enumerate(shapes)]
    grp = ops.bbb_sample_weights([dict(l) for l in layers], n_samples=S, seed=5, sample_offset=40)
    for s_ in range(S):
        one = ops.bbb_sample_weights([dict(l) for l in layers], n_samples=1, seed=5, sample_offset=40 + s_)
        for li, (n, k) in enumerate(shapes):
            assert torch.equal(grp[li]["w"][s_].view(torch.int16), one[li]["w"][0].view(torch.int16)), (S, s_, li)
            assert torch.equal(grp[li]["b"][s_], one[li]["b"][0]), (S, s_, li)
            T = int(grp[li]["workspace"][:1].view(torch.int32)[0])
            assert T == int(one[li]["workspace"][:1].view(torch.int32)[0])
            g = grp[li]["workspace"][4:4 + 4 * S * T].view(S, T, 4)[s_]
            o = one[li]["workspace"][4:4 + 4 * T].view(T, 4)
            # {sum eps^2, sum w^2 | sum log p_mix} per chunk; sum log sigma belongs to global sample index 0 of a launch only
            assert torch.equal(g[:, :2], o[:, :2]), (S, s_, li)
            if s_ == 0:
                assert torch.equal(g[:, 2], o[:, 2])
            else:
                assert float(g[:, 2].abs().max()) == 0.0


def _oracle_pairs(p, xs, ys, seed, base, S, pairs=None, bf16=False):
    """Oracle scalars of (minibatch m, MC sample j) pairs on the eps the device generator draws for global sample index
    base + m * S + j: rows of (log p | KL, log q | 0, nll), plus the logits.  `pairs`: flat indices m * S + j (default
    all).  `bf16`: additionally the same rows with the device's bf16 rounding points (O.network_forward_bf16), on the
    same epsilon: returns (rows, logits, rows16, logits16)."""
    rows, logits, rows16, logits16 = [], [], [], []
    torch.set_num_threads(8)
    for f in (range(len(xs) * S) if pairs is None else pairs):
        m, j = divmod(int(f), S)
        eps = O.philox_eps_for_network(p, xs[m].shape[0], seed, base + m * S + j)
        out, a, b = O.network_forward(p, t(xs[m]), eps)
        rows.append([float(a), float(b) if b is not None else 0.0, float(O.nll(out, t(ys[m]), p.mode))])
        logits.append(out.numpy())
        if bf16:
            out, a, b = O.network_forward_bf16(p, t(xs[m]), eps)
            rows16.append([float(a), float(b) if b is not None else 0.0, float(O.nll(out, t(ys[m]), p.mode))])
            logits16.append(out.numpy())
    torch.set_num_threads(1)
    if bf16:
        return np.asarray(rows, np.float64), np.stack(logits), np.asarray(rows16, np.float64), np.stack(logits16)
    return np.asarray(rows, np.float64), np.stack(logits)


# beta of the reference's schedule (class_task.py:70: beta halves every minibatch, 0.5 at idx 0, exactly 0 in fp32 from idx 149
# of 468): the first minibatch, one where the complexity term (~7.4e6 * beta) is of the NLL's size, and the tail
BETAS = (0.5, 2.0 ** -10, 0.0)
# Pinned tolerance of the bf16 path against the oracle WITH the device's rounding points (O.network_forward_bf16): only
# fp32 accumulation order (and the rare operand that rounds the other way after an fma / softplus ulp) differs.  This is
# the regression guard of the bf16 kernels: set from the arithmetic, not from a measurement of the kernels.
BF16_POINTS_NLL_RTOL = 1e-4
# logits: a hidden activation whose fp32 sum lands within ~1e-6 of a bf16 rounding boundary rounds the other way on the
# device (different summation order): ~1e-3 of the 128 x 1200 activations of a layer, each a 2^-8 relative step that fans
# out through the next layer's 1200 weights -- a few 1e-4 of the logit scale on the rows it hits, invisible in the NLL
BF16_POINTS_LOGIT_TOL = 2e-3          # of the logit scale


def _elbo(rows, S, beta, lr):
    """networks.py:205-208 / :222-224 from per-evaluation sums [.., (sum log p | KL, sum log q, sum nll)]"""
    b32 = float(np.float32(beta))                      # the reference multiplies fp32 tensors by beta
    if lr:
        return b32 * rows[..., 0] / S + rows[..., 2] / S
    return b32 * rows[..., 1] / S - b32 * rows[..., 0] / S + rows[..., 2] / S


# (G, S, pairs checked against the oracle): the driver's old 20-minibatch group (K-sliced form), the default 256-minibatch
# launch group (block-GEMM form: a strided subset of its pairs, ~2 s of oracle each), 64 MC samples of one minibatch
# (1, 4) and (1, 7): the LR hidden layer's forms between the latency and the throughput regime (K3s in two blocks per CU; K3b
# over prepared fragments from 7 samples)
TIMED_SHAPES = [(1, 1, None), (1, 4, None), (1, 7, None), (1, 8, None), (4, 2, None), (20, 1, None),
                (256, 1, (0, 1, 17, 63, 64, 127, 128, 200, 254, 255)), (1, 64, (0, 1, 7, 8, 31, 32, 62, 63))]


@pytest.mark.parametrize("G,S,pairs", TIMED_SHAPES)
@pytest.mark.parametrize("variant", ["bbb", "lr"])
def test_timed_path_against_oracle_on_philox_eps(dev, variant, G, S, pairs):
    """The path bench.py times -- engine.GraphedElbo, captured hipGraph, bf16 math, on-chip Philox, several
    evaluations per replay, G stacked minibatches x S MC samples per launch group, at the shapes that are timed --
    directly against the oracle on the same epsilon (oracle.philox_eps_for_network for the matching global sample
    indices), per (minibatch, sample) pair:
      * log p / log q / KL (fp32 statistics of un-rounded weights): rtol 1e-5 against the fp32 oracle;
      * NLL and logits against the oracle with the device's bf16 rounding points: rtol 1e-4 / 2e-3 of scale (pinned from
        the arithmetic: accumulation order and the boundary flips it causes), and with it the ELBO at EVERY beta of the schedule, beta = 0 (pure
        NLL) included, to the north star's rtol 1e-4;
      * against the fp32 oracle (= the reference's arithmetic): the bf16 operand rounding moves the NLL by up to
        BF16_NLL_RTOL, so the ELBO agrees to rtol 1e-4 while beta * complexity >> NLL (beta = 0.5: the first minibatches
        of an epoch) and to BF16_NLL_RTOL * NLL / (beta * complexity + NLL) in general -- the stated tolerance of the
        bf16 mode as a function of beta; the exact-fp32 math mode meets 1e-4 at every beta
        (test_c2_mnist_shape_elbo, test_elbo_over_the_beta_schedule_fp32_math)."""
    from bnn_hip import engine
    lr = variant == "lr"
    B, dims, seed, first, E = 128, (784, 1200, 10), 424242, 7000, 2
    bnn_hip.set_math("bf16")
    net, sd = build_net(dev, lr, dims, "classification")
    p = O.NetParams.from_state_dict(sd, "classification", dims[0], lr, O.Prior.from_init([1.0], False))
    xs, ys = zip(*[synth.synth_batch("classification", B, dims[0], dims[2], seed=100 + m) for m in range(G)])
    xd = torch.from_numpy(np.stack(xs)).to(dev)
    yd = torch.from_numpy(np.stack(ys)).to(dev)
    bnn_hip.manual_seed(seed, counter=first)
    ev = engine.GraphedElbo(net, xd if G > 1 else xd[0], yd if G > 1 else yd[0], S, stacked=G > 1, evals_per_replay=E)
    sums = ev.replay().clone().view(G, 4).double().cpu().numpy()
    torch.cuda.synchronize()
    total = G * S
    assert int(ev.counter.item()) == first + total * (1 + E)       # warm-up + E evaluations
    base = first + total * E                                        # the LAST evaluation of the replay
    idx = np.arange(total) if pairs is None else np.asarray(pairs)
    # (LR: ONE oracle variant -- the x^2 operand of the variance product is bf16(bf16(x)^2) in every kernel form, so the launch
    # plans -- carried squares for K3b, fragments squared in place by K3a / K3s / K3r -- cannot change a bit of the result)
    want, want_logits, w16, w16_logits = _oracle_pairs(p, xs, ys, seed, base, S, pairs=idx, bf16=True)
    keys = ("kl",) if lr else ("log_prior", "log_q")
    for c, k in enumerate(keys):
        close(ev.out[k].double().cpu().numpy()[idx], want[:, c], rtol=1e-5)
        close(w16[:, c], want[:, c], rtol=1e-6)                    # the statistics never see a rounded operand
    got_nll = ev.out["nll"].double().cpu().numpy()[idx]
    lg = ev.logits.double().cpu().numpy().reshape(total, B, dims[2])[idx]
    # (i) pinned: against the oracle with the device's rounding points
    nll16_err = np.abs(got_nll - w16[:, 2]) / np.abs(w16[:, 2])
    lg16_err = np.abs(lg - w16_logits).max() / np.abs(w16_logits).max()
    # (ii) the precision choice itself: against the fp32 oracle
    nll_err = np.abs(got_nll - want[:, 2]) / np.abs(want[:, 2])
    lg_err = np.abs(lg - want_logits).max() / np.abs(want_logits).max()
    print(f"\n[bf16 at C2] {variant} G={G} S={S}: vs rounding-point oracle nll {nll16_err.max():.2e} logits {lg16_err:.2e}; "
          f"vs fp32 oracle nll {nll_err.max():.2e} logits {lg_err:.2e} of scale")
    assert nll16_err.max() <= BF16_POINTS_NLL_RTOL and lg16_err <= BF16_POINTS_LOGIT_TOL
    assert nll_err.max() <= BF16_NLL_RTOL and lg_err <= BF16_LOGIT_TOL
    assert (sums[:, 3] == S).all()
    # per-evaluation sums and the ELBO over the beta schedule (evaluations all of whose pairs were checked)
    full_eval = [m for m in range(G) if all((m * S + j) in set(idx.tolist()) for j in range(S))]
    pos = {int(f): i for i, f in enumerate(idx.tolist())}
    for m in full_eval:
        sel = [pos[m * S + j] for j in range(S)]
        e32, e16 = want[sel].sum(0), w16[sel].sum(0)
        close(sums[m, 0], e32[0], rtol=1e-5)
        if not lr:
            close(sums[m, 1], e32[1], rtol=1e-5)
        close(sums[m, 2], e16[2], rtol=BF16_POINTS_NLL_RTOL)
        for beta in BETAS:
            got = _elbo(sums[m, :3], S, beta, lr)
            close(got, _elbo(e16, S, beta, lr), rtol=1e-4)          # every beta, beta = 0 included
            ref = _elbo(e32, S, beta, lr)
            cplx = abs(ref - e32[2] / S)                            # beta * complexity
            close(got, ref, rtol=max(1e-4 if beta == 0.5 else 0.0, BF16_NLL_RTOL * (e32[2] / S) / (cplx + e32[2] / S)))


@pytest.mark.parametrize("variant", ["bbb", "lr"])
def test_elbo_over_the_beta_schedule_fp32_math(dev, variant):
    """The exact-fp32 math mode (v_mfma_f32_16x16x4_f32) meets the north star's ELBO rtol 1e-4 at EVERY beta of the
    reference's schedule (class_task.py:70), the pure-NLL tail (beta = 0) included: C2 network, 2 minibatches x 2 MC
    samples, on-chip Philox, against the fp32 oracle on the same epsilon."""
    from bnn_hip import engine
    lr = variant == "lr"
    G, S, B, dims, seed, first = 2, 2, 128, (784, 1200, 10), 515151, 300
    bnn_hip.set_math("f32")
    net, sd = build_net(dev, lr, dims, "classification")
    p = O.NetParams.from_state_dict(sd, "classification", dims[0], lr, O.Prior.from_init([1.0], False))
    xs, ys = zip(*[synth.synth_batch("classification", B, dims[0], dims[2], seed=300 + m) for m in range(G)])
    bnn_hip.manual_seed(seed, counter=first)
    got = net.elbo_many(torch.from_numpy(np.stack(xs)).to(dev), torch.from_numpy(np.stack(ys)).to(dev), S).double().cpu().numpy()
    want, _ = _oracle_pairs(p, xs, ys, seed, first, S)
    w = want.reshape(G, S, 3).sum(1)
    close(got[:, 2], w[:, 2], rtol=2e-5)                            # NLL in fp32 math
    for beta in BETAS:
        close(_elbo(got[:, :3], S, beta, lr), _elbo(w, S, beta, lr), rtol=1e-4)


@pytest.mark.parametrize("lr", [False, True])
def test_stacked_minibatches_equal_separate_evaluations(dev, lr):
    """G minibatches x S samples in one launch per layer (BayesianNetwork.elbo_many, the product API bench.py times)
    against G separate sample_elbo evaluations of the same minibatches on the same global sample indices, fp32 math:
    identical epsilon, so the 4-vectors agree to fp32 summation order."""
    from bnn_hip import engine
    G, S, B, dims = 5, 3, 64, (784, 1200, 10)
    net, _ = build_net(dev, lr, dims, "classification", B=B)
    xs, ys = zip(*[synth.synth_batch("classification", B, dims[0], dims[2], seed=900 + m) for m in range(G)])
    xd, yd = torch.from_numpy(np.stack(xs)).to(dev), torch.from_numpy(np.stack(ys)).to(dev)
    bnn_hip.manual_seed(31, counter=500)
    got = net.elbo_many(xd, yd, S).double().cpu().numpy()
    assert got.shape == (G, 4) and bnn_hip.runtime.state.counter == 500 + G * S
    for m in range(G):
        bnn_hip.manual_seed(31, counter=500 + m * S)
        with torch.no_grad():
            r = (net.sample_elbo_lr if lr else net.sample_elbo)(xd[m], yd[m], 0.5, S)
        if lr:
            close(got[m, 0] / S, float(r[1]), rtol=2e-6)
            close(got[m, 2] / S, float(r[2]), rtol=5e-6)
        else:
            close(got[m, 0] / S, float(r[1]), rtol=2e-6)
            close(got[m, 1] / S, float(r[2]), rtol=2e-6)
            close(got[m, 2] / S, float(r[3]), rtol=5e-6)
        assert got[m, 3] == S


# One 4096 x 4096 layer in bf16 math against the CPU restatement with the same rounding points (O.bbb_linear_bf16): what is
# left is fp32 summation order over K = 4096 (~1e-6 of scale) and the sampled weights whose bf16 rounding flips because the
# device's epsilon differs from the numpy restatement's by an ulp of v_log / v_sin (a relative 1e-6 of w against the 2^-8
# rounding grid: ~1e-3 of the weights, each a 2^-8 step of one product: a few 1e-5 of the output scale on the rows it
# hits).  Set from that arithmetic; the measured values are printed.
C5_POINTS_TOL = 1.5e-4      # measured 4.3e-5 (K1s + K1g, batch 1024 and 4096)


@pytest.mark.parametrize("form", ["tile", "gemm", "gemm_kslice"])
def test_c5_wide_bbb_layer_against_oracle(dev, form):
    """BASELINE configs[4] at full size: one 4096 x 4096 BayesianLinear layer, batch 128, 4 MC samples (the per-GPU
    share of C5's 32), on-chip Philox, through each kernel form, against the oracle's layer on the same epsilon.
    fp32 math (tile form): y to 2e-5 of scale; bf16 operands: y to C5_POINTS_TOL of scale against the oracle's layer WITH the
    device's rounding points (O.bbb_linear_bf16: only fp32 summation order and the handful of sampled weights that round
    the other way after an ulp of the generator's arithmetic differ) -- the pin of the kernels -- and to 2e-2 of scale
    against the reference's fp32 arithmetic (K = 4096 products of bf16-rounded operands: the precision choice, not the
    kernels); the fp32 statistics to 1e-5 in every form."""
    S, B, K, N, seed, off = 4, 128, 4096, 4096, 99, 40
    rs = np.random.RandomState(77)
    w_mu = rs.uniform(-0.2, 0.2, (N, K)).astype(np.float32)
    w_rho = rs.uniform(-5, -4, (N, K)).astype(np.float32)
    b_mu = rs.uniform(-0.2, 0.2, N).astype(np.float32)
    b_rho = rs.uniform(-5, -4, N).astype(np.float32)
    x = rs.uniform(0, 1, (B, K)).astype(np.float32)
    prior = O.Prior.from_init([1.0], False)
    dw = [t(a).to(dev) for a in (w_mu, w_rho, b_mu, b_rho)]
    math_mode = L.MATH_F32 if form == "tile_f32" else L.MATH_BF16
    runs = [("bf16", L.MATH_BF16, {"tile": L.FORM_TILE, "gemm": L.FORM_GEMM, "gemm_kslice": L.FORM_GEMM_KSLICE}[form])]
    if form == "tile":
        runs.append(("f32", L.MATH_F32, L.FORM_TILE))
    ref = []
    torch.set_num_threads(8)
    for s in range(S):
        ew = t(O.philox_normal(seed, O.tensor_id(1, 0), off + s, N, K))
        eb = t(O.philox_normal(seed, O.tensor_id(1, 1), off + s, 1, N))[0]
        y, lp, lq = O.bbb_linear(t(x), t(w_mu), t(w_rho), t(b_mu), t(b_rho), ew, eb, prior)
        y16, _, _ = O.bbb_linear_bf16(t(x), t(w_mu), t(w_rho), t(b_mu), t(b_rho), ew, eb, prior)
        ref.append((torch.relu(y).numpy(), float(lp), float(lq), torch.relu(y16).numpy()))
    torch.set_num_threads(1)
    for name, mm, fm in runs:
        xin = t(x).to(dev) if mm == L.MATH_F32 else t(x).to(dev).to(torch.bfloat16)
        kw = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=mm, relu=True, y_dtype=torch.float32,
                  eps_mode=L.EPS_PHILOX, seed=seed, layer_id=1, sample_offset=off, want_stats=True, want_scalars=True, form=fm,
                  split_scratch=ops.split_scratch(S, B, N, dev) if fm == L.FORM_GEMM_KSLICE else None)
        plan = ops.bbb_plan(xin, *dw, **kw)
        assert plan["form"] == fm, plan
        out = ops.bbb_linear_fwd(xin, *dw, **kw)
        for s in range(S):
            scale = float(np.abs(ref[s][0]).max())
            got = out["y"][s].double().cpu().numpy()
            err = float(np.abs(got - ref[s][0]).max())
            assert err <= (2e-5 if mm == L.MATH_F32 else 2e-2) * scale, (name, form, s, err, scale)
            if mm == L.MATH_BF16:
                err16 = float(np.abs(got - ref[s][3]).max())
                assert err16 <= C5_POINTS_TOL * scale, (name, form, s, err16 / scale)
            close(out["log_prior"][s], ref[s][1], rtol=1e-5)
            close(out["log_q"][s], ref[s][2], rtol=1e-5)


@pytest.mark.parametrize("shape", [(1, 128, 1200, 1200), (1, 128, 784, 1200), (2, 128, 1200, 1200), (3, 100, 264, 72),
                                   (1, 20, 1000, 1200), (1, 128, 64, 4096), (10, 128, 784, 1200), (23, 100, 264, 136), (64, 128, 784, 1200)])
def test_lr_k_sliced_form_against_oracle_and_tile_form(dev, shape):
    """K3s (lr_fwd_kslice_kernel, BNN_FORM_GEMM_KSLICE): 32-feature groups x K slices meeting through a scratch, for 1-2
    samples on a wide [in,out] layer (more where the layer is narrower) -- and for 2 .. 64 samples on ONE input (the first
    layer of sample_elbo_lr / predict, networks.py:211-225: the reference runs forward(x) per sample on the same x), where a
    unit's two products are made once and its epilogue runs per sample (the S > 1 shapes here: x is [batch, in]).
    y, y^2, the saved variance and the backward factor against the oracle
    (networks.py:116-138) and against K3a on the same inputs; the KL sums; twice the same bits (the slices are added in
    slice order whoever arrives last); ragged batch, K tail inside a k-step, a slice count that does not divide the
    k-steps, N % 32 != 0."""
    S, B, K, N = shape
    seed, off = 9, 5
    rs = np.random.RandomState(sum(shape))
    w_mu = rs.uniform(-0.2, 0.2, (K, N)).astype(np.float32)
    w_rho = rs.uniform(-5, -4, (K, N)).astype(np.float32)
    b_mu = rs.uniform(-0.2, 0.2, N).astype(np.float32)
    b_rho = rs.uniform(-5, -4, N).astype(np.float32)
    x = rs.uniform(0, 1, (B, K)).astype(np.float32)
    dw = [t(a).to(dev) for a in (w_mu, w_rho, b_mu, b_rho)]
    x16 = t(x).to(dev).to(torch.bfloat16)
    xr = x16.float().cpu()                                     # the oracle sees the bf16-rounded input
    ref = []
    torch.set_num_threads(8)
    for s in range(S):
        ea = t(O.philox_normal(seed, O.tensor_id(2, 2), off + s, B, N))
        eb = t(O.philox_normal(seed, O.tensor_id(2, 1), off + s, 1, N))[0]
        y, kw_, kb_ = O.lr_linear(xr, t(w_mu), t(w_rho), t(b_mu), t(b_rho), ea, eb, 1.0)
        ref.append((torch.relu(y).numpy(), float(kw_ + kb_)))
    torch.set_num_threads(1)
    kw = dict(n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.float32, eps_mode=L.EPS_PHILOX, seed=seed,
              layer_id=2, sample_offset=off, want_kl=True, want_scalars=True, want_v=True, want_hfac=True, want_y16=True)
    SU = 1                                                      # samples a unit counts: they all read the one x
    scratch = ops.lr_split_scratch(SU, B, N, dev)
    plan = ops.lr_plan(x16, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, **kw)
    assert plan["form"] == L.FORM_GEMM_KSLICE and plan["features_per_block"] == 32, plan
    assert plan["blocks"] == ((N + 31) // 32) * SU * ((B + 127) // 128) * plan["k_slices"], plan
    if S > 1:                                                   # without being asked for, ahead of the per-sample forms
        assert ops.lr_plan(x16, *dw, form=L.FORM_AUTO, split_scratch=scratch, **kw)["form"] == L.FORM_GEMM_KSLICE
    sq = lambda: torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)
    a = ops.lr_linear_fwd(x16, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, out_sq=sq(), **kw)
    a2 = ops.lr_linear_fwd(x16, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, out_sq=sq(), **kw)
    b = ops.lr_linear_fwd(x16, *dw, form=L.FORM_TILE, out_sq=sq(), **kw)
    # the fp32-input variant (first layer of an evaluation: no cast launch) rounds x where the cast would have: on an
    # input that is already bf16-representable it gives the same bits
    a3 = ops.lr_linear_fwd(x16.float(), *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, out_sq=sq(), **kw)
    for key in ("y", "y_sq", "v", "hfac", "y16", "kl3"):
        assert torch.equal(a[key], a3[key]), key
    # bnn_lr_rider: the NEXT (narrow) layer's operands prepared beside this launch -- as extra blocks of the K-sliced form, or
    # by a bnn_lr_prepare launch ahead of any other form: bitwise the fragments of a stand-alone bnn_lr_prepare, its KL
    # sums, and this layer's own results unchanged
    w3 = [t(rs.uniform(-0.3, 0.3, (N, 10)).astype(np.float32)).to(dev), t(rs.uniform(-5, -4, (N, 10)).astype(np.float32)).to(dev),
          t(rs.uniform(-0.3, 0.3, 10).astype(np.float32)).to(dev), t(rs.uniform(-5, -4, 10).astype(np.float32)).to(dev)]
    ref_frag, ref_ws = ops.lr_prepare(*w3)
    ws_sum = lambda w: w.view(-1, 4)[1:1 + int(w.view(torch.int32)[0])].double().sum(0)[:3]
    for fm in (L.FORM_GEMM_KSLICE, L.FORM_TILE):
        frag = torch.zeros_like(ref_frag)
        wsr = ops.lr_workspace(10, dev)
        a4 = ops.lr_linear_fwd(x16, *dw, form=fm, split_scratch=scratch if fm == L.FORM_GEMM_KSLICE else None, out_sq=sq(),
                               rider=dict(w_mu=w3[0], w_rho=w3[1], b_mu=w3[2], b_rho=w3[3], w_frag=frag, workspace=wsr), **kw)
        assert torch.equal(frag.view(torch.int32), ref_frag.view(torch.int32)), fm
        assert float((ws_sum(wsr) - ws_sum(ref_ws)).abs().max()) <= 1e-5 * float(ws_sum(ref_ws).abs().max()), fm
        assert torch.equal(a4["y"], (a if fm == L.FORM_GEMM_KSLICE else b)["y"]), fm
    zero = L.load().bnn_lr_split_scratch_zero_bytes(SU, B, N) // 4
    assert int(scratch[:zero].abs().sum()) == 0                # the arrival counters are left at zero
    for key in ("y", "y_sq", "v", "hfac", "y16", "kl3"):
        assert torch.equal(a[key], a2[key]), key               # reproducible bit for bit
    if S > 1:
        # the shared products: every sample's saved variance is the same tile, and each sample alone (its own launch at its own
        # Philox offset, one unit per sample) gives the same bits
        assert all(torch.equal(a["v"][0], a["v"][s]) for s in range(1, S))
        for s in (0, S - 1):
            one = ops.lr_linear_fwd(x16, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, out_sq=torch.empty((1, B, N), dtype=torch.bfloat16, device=dev),
                                    **dict(kw, n_samples=1, sample_offset=off + s))
            assert torch.equal(one["y"][0], a["y"][s]) and torch.equal(one["hfac"][0], a["hfac"][s]), s
    close(a["kl3"][0], ref[0][1], rtol=1e-5)
    close(a["kl3"][0], float(b["kl3"][0]), rtol=1e-6)
    for s in range(S):
        scale = float(np.abs(ref[s][0]).max())
        err = float(np.abs(a["y"][s].double().cpu().numpy() - ref[s][0]).max())
        assert err <= 2e-2 * scale, (s, err, scale)
        # against K3a: the same operands rounded to bf16 at the same points, another summation order
        assert float((a["y"][s] - b["y"][s]).abs().max()) <= 2e-3 * scale
        assert float((a["v"][s] - b["v"][s]).abs().max()) <= 1e-3 * float(b["v"][s].abs().max())
        assert float((a["y16"][s].float() - a["y"][s]).abs().max()) <= 4e-3 * scale
        assert float((a["y_sq"][s].float() - a["y"][s] ** 2).abs().max()) <= 1.2e-2 * scale * scale     # (3 x 2^-8: x rounded, squared, rounded)
    # ... exactly: the square of the bf16-rounded output, rounded -- what a consumer squaring the bf16 y it loads would form
    for res in (a, b):
        assert torch.equal(res["y_sq"], _sq_of_rounded(res["y"])) and torch.equal(res["y16"], res["y"].to(torch.bfloat16))
        sd = torch.sqrt(a["v"][s])
        ea = t(O.philox_normal(seed, O.tensor_id(2, 2), off + s, B, N)).to(dev)
        want = torch.where(sd > 0, ea / (2 * sd), torch.zeros_like(sd))
        assert float((a["hfac"][s] - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_lr_k_sliced_form_per_sample_inputs_and_injected_eps(dev):
    """K3s on a hidden layer's inputs (one x per MC sample) and with epsilon from memory: the on-chip draw dumped by one
    launch and injected into the next gives the same bits; K3a on the same injected epsilon agrees to summation order;
    sample groups (stacked minibatches) index x and the Philox subsequence as K3a does."""
    S, B, K, N = 2, 77, 392, 328
    rs = np.random.RandomState(5)
    mk = lambda *sh, lo=-0.3, hi=0.3: t(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
    dw = [mk(K, N), mk(K, N, lo=-5, hi=-4), mk(N), mk(N, lo=-5, hi=-4)]
    x3 = mk(S, B, K, lo=0, hi=1).to(torch.bfloat16)
    kw = dict(n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, seed=3, layer_id=1, sample_offset=9,
              want_kl=True, want_scalars=True)
    scratch = ops.lr_split_scratch(S, B, N, dev)
    ks = dict(form=L.FORM_GEMM_KSLICE, split_scratch=scratch)
    assert ops.lr_plan(x3, *dw, eps_mode=L.EPS_PHILOX, **ks, **kw)["form"] == L.FORM_GEMM_KSLICE
    a = ops.lr_linear_fwd(x3, *dw, eps_mode=L.EPS_PHILOX, dump_eps=True, **ks, **kw)
    b = ops.lr_linear_fwd(x3, *dw, eps_mode=L.EPS_MEMORY, eps_act=a["eps_act"], eps_b=a["eps_b"], **ks, **kw)
    c = ops.lr_linear_fwd(x3, *dw, eps_mode=L.EPS_MEMORY, eps_act=a["eps_act"], eps_b=a["eps_b"], form=L.FORM_TILE, **kw)
    d = ops.lr_linear_fwd(x3, *dw, eps_mode=L.EPS_PHILOX, dump_eps=True, form=L.FORM_TILE, **kw)
    assert torch.equal(a["y"], b["y"]) and torch.equal(a["kl3"], b["kl3"])
    assert torch.equal(a["eps_act"], d["eps_act"]) and torch.equal(a["eps_b"], d["eps_b"])     # the same epsilon map
    scale = float(c["y"].float().abs().max())
    assert float((a["y"].float() - c["y"].float()).abs().max()) <= 1e-2 * scale               # bf16 outputs: one ulp of rounding
    assert float((a["y"].float() - c["y"].float()).abs().mean()) <= 2e-4 * scale
    # the samples really read their own x: swapping the inputs swaps the outputs' pre-noise part
    e0 = torch.zeros_like(a["eps_act"])
    z = ops.lr_linear_fwd(x3, *dw, eps_mode=L.EPS_MEMORY, eps_act=e0, eps_b=torch.zeros_like(a["eps_b"]), **ks, **kw)
    zs = ops.lr_linear_fwd(x3.flip(0).contiguous(), *dw, eps_mode=L.EPS_MEMORY, eps_act=e0, eps_b=torch.zeros_like(a["eps_b"]), **ks, **kw)
    assert torch.equal(z["y"], zs["y"].flip(0)) and not torch.equal(z["y"][0], z["y"][1])
    # sample groups: 2 stacked minibatches x 1 sample = the two one-sample launches with the group's offsets
    g = ops.lr_linear_fwd(x3, *dw, eps_mode=L.EPS_PHILOX, sample_group=1, sample_group_stride=7, **ks, **kw)
    for s in range(S):
        kw1 = dict(kw, n_samples=1, sample_offset=9 + 7 * s)
        one = ops.lr_linear_fwd(x3[s], *dw, eps_mode=L.EPS_PHILOX, form=L.FORM_GEMM_KSLICE, split_scratch=ops.lr_split_scratch(1, B, N, dev), **kw1)
        assert torch.equal(g["y"][s], one["y"][0]), s


def test_lr_k_sliced_shared_input_with_injected_eps(dev):
    """Samples on ONE input (the unit's products made once; from four samples on the epilogues spread over the slice blocks)
    with epsilon from memory -- the seam the reference's stubbed `.normal` goes through: the on-chip draw dumped by one launch
    and injected into the next gives the same bits, K3a on the same injected epsilon agrees to summation order, and the
    samples differ from each other only by their noise (zero epsilon: every sample the same tile)."""
    for S, B, K, N in ((6, 77, 392, 328), (3, 128, 784, 1200), (10, 128, 784, 1200)):
        rs = np.random.RandomState(S)
        mk = lambda *sh, lo=-0.3, hi=0.3: t(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
        dw = [mk(K, N), mk(K, N, lo=-5, hi=-4), mk(N), mk(N, lo=-5, hi=-4)]
        x = mk(B, K, lo=0, hi=1)                                # fp32, [batch, in]: shared by all samples
        kw = dict(n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.float32, seed=11, layer_id=0, sample_offset=3,
                  want_kl=True, want_scalars=True)
        scratch = ops.lr_split_scratch(1, B, N, dev)
        ks = dict(form=L.FORM_GEMM_KSLICE, split_scratch=scratch)
        assert ops.lr_plan(x, *dw, eps_mode=L.EPS_PHILOX, **ks, **kw)["blocks"] == ((N + 31) // 32) * ((B + 127) // 128) * \
            ops.lr_plan(x, *dw, eps_mode=L.EPS_PHILOX, **ks, **kw)["k_slices"]
        a = ops.lr_linear_fwd(x, *dw, eps_mode=L.EPS_PHILOX, dump_eps=True, **ks, **kw)
        b = ops.lr_linear_fwd(x, *dw, eps_mode=L.EPS_MEMORY, eps_act=a["eps_act"], eps_b=a["eps_b"], **ks, **kw)
        c = ops.lr_linear_fwd(x.to(torch.bfloat16), *dw, eps_mode=L.EPS_MEMORY, eps_act=a["eps_act"], eps_b=a["eps_b"], form=L.FORM_TILE, **kw)
        d = ops.lr_linear_fwd(x.to(torch.bfloat16), *dw, eps_mode=L.EPS_PHILOX, dump_eps=True, form=L.FORM_TILE, **kw)
        assert torch.equal(a["y"], b["y"]) and torch.equal(a["kl3"], b["kl3"]), S
        assert torch.equal(a["eps_act"], d["eps_act"]) and torch.equal(a["eps_b"], d["eps_b"]), S     # the same epsilon map
        scale = float(c["y"].abs().max())
        assert float((a["y"] - c["y"]).abs().max()) <= 3e-3 * scale, S
        z = ops.lr_linear_fwd(x, *dw, eps_mode=L.EPS_MEMORY, eps_act=torch.zeros_like(a["eps_act"]), eps_b=torch.zeros_like(a["eps_b"]), **ks, **kw)
        assert all(torch.equal(z["y"][0], z["y"][s_]) for s_ in range(1, S)) and not torch.equal(a["y"][0], a["y"][1])
        assert int(scratch[:L.load().bnn_lr_split_scratch_zero_bytes(1, B, N) // 4].abs().sum()) == 0


def test_lr_k_sliced_form_over_random_shapes_against_k3a(dev):
    """K3s against K3a over shapes drawn at random among those its plan accepts (batch 1..300: ragged 16-row tiles and several
    batch blocks; K any multiple of 8 from 64: K tails inside a k-step, 1..8 slices, slices of unequal length; N any multiple
    of 4 from 64: a last 32-feature group of 4..28 features), both layer-input types: same epsilon map, outputs and KL sums
    within summation order, counters left at zero."""
    rs = np.random.RandomState(2024)
    done = 0
    for _ in range(60):
        S = int(rs.randint(1, 3)); B = int(rs.randint(1, 301)); K = 8 * int(rs.randint(8, 257)); N = 4 * int(rs.randint(16, 151))
        mk = lambda *sh, lo=-0.3, hi=0.3: t(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
        dw = [mk(K, N), mk(K, N, lo=-5, hi=-4), mk(N), mk(N, lo=-5, hi=-4)]
        x = mk(S, B, K, lo=0, hi=1) if rs.rand() < 0.5 else mk(B, K, lo=0, hi=1)
        if rs.rand() < 0.6:
            x = x.to(torch.bfloat16)
        kw = dict(n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=bool(rs.rand() < 0.5), y_dtype=torch.float32, seed=int(rs.randint(1 << 30)),
                  layer_id=int(rs.randint(4)), sample_offset=int(rs.randint(100)), want_kl=True, want_scalars=True, eps_mode=L.EPS_PHILOX)
        scratch = ops.lr_split_scratch(S, B, N, dev)
        try:
            plan = ops.lr_plan(x, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, **kw)
        except ops.BnnHipError:
            continue                                            # more than one round of blocks: the plan declines
        assert plan["form"] == L.FORM_GEMM_KSLICE and plan["blocks"] <= 256 and 1 <= plan["k_slices"] <= 8, (S, B, K, N, plan)
        a = ops.lr_linear_fwd(x, *dw, form=L.FORM_GEMM_KSLICE, split_scratch=scratch, **kw)
        b = ops.lr_linear_fwd(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), *dw, form=L.FORM_TILE, **kw)
        scale = float(b["y"].abs().max()) + 1e-6
        assert torch.isfinite(a["y"]).all(), (S, B, K, N)
        assert float((a["y"] - b["y"]).abs().max()) <= 3e-3 * scale, (S, B, K, N, plan, float((a["y"] - b["y"]).abs().max()), scale)
        close(a["kl3"][0], float(b["kl3"][0]), rtol=2e-6)
        SU = 1 if x.dim() == 2 else S                           # samples on one input share a unit (and its arrival counter)
        assert int(scratch[:L.load().bnn_lr_split_scratch_zero_bytes(SU, B, N) // 4].abs().sum()) == 0
        done += 1
    assert done >= 25, done


@pytest.mark.parametrize("form", ["tile", "gemm"])
def test_c5_wide_lr_layer_against_oracle(dev, form):
    """The local-reparameterisation twin of the C5 layer test: 4096 x 4096 [in,out] weights, batch 128, 4 MC
    samples, activation eps from the on-chip generator, K3a and K3b (with prepared fragments), against the oracle."""
    S, B, K, N, seed, off = 4, 128, 4096, 4096, 5, 11
    rs = np.random.RandomState(78)
    w_mu = rs.uniform(-0.2, 0.2, (K, N)).astype(np.float32)
    w_rho = rs.uniform(-5, -4, (K, N)).astype(np.float32)
    b_mu = rs.uniform(-0.2, 0.2, N).astype(np.float32)
    b_rho = rs.uniform(-5, -4, N).astype(np.float32)
    x = rs.uniform(0, 1, (B, K)).astype(np.float32)
    dw = [t(a).to(dev) for a in (w_mu, w_rho, b_mu, b_rho)]
    ref = []
    torch.set_num_threads(8)
    for s in range(S):
        ea = t(O.philox_normal(seed, O.tensor_id(2, 2), off + s, B, N))
        eb = t(O.philox_normal(seed, O.tensor_id(2, 1), off + s, 1, N))[0]
        y, kw_, kb_ = O.lr_linear(t(x), t(w_mu), t(w_rho), t(b_mu), t(b_rho), ea, eb, 1.0)
        ref.append((torch.relu(y).numpy(), float(kw_ + kb_)))
    torch.set_num_threads(1)
    runs = [("bf16", L.MATH_BF16)] + ([("f32", L.MATH_F32)] if form == "tile" else [])
    for name, mm in runs:
        if mm == L.MATH_F32:
            xin, xsq, wfrag, ws = t(x).to(dev), None, None, None
        else:
            xin, xsq = ops.cast_bf16(t(x).to(dev), want_sq=True)
            wfrag, ws = ops.lr_prepare(*dw) if form == "gemm" else (None, None)
        kw = dict(n_samples=S, sigma_p=1.0, math_mode=mm, relu=True, y_dtype=torch.float32, eps_mode=L.EPS_PHILOX, seed=seed,
                  layer_id=2, sample_offset=off, want_kl=True, want_scalars=True, x_sq=xsq, w_frag=wfrag, workspace=ws,
                  form=L.FORM_GEMM if form == "gemm" else L.FORM_TILE)
        plan = ops.lr_plan(xin, *dw, **kw)
        assert plan["form"] == (L.FORM_GEMM if form == "gemm" else L.FORM_TILE), plan
        out = ops.lr_linear_fwd(xin, *dw, **kw)
        close(out["kl3"][0], ref[0][1], rtol=1e-5)
        if mm == L.MATH_BF16:          # the squares this form hands the next layer: bf16(bf16(y)^2), as every other form's
            sq_out = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev)
            again = ops.lr_linear_fwd(xin, *dw, out_sq=sq_out, **kw)
            assert torch.equal(again["y"], out["y"]) and torch.equal(sq_out, _sq_of_rounded(out["y"])), (name, form)
        for s in range(S):
            scale = float(np.abs(ref[s][0]).max())
            err = float(np.abs(out["y"][s].double().cpu().numpy() - ref[s][0]).max())
            assert err <= (2e-5 if mm == L.MATH_F32 else 2e-2) * scale, (name, form, s, err, scale)


@pytest.mark.parametrize("shape", [(1, 128, 1200, 1200), (3, 128, 784, 1200), (2, 20, 72, 40), (24, 128, 1200, 1200)])
def test_sampling_job_riding_on_a_layer_launch(dev, shape):
    """bnn_bbb_fwd_args.rider: another layer's sampling (K1s) carried by a layer launch as extra blocks -- or, when the
    launch cannot carry it (block-GEMM form), launched ahead of it -- leaves the layer's own results bitwise unchanged
    and produces bitwise the weights, biases and statistics of a stand-alone bnn_bbb_sample_weights call."""
    S, B, K, N = shape
    rs = np.random.RandomState(33)
    mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
    x = mk(B, K, lo=0, hi=1).to(torch.bfloat16)
    w = [mk(N, K), mk(N, K, lo=-5, hi=-4), mk(N), mk(N, lo=-5, hi=-4)]
    w3 = [mk(10, N), mk(10, N, lo=-5, hi=-4), mk(10), mk(10, lo=-5, hi=-4)]
    prior = ops.PriorSpec(False, 1.0)
    kw = dict(n_samples=S, prior=prior, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=5,
              layer_id=1, sample_offset=17, want_stats=True)
    job = lambda: ops.build_sample_job([dict(w_mu=w3[0], w_rho=w3[1], b_mu=w3[2], b_rho=w3[3], prior=prior, layer_id=2)],
                                       n_samples=S, seed=5, sample_offset=17)
    plain = ops.bbb_linear_fwd(x, *w, **kw)
    alone = ops.bbb_sample_weights([dict(w_mu=w3[0], w_rho=w3[1], b_mu=w3[2], b_rho=w3[3], prior=prior, layer_id=2)],
                                   n_samples=S, seed=5, sample_offset=17)[0]
    j = job()
    ridden = ops.bbb_linear_fwd(x, *w, rider=j, **kw)
    torch.cuda.synchronize()
    Tm = int(plain["workspace"][:1].view(torch.int32)[0])                 # the layer's own statistics entries in use
    assert torch.equal(ridden["y"], plain["y"])
    assert torch.equal(ridden["workspace"][:4 * (1 + S * Tm)], plain["workspace"][:4 * (1 + S * Tm)])
    r = j[1][0]
    assert torch.equal(r["w"], alone["w"]) and torch.equal(r["b"], alone["b"])
    T = int(alone["workspace"][:1].view(torch.int32)[0])
    assert torch.equal(r["workspace"][:4 * (1 + S * T)], alone["workspace"][:4 * (1 + S * T)])


@pytest.mark.parametrize("dims,mode,B,G,S", [((784, 1200, 10), "classification", 128, 1, 1), ((784, 1200, 10), "classification", 128, 1, 16),
                                             ((784, 1200, 10), "classification", 100, 3, 2), ((16, 64, 1), "regression", 128, 1, 5),
                                             ((40, 72, 16), "classification", 37, 2, 3)])
def test_row_split_output_layer_equals_the_two_launch_form(dev, monkeypatch, dims, mode, B, G, S):
    """K1r (output layer over weights drawn by the rider, split by batch rows, finalize by the last block to arrive)
    against the plain sequence (sampling fused into the output layer's launch + bnn_elbo_finalize): the same Philox
    elements, so log p / log q agree to fp32 summation order and the logits / NLL to the bf16 rounding of w (the
    pre-sampled form rounds w to bf16 once, the fused form feeds the same rounded value to the matrix core), replay
    after replay (tickets back at zero, device counter advanced)."""
    from bnn_hip import engine
    bnn_hip.set_math("bf16")
    net, _ = build_net(dev, False, dims, mode, B=B)
    xs, ys = zip(*[synth.synth_batch(mode, B, dims[0], dims[2], seed=70 + m) for m in range(G)])
    xd, yd = torch.from_numpy(np.stack(xs)).to(dev), torch.from_numpy(np.stack(ys)).to(dev)
    res = []
    for form in (L.FORM_AUTO, L.FORM_TILE):
        monkeypatch.setattr(bnn_hip.runtime.state, "form", form)
        bnn_hip.manual_seed(8, counter=40)
        ev = engine.GraphedElbo(net, xd if G > 1 else xd[0], yd if G > 1 else yd[0], S, sigma=0.3, stacked=G > 1)
        assert ev.rows == (form == L.FORM_AUTO)
        if form == L.FORM_AUTO:
            ev0_split = list(ev.split)
        a = ev.replay().clone()
        b = ev.replay().clone()
        res.append((a, b, {k: v.clone() for k, v in ev.out.items()}, ev.logits.clone(), int(ev.counter.item())))
        if ev.rows:
            assert int(ev.scratch.view(torch.int32)[:G * S].abs().sum()) == 0 and int(ev.ticket.item()) == 0
    (a1, b1, o1, lg1, c1), (a2, b2, o2, lg2, c2) = res
    assert c1 == c2 == 40 + 3 * G * S and not torch.equal(a1, b1)
    scale = float(lg2.abs().max()) + 1e-6
    # the output layer alone pre-sampled: same bf16 operands, fp32 summation order only.  Hidden layers pre-sampled too
    # (n <= engine.PRESAMPLE_HIDDEN_MAX_SAMPLES): their matmul-only launches sum in another order, and a hidden
    # activation that lands on the other side of a bf16 rounding boundary moves by 2^-8 of its value
    # (and so does a hidden layer that takes the K-sliced block GEMM under FORM_AUTO: from 4 pairs per launch)
    hidden_pre = G * S <= engine.PRESAMPLE_HIDDEN_MAX_SAMPLES and len(dims) == 3
    hidden_other_form = any(sp is not None for sp in ev0_split)
    lg_tol, nll_tol = (2e-3, 1e-3) if (hidden_pre or hidden_other_form) else (2e-5, 2e-5)
    err = float((lg1 - lg2).abs().max())
    assert err <= lg_tol * scale, (err, scale)
    for k in o1:
        close(o1[k], o2[k].cpu().numpy(), rtol=nll_tol if k == "nll" else 2e-6)
    close(a1, a2.cpu().numpy(), rtol=nll_tol)
    close(b1, b2.cpu().numpy(), rtol=nll_tol)


def test_large_batch_layers_take_the_block_gemm(dev, monkeypatch):
    """Batches of >= 512 rows: every BBB layer is one sampling launch (K1s) + the 256 x 256 block form of the matmul
    over the sampled weights (K1g, bnn_bbb_linear_fwd with w_sampled) instead of the fused kernels; same Philox
    elements, so the evaluation agrees with the fused path to bf16 rounding of the operands, and with the fp32 oracle
    within the bf16 tolerances."""
    from bnn_hip import engine
    bnn_hip.set_math("bf16")
    B, dims, S = 512, (64, 96, 8), 3
    net, sd = build_net(dev, False, dims, "classification", B=B)
    x, y = synth.synth_batch("classification", B, dims[0], dims[2], seed=9)
    xd, yd = t(x).to(dev), t(y).to(dev)
    res = []
    for form in (L.FORM_AUTO, L.FORM_TILE):
        monkeypatch.setattr(bnn_hip.runtime.state, "form", form)
        bnn_hip.manual_seed(4, counter=10)
        monkeypatch.setattr(engine, "SAMPLE_BESIDE_MATMUL", True)   # (the side-stream form of the sampling launches: checked below)
        ev = engine.GraphedElbo(net, xd, yd, S)
        assert all(ev.lib) == (form == L.FORM_AUTO)
        sums = ev.replay().clone()
        res.append((sums, {k: v.clone() for k, v in ev.out.items()}, ev.logits.clone()))
        if form == L.FORM_AUTO:                                  # the eager product path takes the same route
            bnn_hip.manual_seed(4, counter=10 + S)
            with torch.no_grad():
                tup = net.sample_elbo(xd, yd, 0.5, S)
            close(tup[1] * S, float(sums[0]), rtol=1e-6)
            close(tup[2] * S, float(sums[1]), rtol=1e-6)
            close(tup[3] * S, float(sums[2]), rtol=1e-5)
    # the sampling launches run on a side stream beside the matmuls (forked and joined inside the captured evaluation): the
    # same evaluation with everything on one stream gives the same bits, replay after replay
    monkeypatch.setattr(bnn_hip.runtime.state, "form", L.FORM_AUTO)
    monkeypatch.setattr(engine, "SAMPLE_BESIDE_MATMUL", False)
    bnn_hip.manual_seed(4, counter=10)
    ev1 = engine.GraphedElbo(net, xd, yd, S)
    assert ev1.side is None and all(ev1.lib)
    monkeypatch.setattr(engine, "SAMPLE_BESIDE_MATMUL", True)
    bnn_hip.manual_seed(4, counter=10)
    ev2 = engine.GraphedElbo(net, xd, yd, S, evals_per_replay=2)
    assert ev2.side is not None
    for rep_ in range(3):
        a = ev1.replay().clone(); ev1.replay()                   # two evaluations per step on either side
        b = ev2.replay().clone()
        torch.cuda.synchronize()
        assert torch.equal(ev1.logits, ev2.logits) and torch.equal(ev1.sums, ev2.sums), rep_
    (s1, o1, lg1), (s2, o2, lg2) = res
    close(o1["log_prior"], o2["log_prior"].cpu().numpy(), rtol=2e-6)
    close(o1["log_q"], o2["log_q"].cpu().numpy(), rtol=2e-6)
    assert float((lg1 - lg2).abs().max()) <= BF16_LOGIT_TOL * float(lg2.abs().max())
    close(o1["nll"], o2["nll"].cpu().numpy(), rtol=BF16_NLL_RTOL)
    p = O.NetParams.from_state_dict(sd, "classification", dims[0], False, O.Prior.from_init([1.0], False))
    want, _ = _oracle_pairs(p, [x], [y], 4, 10 + S, S)           # the captured graph's replay followed the warm-up
    close(o1["log_prior"], want[:, 0], rtol=1e-5)
    close(o1["log_q"], want[:, 1], rtol=1e-5)
    close(o1["nll"], want[:, 2], rtol=BF16_NLL_RTOL)


BLOCK_SHAPES = [  # (samples, batch, out, in, x shared by the samples): edges of every kind
    (1, 256, 256, 64, False),      # one K-tile, one block
    (2, 256, 256, 128, False),     # two K-tiles: prologue + the two tail forms only
    (2, 300, 520, 192, False),     # ragged batch and feature edges (clamped rows), three K-tiles
    (2, 512, 1200, 784, True),     # the MNIST first layer at 512 rows: K % 64 = 16 (zero-filled chunks), x shared
    (3, 1024, 1200, 1200, False),  # the MNIST hidden layer at 1024 rows: K % 64 = 48, N % 256 = 176
    (1, 100, 40, 8, False),        # everything smaller than one tile; K = 8: a single chunk
    (5, 70, 10, 1200, False),      # output-layer shape: N % 4 != 0 -> the scalar bias / store path
]


@pytest.mark.parametrize("shape", BLOCK_SHAPES)
@pytest.mark.parametrize("y_dtype", [torch.float32, torch.bfloat16])
def test_block_gemm_against_fp64_reference(dev, shape, y_dtype):
    """K1g alone (bnn_bbb_linear_fwd, w_sampled, BNN_FORM_BLOCK256 forced so small batches take it too): y[s] =
    act(x[s] . w[s]^T + b[s]) on bf16 operands against the same product in float64 on the host.  The bf16 products are
    exact in fp32, so only the accumulation order differs: fp32 y to 1e-5 of the row scale, bf16 y to one bf16 ulp."""
    S, B, N, K, shared = shape
    rs = np.random.RandomState(S * 1000 + N)
    x = torch.from_numpy(rs.uniform(0, 1, ((1 if shared else S), B, K)).astype(np.float32)).to(torch.bfloat16)
    w = torch.from_numpy(rs.uniform(-0.2, 0.2, (S, N, K)).astype(np.float32)).to(torch.bfloat16)
    # asymmetric operands: a transposed C write or a swapped fragment cannot pass
    w[:, 0, :] = 0.25
    x[:, -1, :] = 0.5
    b = torch.from_numpy(rs.uniform(-0.2, 0.2, (S, N)).astype(np.float32))
    ref = torch.einsum("sbk,snk->sbn", x.double().expand(S, B, K), w.double()) + b.double()[:, None, :]
    for relu in (True, False):
        want = torch.relu(ref) if relu else ref
        xin = x[0].to(dev) if shared else x.to(dev)
        sentinel = torch.full((S, B, N), float("nan"), dtype=y_dtype, device=dev)
        got = ops.bbb_sampled_matmul(xin, w.to(dev), b.to(dev), n_samples=S, relu=relu, y_dtype=y_dtype, out=sentinel,
                                     form=L.FORM_BLOCK256)
        g = got.double().cpu()
        assert torch.isfinite(g).all()
        scale = float(want.abs().max())
        tol = (1e-5 if y_dtype == torch.float32 else 2.0 ** -8) * scale
        err = float((g - want).abs().max())
        assert err <= tol, (shape, y_dtype, relu, err, scale)
    plan_args = ops._bbb_build(xin, None, None, None, None, n_samples=S, prior=ops.PriorSpec(), math_mode=L.MATH_BF16, relu=True,
                               y_dtype=y_dtype, eps_mode=L.EPS_ZERO, want_stats=False, w_sampled=w.to(dev), b_sampled=b.to(dev),
                               form=L.FORM_BLOCK256)[0]
    pl = L.Plan()
    L.check(L.load().bnn_bbb_plan(plan_args, pl), "bnn_bbb_plan")
    assert pl.form == L.FORM_BLOCK256 and pl.blocks == S * ((B + 255) // 256) * ((N + 255) // 256) and pl.waves == 8


@pytest.mark.parametrize("S,B", [(2, 1024), (1, 4096)])
def test_c5_wide_bbb_layer_batch_1024_block_form_against_oracle(dev, S, B):
    """BASELINE configs[4]'s matrix-core-bound point at full size: one 4096 x 4096 BayesianLinear layer fed 1024 batch
    rows (2 MC samples) and 4096 batch rows (the benched shape: 16 x 16 block tiles per sample), on-chip Philox: the sampling
    launch (K1s) + the block form of the matmul (K1g, chosen by the plan for >= 512 rows), against the oracle's layer
    (networks.py:73-88) on the same epsilon: y to C5_POINTS_TOL of scale against the restatement with the device's rounding
    points, 2e-2 against the fp32 arithmetic (K = 4096 products of bf16-rounded operands), the fp32 statistics to 1e-5."""
    K, N, seed, off = 4096, 4096, 99, 40
    rs = np.random.RandomState(78)
    w_mu = rs.uniform(-0.2, 0.2, (N, K)).astype(np.float32)
    w_rho = rs.uniform(-5, -4, (N, K)).astype(np.float32)
    b_mu = rs.uniform(-0.2, 0.2, N).astype(np.float32)
    b_rho = rs.uniform(-5, -4, N).astype(np.float32)
    x = rs.uniform(0, 1, (B, K)).astype(np.float32)
    prior = O.Prior.from_init([1.0], False)
    dw = [t(a).to(dev) for a in (w_mu, w_rho, b_mu, b_rho)]
    sm = ops.bbb_sample_weights([dict(w_mu=dw[0], w_rho=dw[1], b_mu=dw[2], b_rho=dw[3], prior=ops.PriorSpec(False, 1.0), layer_id=1)],
                                n_samples=S, seed=seed, sample_offset=off)[0]
    xin = t(x).to(dev).to(torch.bfloat16)
    y = ops.bbb_sampled_matmul(xin, sm["w"], sm["b"], n_samples=S, relu=True, y_dtype=torch.float32)
    # per-sample statistics from the sampler's workspace: {sum eps^2, sum w^2, sum log sigma (sample 0's entries)} per tile
    ws = sm["workspace"]
    T = int(ws[:1].view(torch.int32)[0])
    per = ws[4:4 + 4 * S * T].view(S, T, 4).double().sum(1).cpu().numpy()
    sls = float(ws[4:4 + 4 * T].view(T, 4).double().sum(0)[2])
    n_el, c0 = N * K + N, -0.5 * math.log(2.0 * math.pi)
    torch.set_num_threads(8)
    for s in range(S):
        ew = t(O.philox_normal(seed, O.tensor_id(1, 0), off + s, N, K))
        eb = t(O.philox_normal(seed, O.tensor_id(1, 1), off + s, 1, N))[0]
        yr, lp, lq = O.bbb_linear(t(x), t(w_mu), t(w_rho), t(b_mu), t(b_rho), ew, eb, prior)
        yr16 = torch.relu(O.bbb_linear_bf16(t(x), t(w_mu), t(w_rho), t(b_mu), t(b_rho), ew, eb, prior)[0]).numpy()
        yr = torch.relu(yr).numpy()
        scale = float(np.abs(yr).max())
        got = y[s].double().cpu().numpy()
        err = float(np.abs(got - yr).max())
        err16 = float(np.abs(got - yr16).max())
        print(f"\n[C5 K1s + K1g, batch {B}] sample {s}: vs rounding-point oracle {err16 / scale:.2e}, vs fp32 oracle {err / scale:.2e} of scale")
        assert err16 <= C5_POINTS_TOL * scale, (s, err16 / scale)            # the kernels: same rounding points
        assert err <= 2e-2 * scale, (s, err, scale)                          # the precision choice (bf16 operands, K = 4096)
        close(n_el * c0 - 0.5 * per[s, 1], float(lp), rtol=1e-5)             # Gaussian prior, sigma_p = 1 (networks.py:67-68)
        close(n_el * c0 - sls - 0.5 * per[s, 0], float(lq), rtol=1e-5)       # networks.py:46 at w = mu + sigma eps
    torch.set_num_threads(1)


def test_c5_wide_network_elbo_against_oracle(dev):
    """BASELINE configs[4] end to end at full size: the 4096-4096-4096 BayesianLinear stack (50 343 936 stochastic
    parameters), batch 128, 4 MC samples (the per-GPU share of C5's 32), on-chip Philox, through the product API
    (BayesianNetwork.elbo_many -> engine.GraphedElbo: networks.py:166-172 three times per sample, :174-178, :183-190,
    :192-209), per math mode against the oracle on the same epsilon:
      bf16    -- log p / log q (fp32 statistics) rtol 1e-5 against the fp32 oracle; the NLL and the ELBO at every beta of the
                 schedule rtol 1e-4 against the oracle with the device's rounding points (O.network_forward_bf16);
      bf16x3  -- everything, the ELBO at every beta included, rtol 1e-4 against the FP32 oracle (the reference's arithmetic)."""
    S, B, dims, seed, first = 4, 128, (4096, 4096, 4096), 777, 50
    net, sd = build_net(dev, False, dims, "regression", B=B)
    p = O.NetParams.from_state_dict(sd, "regression", dims[0], False, O.Prior.from_init([1.0], False))
    x, y = synth.synth_batch("regression", B, dims[0], dims[2])
    rows = {"f32": [], "bf16": []}
    torch.set_num_threads(8)
    for j in range(S):
        eps = O.philox_eps_for_network(p, B, seed, first + j)
        for key, fwd in (("f32", O.network_forward), ("bf16", O.network_forward_bf16)):
            out, a, b = fwd(p, t(x), eps)
            rows[key].append([float(a), float(b), float(O.nll(out, t(y), "regression", 1.0))])
        del eps
    torch.set_num_threads(1)
    want = {k: np.asarray(v, np.float64).sum(0) for k, v in rows.items()}
    xd, yd = t(x).to(dev)[None], t(y).to(dev)[None]
    for mode, key in (("bf16", "bf16"), ("bf16x3", "f32")):
        bnn_hip.set_math(mode)
        bnn_hip.manual_seed(seed, counter=first)
        sums = net.elbo_many(xd, yd, S).double().cpu().numpy()[0]
        assert sums[3] == S
        close(sums[0], want["f32"][0], rtol=1e-5)
        close(sums[1], want["f32"][1], rtol=1e-5)
        nll_err = abs(sums[2] - want[key][2]) / want[key][2]
        print(f"\n[C5 network, {mode}] sum nll vs the {key} oracle {nll_err:.2e}; vs the fp32 oracle {abs(sums[2] - want['f32'][2]) / want['f32'][2]:.2e}")
        assert nll_err <= 1e-4
        for beta in BETAS:
            close(_elbo(sums[:3], S, beta, False), _elbo(want[key], S, beta, False), rtol=1e-4)


# ------------------------------------------------------------------ split-bf16 math (BNN_MATH_BF16X3)
# The mode's promise is the reference's fp32 F.linear (networks.py:88) to ~1e-5 of the output scale: every product runs as
# x_hi w_hi + x_hi w_lo + x_lo w_hi with (hi, lo) = (bf16(v), bf16(v - hi)), i.e. <= ~2^-15 relative per product against 2^-8
# in plain bf16 math.  Two bounds: (i) the KERNELS against the CPU restatement with the same rounding points
# (O.linear_bf16x3 / O.network_forward_bf16x3: fp32 accumulation order is all that differs), (ii) the MODE against the fp32
# oracle = the reference's arithmetic -- NLL rtol 1e-4, which makes the ELBO rtol 1e-4 at every beta of class_task.py:70.
X3_POINTS_RTOL = 2e-5          # of the output scale, kernels vs the rounding-point restatement
X3_NLL_RTOL = 1e-4             # the north star's tolerance, against the fp32 oracle


def _split(x):
    hi = x.to(torch.bfloat16)
    return hi, (x - hi.float()).to(torch.bfloat16)


@pytest.mark.parametrize("shape", [(1200, 1200, 128, 6), (784, 1200, 128, 5), (200, 1000, 100, 9), (256, 200, 257, 3), (72, 40, 300, 5),
                                   (33, 65, 5, 2), (1, 7, 3, 1), (1200, 10, 128, 4)])
def test_split_bf16_layer_forms_against_the_rounding_point_oracle(dev, shape):
    """One BayesianLinear layer in split-bf16 math through every form it has -- the tile form on fp32 activations (split in
    registers) and on bf16 plane pairs (x / x_lo), the block-GEMM form K1b2<X3> (hoisted sigma, pairs of units) -- on
    injected epsilon against O.linear_bf16x3 over the oracle's sampled weights, against the fp32 F.linear of networks.py:88,
    the statistics against the fp32 oracle; the bf16 output planes (y / y_lo) must be the split of the fp32 output."""
    K, N, B, S = shape
    gen = torch.Generator(device="cpu").manual_seed(K * 17 + N + S)
    x = torch.rand(S, B, K, generator=gen)
    wm = ((torch.rand((N, K), generator=gen) - 0.5) * 0.4)
    wr = (torch.rand((N, K), generator=gen) - 5.0)
    bm = ((torch.rand(N, generator=gen) - 0.5) * 0.4)
    br = (torch.rand(N, generator=gen) - 5.0)
    ew, eb = torch.randn(S, N, K, generator=gen), torch.randn(S, N, generator=gen)
    prior = O.Prior.from_init([0.9], False)
    want3, want32, lp, lq = [], [], [], []
    for s in range(S):
        w, b = O.sample_gaussian(wm, wr, ew[s]), O.sample_gaussian(bm, br, eb[s])
        want3.append(torch.relu(O.linear_bf16x3(x[s], w, b)))
        want32.append(torch.relu(torch.nn.functional.linear(x[s], w, b)))
        lp.append(float(prior.log_prob(w).sum() + prior.log_prob(b).sum()))
        lq.append(float(O.log_q(w, wm, wr).sum() + O.log_q(b, bm, br).sum()))
    want3, want32 = torch.stack(want3).numpy(), torch.stack(want32).numpy()
    scale = float(np.abs(want32).max()) + 1e-6
    assert np.abs(want3 - want32).max() <= 4e-5 * scale                # the restatement itself: the mode's arithmetic
    d = lambda a: a.to(dev)
    xd, (xh, xl) = d(x), _split(d(x))
    kw = dict(n_samples=S, prior=ops.PriorSpec(False, 0.9), math_mode=L.MATH_BF16X3, relu=True, eps_mode=L.EPS_MEMORY, eps_w=d(ew),
              eps_b=d(eb), want_stats=True, want_scalars=True)
    sig = ops.softplus(d(wr))
    runs = [("tile/f32 x", xd, dict(form=L.FORM_TILE)), ("tile/plane pair", xh, dict(form=L.FORM_TILE, x_lo=xl))]
    if K % 8 == 0 and K >= 8:
        runs.append(("block GEMM", xh, dict(form=L.FORM_GEMM, x_lo=xl, w_sigma=sig)))
    for name, xin, extra in runs:
        for y_dtype in (torch.float32, torch.bfloat16):
            plan = ops.bbb_plan(xin, d(wm), d(wr), d(bm), d(br), y_dtype=y_dtype, **kw, **extra)
            if name == "block GEMM" and S * ((B + 127) // 128) >= 4:
                assert plan["form"] == L.FORM_GEMM and plan["waves"] == 8 and plan["lds_bytes"] == (4 * 256 + 2 * 2 * 1024) * 16, plan
            else:
                assert plan["form"] == L.FORM_TILE, (name, plan)
            out = ops.bbb_linear_fwd(xin, d(wm), d(wr), d(bm), d(br), y_dtype=y_dtype, **kw, **extra)
            if y_dtype == torch.float32:
                y = out["y"].cpu().numpy()
                assert out["y_lo"] is None
            else:
                y = (out["y"].float() + out["y_lo"].float()).cpu().numpy()
                hi = torch.from_numpy(want3).to(torch.bfloat16)
                # the hi plane is the RNE rounding of the fp32 output (ulp flips where the device's sum rounds the other way)
                assert float((out["y"].float().cpu() - hi.float()).abs().max()) <= 2.0 ** -7 * scale, name
            assert np.abs(y - want3).max() <= (X3_POINTS_RTOL if y_dtype == torch.float32 else 2e-5 + 2.0 ** -16) * scale, (name, y_dtype, np.abs(y - want3).max() / scale)
            assert np.abs(y - want32).max() <= 6e-5 * scale, (name, y_dtype)
            close(out["log_prior"], lp, rtol=2e-5)
            close(out["log_q"], lq, rtol=2e-5)


X3_SHAPES = [(1, 1, None), (1, 8, None), (4, 2, None), (20, 1, None), (256, 1, (0, 1, 17, 63, 64, 127, 128, 200, 254, 255))]


@pytest.mark.parametrize("G,S,pairs", [(1, 8, None), (4, 2, None), (20, 1, None), (256, 1, (0, 1, 17, 63, 64, 127, 128, 200, 254, 255))])
def test_split_bf16_lr_timed_path_meets_the_fp32_tolerance_at_every_beta(dev, G, S, pairs):
    """The local-reparameterisation network under set_math('bf16x3') through engine.GraphedElbo (captured, on-chip Philox): a
    stream of stacked minibatches runs its hidden layers as K3b<X3> (split-bf16 mean product, bf16 variance product over
    bnn_lr_prepare_x3's fragments, plane pairs between the layers) and its output layer in exact fp32; one minibatch runs exact
    fp32 throughout.  Either way, against the FP32 oracle (= the reference's arithmetic, networks.py:116-138, :211-225) on the
    same epsilon: KL rtol 1e-5, NLL rtol 1e-4, hence the ELBO rtol 1e-4 at every beta; and against the CPU restatement with
    the mode's rounding points (O.network_forward_bf16x3) 2e-5."""
    from bnn_hip import engine
    B, dims, seed, first, E = 128, (784, 1200, 10), 454545, 11000, 2
    bnn_hip.set_math("bf16x3")
    net, sd = build_net(dev, True, dims, "classification")
    p = O.NetParams.from_state_dict(sd, "classification", dims[0], True, O.Prior.from_init([1.0], False))
    xs, ys = zip(*[synth.synth_batch("classification", B, dims[0], dims[2], seed=100 + m) for m in range(G)])
    xd = torch.from_numpy(np.stack(xs)).to(dev)
    yd = torch.from_numpy(np.stack(ys)).to(dev)
    bnn_hip.manual_seed(seed, counter=first)
    ev = engine.GraphedElbo(net, xd if G > 1 else xd[0], yd if G > 1 else yd[0], S, stacked=G > 1, evals_per_replay=E)
    assert ev.lr_x3 == (G > 1) and ev.math == (L.MATH_BF16X3 if G > 1 else L.MATH_F32)
    sums = ev.replay().clone().view(G, 4).double().cpu().numpy()
    torch.cuda.synchronize()
    total = G * S
    base = first + total * E
    idx = np.arange(total) if pairs is None else np.asarray(pairs)
    want, want_logits = _oracle_pairs(p, xs, ys, seed, base, S, pairs=idx)
    close(ev.out["kl"].double().cpu().numpy()[idx], want[:, 0], rtol=1e-5)
    got_nll = ev.out["nll"].double().cpu().numpy()[idx]
    lg = ev.logits.double().cpu().numpy().reshape(total, B, dims[2])[idx]
    nll_err = np.abs(got_nll - want[:, 2]) / np.abs(want[:, 2])
    lg_err = np.abs(lg - want_logits).max() / np.abs(want_logits).max()
    msg = f"\n[bf16x3 LR at C3] G={G} S={S}: vs fp32 oracle nll {nll_err.max():.2e} logits {lg_err:.2e} of scale"
    if G > 1:
        w3, w3_logits = [], []
        torch.set_num_threads(8)
        for f in idx:
            m, j = divmod(int(f), S)
            out, a, _ = O.network_forward_bf16x3(p, t(xs[m]), O.philox_eps_for_network(p, B, seed, base + m * S + j))
            w3.append(float(O.nll(out, t(ys[m]), p.mode)))
            w3_logits.append(out.numpy())
        torch.set_num_threads(1)
        nll3_err = np.abs(got_nll - np.asarray(w3)) / np.abs(np.asarray(w3))
        lg3_err = np.abs(lg - np.stack(w3_logits)).max() / np.abs(np.stack(w3_logits)).max()
        msg += f"; vs rounding-point restatement nll {nll3_err.max():.2e} logits {lg3_err:.2e}"
        assert nll3_err.max() <= 2e-5 and lg3_err <= 5e-5
    print(msg)
    assert nll_err.max() <= X3_NLL_RTOL and lg_err <= 1e-4
    assert (sums[:, 3] == S).all()
    full_eval = [m for m in range(G) if all((m * S + j) in set(idx.tolist()) for j in range(S))]
    pos = {int(f): i for i, f in enumerate(idx.tolist())}
    assert full_eval
    for m in full_eval:
        e32 = want[[pos[m * S + j] for j in range(S)]].sum(0)
        close(sums[m, 0], e32[0], rtol=1e-5)
        close(sums[m, 2], e32[2], rtol=X3_NLL_RTOL)
        for beta in BETAS:
            close(_elbo(sums[m, :3], S, beta, True), _elbo(e32, S, beta, True), rtol=1e-4)


@pytest.mark.parametrize("G,S,pairs", X3_SHAPES)
def test_split_bf16_timed_path_meets_the_fp32_tolerance_at_every_beta(dev, G, S, pairs):
    """The path bench.py times (engine.GraphedElbo, captured, on-chip Philox, G stacked minibatches x S samples per launch
    group) in split-bf16 math against the FP32 oracle (O.network_forward = the reference's arithmetic) on the same epsilon,
    per (minibatch, sample) pair: NLL rtol 1e-4, hence the ELBO rtol 1e-4 at beta = 0.5, 2^-10 and 0 (class_task.py:70) --
    what plain bf16 math meets only for the first ~6 minibatches of an epoch; statistics rtol 1e-5; and against the CPU
    restatement with the mode's rounding points 2e-5."""
    from bnn_hip import engine
    B, dims, seed, first, E = 128, (784, 1200, 10), 434343, 9000, 2
    bnn_hip.set_math("bf16x3")
    net, sd = build_net(dev, False, dims, "classification")
    p = O.NetParams.from_state_dict(sd, "classification", dims[0], False, O.Prior.from_init([1.0], False))
    xs, ys = zip(*[synth.synth_batch("classification", B, dims[0], dims[2], seed=100 + m) for m in range(G)])
    xd = torch.from_numpy(np.stack(xs)).to(dev)
    yd = torch.from_numpy(np.stack(ys)).to(dev)
    bnn_hip.manual_seed(seed, counter=first)
    ev = engine.GraphedElbo(net, xd if G > 1 else xd[0], yd if G > 1 else yd[0], S, stacked=G > 1, evals_per_replay=E)
    assert ev.x3 and all((b is not None) == (i < 2) for i, b in enumerate(ev.bufs_lo))
    sums = ev.replay().clone().view(G, 4).double().cpu().numpy()
    torch.cuda.synchronize()
    total = G * S
    assert int(ev.counter.item()) == first + total * (1 + E)
    base = first + total * E
    idx = np.arange(total) if pairs is None else np.asarray(pairs)
    want, want_logits = _oracle_pairs(p, xs, ys, seed, base, S, pairs=idx)
    w3, w3_logits = [], []
    torch.set_num_threads(8)
    for f in idx:
        m, j = divmod(int(f), S)
        out, a, b = O.network_forward_bf16x3(p, t(xs[m]), O.philox_eps_for_network(p, B, seed, base + m * S + j))
        w3.append(float(O.nll(out, t(ys[m]), p.mode)))
        w3_logits.append(out.numpy())
    torch.set_num_threads(1)
    w3, w3_logits = np.asarray(w3), np.stack(w3_logits)
    for c, k in enumerate(("log_prior", "log_q")):
        close(ev.out[k].double().cpu().numpy()[idx], want[:, c], rtol=1e-5)
    got_nll = ev.out["nll"].double().cpu().numpy()[idx]
    lg = ev.logits.double().cpu().numpy().reshape(total, B, dims[2])[idx]
    nll_err = np.abs(got_nll - want[:, 2]) / np.abs(want[:, 2])
    lg_err = np.abs(lg - want_logits).max() / np.abs(want_logits).max()
    nll3_err = np.abs(got_nll - w3) / np.abs(w3)
    lg3_err = np.abs(lg - w3_logits).max() / np.abs(w3_logits).max()
    print(f"\n[bf16x3 at C2] G={G} S={S}: vs fp32 oracle nll {nll_err.max():.2e} logits {lg_err:.2e} of scale; "
          f"vs rounding-point restatement nll {nll3_err.max():.2e} logits {lg3_err:.2e}")
    assert nll_err.max() <= X3_NLL_RTOL and lg_err <= 1e-4
    assert nll3_err.max() <= 2e-5 and lg3_err <= 3e-5
    assert (sums[:, 3] == S).all()
    full_eval = [m for m in range(G) if all((m * S + j) in set(idx.tolist()) for j in range(S))]
    pos = {int(f): i for i, f in enumerate(idx.tolist())}
    assert full_eval
    for m in full_eval:
        e32 = want[[pos[m * S + j] for j in range(S)]].sum(0)
        close(sums[m, 0], e32[0], rtol=1e-5)
        close(sums[m, 1], e32[1], rtol=1e-5)
        close(sums[m, 2], e32[2], rtol=X3_NLL_RTOL)
        for beta in BETAS:
            close(_elbo(sums[m, :3], S, beta, False), _elbo(e32, S, beta, False), rtol=1e-4)      # EVERY beta, against fp32


def test_split_bf16_math_through_the_drop_in_module(dev):
    """set_math('bf16x3') through the nn.Module surface: sample_elbo (4-tuple, injected epsilon, differentiable) equals the
    exact-fp32 mode to the mode's tolerance, gradients included (the backward kernels run the exact-fp32 matrix core), and a
    local-reparameterisation network runs the mode as exact fp32 (same results as set_math('f32'))."""
    B, dims, S = 64, (40, 72, 5), 2
    outs = {}
    for lr in (False, True):
        for mode in ("f32", "bf16x3"):
            bnn_hip.set_math(mode)
            net, sd = build_net(dev, lr, dims, "classification", B=B)
            install_eps(net, B, S, lr)
            x, y = synth.synth_batch("classification", B, dims[0], dims[2], seed=3)
            res = (net.sample_elbo_lr if lr else net.sample_elbo)(t(x).to(dev), t(y).to(dev), 0.25, S)
            res[0].backward()
            outs[(lr, mode)] = ([r.detach().double().cpu().numpy() for r in res],
                                [p.grad.detach().double().cpu().numpy() for p in net.parameters()])
    for lr in (False, True):
        (r32, g32), (r3, g3) = outs[(lr, "f32")], outs[(lr, "bf16x3")]
        for a, b in zip(r3, r32):
            close(a, b, rtol=0.0 if lr else 1e-4)
        for a, b in zip(g3, g32):
            assert np.abs(a - b).max() <= (0.0 if lr else 2e-4) * (np.abs(b).max() + 1e-12)
    # the forward-only product paths on on-chip epsilon (engine.run_layers: the input cast to a plane pair from 4 samples on,
    # plane pairs between the layers, the tile / block forms as the plans choose): sample_elbo under no_grad, predict_mc and
    # forward_mc in split-bf16 math against exact-fp32 math on the same Philox indices, C2-sized layers
    dims, B = (784, 1200, 10), 128
    res = {}
    for mode in ("f32", "bf16x3"):
        bnn_hip.set_math(mode)
        net, sd = build_net(dev, False, dims, "classification", B=B)
        x, y = synth.synth_batch("classification", B, dims[0], dims[2], seed=5)
        xd, yd = t(x).to(dev), t(y).to(dev)
        bnn_hip.manual_seed(77, counter=0)
        with torch.no_grad():
            e1 = net.sample_elbo(xd, yd, 2.0 ** -10, 1)
            e8 = net.sample_elbo(xd, yd, 2.0 ** -10, 8)
            lg = net.forward_mc(xd, 10)
            preds, probs = net.predict_mc(xd, 10)
        res[mode] = ([v.double().cpu().numpy() for v in e1 + e8], lg.double().cpu().numpy(), probs.double().cpu().numpy())
    for a, b in zip(res["bf16x3"][0], res["f32"][0]):
        close(a, b, rtol=1e-4)
    assert np.abs(res["bf16x3"][1] - res["f32"][1]).max() <= 1e-4 * np.abs(res["f32"][1]).max()
    assert np.abs(res["bf16x3"][2] - res["f32"][2]).max() <= 1e-4
