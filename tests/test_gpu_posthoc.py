"""F3 / F4 on the GPU: the ECE and SNR-pruning kernels against the oracle's restatement of compute_ece.py:14-57 and
weight_pruning.py:85-115 (parity UNPINNED for these rows: neither reference module imports here, see oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bnn_hip import ops, posthoc, synth
from oracle import bnn_oracle as O


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _probs(n, classes, seed, alpha):
    rs = np.random.RandomState(seed)
    p = rs.dirichlet([alpha] * classes, size=n).astype(np.float32)
    labels = np.where(rs.uniform(size=n) < 0.7, p.argmax(1), rs.randint(0, classes, n)).astype(np.int64)
    return p, labels


@pytest.mark.parametrize("n,classes,step,alpha", [(10000, 10, 0.1, 0.3), (1000, 10, 0.1, 0.2), (4097, 3, 0.25, 0.5), (257, 37, 0.1, 0.05)])
def test_ece_kernel_matches_the_reference_restatement(dev, n, classes, step, alpha):
    p, labels = _probs(n, classes, 11 + n, alpha)
    ece_ref, centers_ref, acc_ref, (cnt, cor, conf) = O.ece_reference(p, labels, step, classes)
    crit = posthoc.ECELoss(bin_step=step, num_classes=classes)
    ece, centers, acc = crit(torch.from_numpy(p).to(dev), torch.from_numpy(labels).to(dev))
    c2, r2, m2 = crit.bins(torch.from_numpy(p).to(dev), torch.from_numpy(labels).to(dev))
    np.testing.assert_array_equal(c2, cnt)
    np.testing.assert_array_equal(r2, cor)
    have = cnt > 0
    np.testing.assert_allclose(m2[have], conf[have], rtol=1e-6)
    np.testing.assert_allclose(centers, centers_ref, rtol=0, atol=0)
    np.testing.assert_allclose(acc, acc_ref, rtol=1e-12)
    if have.all():                                         # the reference's own loop is only defined then
        np.testing.assert_allclose(ece, ece_ref, rtol=1e-6)
    else:                                                  # the kernel skips empty bins: the plain definition of ECE
        want = sum(abs(conf[b] - cor[b] / cnt[b]) * cnt[b] / cnt.sum() for b in range(len(cnt)) if cnt[b] > 0)
        np.testing.assert_allclose(ece, want, rtol=1e-6)
    again, _, _ = crit(torch.from_numpy(p).to(dev), torch.from_numpy(labels).to(dev))
    assert again == ece                                    # no float atomics: bitwise reproducible


def test_ece_of_mc_averaged_predictions(dev):
    """The consumer chain of compute_ece.py:62-78: predict (MC-averaged softmax) -> ECE, all on the device."""
    import networks
    torch.manual_seed(0)
    mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=64, mode="classification", mu_init=[-0.2, 0.2],
              rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=False)
    net = networks.BayesianNetwork(mp).to(dev).eval()
    xs, ys = zip(*[synth.synth_batch("classification", 128, 784, 10, seed=40 + i) for i in range(4)])
    probs = torch.cat([net.predict_mc(torch.from_numpy(x).to(dev), 5)[1] for x in xs])
    labels = torch.from_numpy(np.concatenate(ys)).to(dev)
    ece, centers, acc = posthoc.ECELoss()(probs, labels)
    ref = O.ece_reference(probs.cpu().numpy(), labels.cpu().numpy())
    cnt = ref[3][0]
    assert 0.0 <= ece <= 1.0 and len(centers) == int((cnt > 0).sum())
    if (cnt > 0).all():
        np.testing.assert_allclose(ece, ref[0], rtol=1e-6)


@pytest.mark.parametrize("shape", [(1200, 784), (37, 5), (4096, 4096)])
def test_snr_kernel_matches_torch_and_numpy(dev, shape):
    rs = np.random.RandomState(5)
    mu = torch.from_numpy(rs.uniform(-0.2, 0.2, shape).astype(np.float32))
    rho = torch.from_numpy(rs.uniform(-5, -4, shape).astype(np.float32))
    mu.view(-1)[3] = 0.0                                    # |mu| = 0: -inf dB, pruned at any threshold
    got = ops.snr_db(mu.to(dev), rho.to(dev)).cpu()
    ref = 10 * torch.log10(torch.abs(mu) / torch.log1p(torch.exp(rho)))          # weight_pruning.py:100
    assert got.view(-1)[3] == -float("inf")
    ok = torch.isfinite(ref)
    np.testing.assert_allclose(got[ok].numpy(), ref[ok].numpy(), rtol=0, atol=3e-5)
    ref64 = O.compute_snr(mu.double().numpy(), np.log(1 + np.exp(rho.double().numpy())))   # :85-87 / :42-43 in numpy
    np.testing.assert_allclose(got[ok].numpy(), ref64[ok.numpy()], rtol=0, atol=3e-5)


@pytest.mark.parametrize("drop", [0.5, 0.25, 0.9])
def test_snr_pruning_of_a_network(dev, drop):
    """posthoc.prune_weights on the drop-in network against the oracle's restatement of weight_pruning.py:89-115 fed
    with the reference's float64 SNR list: same survivors (up to elements within fp32 rounding of the threshold),
    pruned entries left at (0, 0), the state_dict round-trips."""
    import networks
    mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification", mu_init=[-0.2, 0.2],
              rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=False)
    net = networks.BayesianNetwork(mp)
    sd = synth.synth_state_dict(784, 1200, 10, False)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    layers = [tuple(torch.from_numpy(sd[f"{l}.{n}"]) for n in synth.PARAM_NAMES) for l in synth.LAYER_NAMES]
    # the reference's threshold list: named_parameters() order, python floats, numpy float64 (weight_pruning.py:15-40)
    mus = np.concatenate([sd[f"{l}.{n}"].ravel() for l in synth.LAYER_NAMES for n in ("weight_mu", "bias_mu")]).astype(np.float64)
    rhos = np.concatenate([sd[f"{l}.{n}"].ravel() for l in synth.LAYER_NAMES for n in ("weight_rho", "bias_rho")]).astype(np.float64)
    snrs = O.compute_snr(mus, np.log(1 + np.exp(rhos)))
    want, thr_ref = O.prune_weights(layers, snrs, drop)
    net.to(dev)
    dsnr = posthoc.compute_snr(net)
    assert dsnr.numel() == snrs.size
    np.testing.assert_allclose(np.sort(dsnr.cpu().numpy()), np.sort(snrs), rtol=0, atol=3e-5)
    thr = posthoc.prune_weights(net, None, drop)
    assert abs(thr - thr_ref) <= 3e-5
    got = {k: v.cpu() for k, v in net.state_dict().items()}
    total, differ = 0, 0
    for l, w in zip(synth.LAYER_NAMES, want):
        for n, ref in zip(synth.PARAM_NAMES, w):
            g = got[f"{l}.{n}"]
            total += g.numel()
            differ += int((g != ref).sum())
            if "mu" in n:
                assert bool(((g == 0) == (got[f"{l}.{n.replace('mu', 'rho')}"] == 0)).all())
    assert differ <= 4, differ                               # only elements whose SNR rounds across the threshold
    kept = sum(int((got[f"{l}.{n}"] != 0).sum()) for l in synth.LAYER_NAMES for n in ("weight_mu", "bias_mu"))
    assert abs(kept - (1 - drop) * snrs.size) <= 4
    net.load_state_dict(got)                                  # the 12-key state_dict survives pruning
