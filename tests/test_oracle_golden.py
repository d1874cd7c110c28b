"""The CPU oracle (oracle/bnn_oracle.py) against vectors recorded from the REAL reference
(oracle/make_golden.py).  This is what pins the oracle: SURVEY §8(c) G1-G9."""
import math

import numpy as np
import pytest
import torch

from oracle import bnn_oracle as O
from bnn_hip import synth

RTOL = 1e-6   # fp32 restatement vs fp32 reference, same op order
torch.set_num_threads(1)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, rtol=RTOL, atol=0.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def run_bbb(c):
    prior = O.Prior.from_init(list(c["prior_init"]), bool(c["mixture"]))
    train, sample, clp = bool(c["train"]), bool(c["sample"]), bool(c["clp"])
    do_sample = train or sample
    want = train or clp
    return O.bbb_linear(t(c["x"]), t(c["weight_mu"]), t(c["weight_rho"]), t(c["bias_mu"]), t(c["bias_rho"]),
                        t(c["eps_w"]) if do_sample else None, t(c["eps_b"]) if do_sample else None, prior, want)


def check_bbb(c):
    y, lp, lq = run_bbb(c)
    close(y.numpy(), c["y"], rtol=2e-6, atol=1e-6)
    if int(c["log_prior_is_int"]):
        assert lp == 0 and lq == 0
    else:
        close(float(lp), c["log_prior"])
        close(float(lq), c["log_q"])


def run_lr(c):
    return O.lr_linear(t(c["x"]), t(c["weight_mu"]), t(c["weight_rho"]), t(c["bias_mu"]), t(c["bias_rho"]),
                       t(c["eps_act"]), t(c["eps_b"]), float(c["sigma_p"]), bool(c["train"]) or bool(c["clp"]))


def test_g1_gaussian_prior(g_layers):
    for name in g_layers.cases("G1"):
        check_bbb(g_layers.case(name))


def test_g2_mixture_prior(g_layers):
    for name in g_layers.cases("G2"):
        check_bbb(g_layers.case(name))


def test_g3_mode_truth_table(g_layers):
    names = g_layers.cases("G3")
    assert len(names) == 8
    n_int = 0
    for name in names:
        c = g_layers.case(name)
        check_bbb(c)
        n_int += int(c["log_prior_is_int"])
    assert n_int == 2          # eval & not calculate_log_probs, sample in {F,T}


def test_g4_lr_layer(g_layers):
    for name in g_layers.cases("G4"):
        c = g_layers.case(name)
        y, kw, kb = run_lr(c)
        close(y.numpy(), c["y"], rtol=2e-6, atol=1e-6)
        if bool(c["train"]) or bool(c["clp"]):
            close(float(kw), c["weight_kl"])
            close(float(kb), c["bias_kl"])
            close(float(kw + kb), c["kl"])
        else:                                  # stale attributes: reference left its init value 0
            assert kw is None and float(c["kl"]) == 0.0


def test_g7_odd_shapes(g_layers):
    names = g_layers.cases("G7")
    assert len(names) == 21
    for name in names:
        c = g_layers.case(name)
        if "lr" in name.split("/")[-1]:
            y, kw, kb = run_lr(c)
            close(y.numpy(), c["y"], rtol=1e-5, atol=1e-5)
            close(float(kw + kb), c["kl"])
        else:
            y, lp, lq = run_bbb(c)
            close(y.numpy(), c["y"], rtol=1e-5, atol=1e-4)   # K up to 1200: fp32 sum-order noise
            close(float(lp), c["log_prior"])
            close(float(lq), c["log_q"])


def test_g8_rho_extremes(g_layers):
    c = g_layers.case("G8")
    sig = O.softplus_naive(t(c["rho"]))
    close(sig.numpy(), c["sigma"])
    assert math.isinf(float(sig[-1])) and float(sig[-1]) > 0          # rho = 89 overflows, as in the reference
    w = O.sample_gaussian(t(c["mu"]), t(c["rho"]), t(c["eps"]))
    np.testing.assert_allclose(w.numpy()[:6], c["w"][:6], rtol=RTOL)
    lq = O.log_q(w[:6], t(c["mu"])[:6], t(c["rho"])[:6])
    close(float(lq), c["log_q_sum_finite"], rtol=2e-6)


def build_net(g, lr, B, mixture=False, dims=(1, 50, 1), mode="regression"):
    sd = synth.synth_state_dict(dims[0], dims[1], dims[2], lr)
    prior = O.Prior.from_init([0.5, 0.0, -6.0], True) if mixture else O.Prior.from_init([1.0], False)
    return O.NetParams.from_state_dict(sd, mode, dims[0], lr, prior)


def net_eps(p, B, S):
    return [[t(a) for a in synth.synth_eps(p.eps_shapes(B), s)] for s in range(S)]


@pytest.mark.parametrize("variant", ["bbb", "lr", "mix"])
def test_g5_c1_elbo_and_grads(g_c1, variant):
    lr = variant == "lr"
    n_checked = 0
    for Bname in g_c1.cases(f"G5/{variant}"):
        B = int(Bname.split("/")[-1][1:])
        for Sname in g_c1.cases(Bname):
            S = int(Sname.split("/")[-1][1:])
            for bname in g_c1.cases(Sname):
                c = g_c1.case(bname)
                p = build_net(g_c1, lr, B, mixture=(variant == "mix"))
                for lay in p.layers:
                    for q in lay:
                        q.requires_grad_(True)
                x, y = synth.synth_batch("regression", B, 1, 1)
                fn = O.sample_elbo_lr if lr else O.sample_elbo
                tup = fn(p, t(x), t(y), float(c["beta"]), S, 0.1, eps=net_eps(p, B, S))
                assert [v.dim() for v in tup] == list(c["t_shapes"])
                for i, v in enumerate(tup):
                    close(v.detach().numpy(), c[f"t{i}"], rtol=2e-6)
                if "grad/l1.weight_mu" in c:
                    tup[0].backward()
                    names = [f"l{li+1}.{n}" for li in range(3) for n in synth.PARAM_NAMES]
                    flat = [q for lay in p.layers for q in lay]
                    for n, q in zip(names, flat):
                        gref = c[f"grad/{n}"]
                        np.testing.assert_allclose(q.grad.numpy(), gref, rtol=2e-5,
                                                   atol=2e-6 * float(np.abs(gref).max()))
                    n_checked += 1
    assert n_checked >= 2


@pytest.mark.parametrize("variant", ["bbb", "mix", "lr"])
def test_g6_c2_full_size(g_c2, variant):
    torch.set_num_threads(8)
    try:
        lr = variant == "lr"
        p = build_net(g_c2, lr, 128, mixture=(variant == "mix"), dims=(784, 1200, 10), mode="classification")
        x, y = synth.synth_batch("classification", 128, 784, 10)
        xt, yt = t(x), t(y)
        c = g_c2.case(f"G6/{variant}/S1")
        eps = net_eps(p, 128, 2)
        logits, a, b = O.network_forward(p, xt, eps[0])
        close(logits[:2].numpy(), c["logits_s0_rows01"], rtol=1e-4, atol=2e-4)   # K=1200 fp32 sum order (threads)
        close(float(O.nll(logits, yt, "classification")), c["nll_s0"], rtol=1e-5)
        for S in (1, 2):
            c = g_c2.case(f"G6/{variant}/S{S}")
            fn = O.sample_elbo_lr if lr else O.sample_elbo
            tup = fn(p, xt, yt, 0.5, S, eps=eps[:S])
            for i, v in enumerate(tup):
                close(v.numpy(), c[f"t{i}"], rtol=3e-6)
        # gradients at full C2 size (S = 2): the oracle's autograd against the reference's, on every 997th element
        # and the L2 norm of all 12 tensors
        c = g_c2.case(f"G6/{variant}/S2")
        for lay in p.layers:
            for q in lay:
                q.requires_grad_(True)
        tup = (O.sample_elbo_lr if lr else O.sample_elbo)(p, xt, yt, 0.5, 2, eps=eps[:2])
        tup[0].backward()
        names = [f"l{li+1}.{n}" for li in range(3) for n in synth.PARAM_NAMES]
        for n, q in zip(names, [q for lay in p.layers for q in lay]):
            g = q.grad.double().flatten()
            sub = c[f"grad_sub/{n}"]
            np.testing.assert_allclose(g[::997].numpy(), sub, rtol=2e-5, atol=2e-6 * float(np.abs(sub).max()))
            close(float(g.pow(2).sum().sqrt()), c[f"grad_l2/{n}"], rtol=1e-5)
    finally:
        torch.set_num_threads(1)


def test_g9_beta_schedule(g_beta):
    c = g_beta.case("G9")
    M = int(c["M"])
    for i, idx in enumerate(c["idx"]):
        b = O.beta_schedule(M, int(idx))
        assert b == float(c["beta"][i])
        got = float((b * torch.tensor(7.4e6, dtype=torch.float32)).item())
        assert got == float(c["beta_times_7p4e6_f32"][i])
    assert float(c["beta_times_7p4e6_f32"][4]) == 0.0      # idx 149: complexity term is exactly 0 in fp32


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors): the round function, the
    multipliers and the key schedule of the generator -- the device and this restatement run the SAME code for
    BNN_PHILOX_ROUNDS = 7 rounds (eps map version 2), for which the tests below check the statistics."""
    r = O.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(v) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = O.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(v) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = O.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(v) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert O.PHILOX_ROUNDS == 7
    assert [int(v) for v in O.philox4x32(0, 0, 0, 0, 0, 0)] != [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(v) for v in O.philox4x32(0, 0, 0, 0, 0, 0, rounds=10)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


def _bits(words):
    """uint32 arrays [4][n] -> bit matrix [n, 128]"""
    return np.concatenate([((w[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.uint8) for w in words], axis=1)


def test_philox7_avalanche_and_bit_balance():
    """The 7-round generator on the counters the map actually uses (consecutive groups, small sample and tensor ids,
    one key): every output bit is balanced, and flipping ANY single counter or key bit flips each of the 128 output bits
    with probability 1/2 (strict avalanche) -- sequential counters are the hard case for a weak round count (with 4
    rounds this test fails by a wide margin, asserted below so that the test is known to have teeth)."""
    n = 4096
    rs = np.random.RandomState(11)
    ctr = [np.arange(n, dtype=np.uint32) * np.uint32(3) + np.uint32(17), rs.randint(0, 64, n).astype(np.uint32),
           rs.randint(0, 12, n).astype(np.uint32), np.zeros(n, np.uint32)]
    key = (np.uint32(2026), np.uint32(0))

    def worst(rounds):
        base = _bits(O.philox4x32(*ctr, *key, rounds=rounds))
        bal = np.abs(base.mean(0) - 0.5).max()
        dev = 0.0
        for word in range(3):                                      # the counter words the map varies
            for bit in (0, 1, 2, 5, 9, 13, 21, 31):
                c2 = [c.copy() for c in ctr]
                c2[word] = c2[word] ^ np.uint32(1 << bit)
                flips = (_bits(O.philox4x32(*c2, *key, rounds=rounds)) ^ base).mean(0)     # per output bit
                dev = max(dev, float(np.abs(flips - 0.5).max()))
        for bit in (0, 7, 31):                                     # key (seed) bits
            flips = (_bits(O.philox4x32(*ctr, np.uint32(2026 ^ (1 << bit)), key[1], rounds=rounds)) ^ base).mean(0)
            dev = max(dev, float(np.abs(flips - 0.5).max()))
        return bal, dev

    bal7, dev7 = worst(7)
    bal10, dev10 = worst(10)
    # 4096 Bernoulli(1/2) trials: sigma = 0.0078; the maximum over ~3500 (bit, flip) pairs sits near 4 sigma
    assert bal7 < 0.04 and dev7 < 0.045, (bal7, dev7)
    assert abs(dev7 - dev10) < 0.015                               # indistinguishable from the 10-round form here
    _, dev4 = worst(4)
    assert dev4 > 0.2                                              # a short generator does NOT pass: the test has teeth


def test_philox7_normals_against_the_normal_distribution():
    """eps of map version 2: moments, Kolmogorov-Smirnov distance to N(0,1), independence of neighbouring elements,
    samples, tensors and seeds."""
    from scipy import stats
    e = O.philox_normal(2026, O.tensor_id(1, 0), 3, 600, 1200).astype(np.float64)     # 720 000 draws
    n = e.size
    assert abs(e.mean()) < 4.0 / np.sqrt(n) and abs(e.var() - 1.0) < 4.0 * np.sqrt(2.0 / n)
    assert abs(stats.skew(e.ravel())) < 4.0 * np.sqrt(6.0 / n) and abs(stats.kurtosis(e.ravel())) < 4.0 * np.sqrt(24.0 / n)
    d, _ = stats.kstest(e.ravel()[::4], "norm")                   # 180 000 draws: the 1 % critical value is 1.63 / sqrt(n)
    assert d < 1.63 / np.sqrt(n / 4)
    flat = e.ravel()
    for lag in (1, 2, 3, 4, 1200):                                  # within a Philox group, across groups, across rows
        assert abs(np.corrcoef(flat[:-lag], flat[lag:])[0, 1]) < 4.0 / np.sqrt(n)
    for other in (O.philox_normal(2026, O.tensor_id(1, 0), 4, 600, 1200), O.philox_normal(2026, O.tensor_id(2, 0), 3, 600, 1200),
                  O.philox_normal(2027, O.tensor_id(1, 0), 3, 600, 1200)):
        assert abs(np.corrcoef(flat, other.ravel())[0, 1]) < 4.0 / np.sqrt(n)
    assert np.abs(e).max() < 6.7                                    # u in (0,1] with 2^-33 granularity: |eps| <= 6.66


def test_philox_normal_moments():
    e = O.philox_normal(2026, O.tensor_id(1, 0), 3, 300, 1201)
    assert e.shape == (300, 1201) and e.dtype == np.float32
    assert abs(float(e.mean())) < 5e-3 and abs(float(e.std()) - 1.0) < 5e-3
    # sample/tensor/seed separate the streams; the map does not depend on how rows are tiled
    e2 = O.philox_normal(2026, O.tensor_id(1, 0), 4, 300, 1201)
    assert abs(float(np.corrcoef(e.ravel(), e2.ravel())[0, 1])) < 5e-3
    sub = O.philox_normal(2026, O.tensor_id(1, 0), 3, 7, 1201)
    np.testing.assert_array_equal(sub, e[:7])


def test_ece_restatement_known_answer():
    """F3 oracle (parity UNPINNED: compute_ece.py needs seaborn to import): a hand-computed case.  Two classes, bins of
    width 0.5: rows (0.9, 0.1) label 0, (0.6, 0.4) label 1, (0.2, 0.8) label 1.  Bin (0, 0.5] holds {0.1, 0.4, 0.2}
    with no "correct" entry; bin (0.5, 1] holds {0.9, 0.6, 0.8} of which 0.9 and 0.8 are correct argmax entries.
    ECE = |0.7/3 - 0| * 3/6 + |2.3/3 - 2/3| * 3/6 = 0.11666.. + 0.05 = 0.16666.."""
    probs = np.asarray([[0.9, 0.1], [0.6, 0.4], [0.2, 0.8]], np.float32)
    labels = np.asarray([0, 1, 1], np.int64)
    ece, centers, acc, (cnt, cor, conf) = O.ece_reference(probs, labels, bin_step=0.5, num_classes=2)
    assert list(cnt) == [3, 3] and list(cor) == [0, 2]
    close(conf, [np.float32(0.1) / 3 + np.float32(0.4) / 3 + np.float32(0.2) / 3, (0.9 + 0.6 + 0.8) / 3], rtol=1e-6)
    close(ece, (0.7 / 3) * 0.5 + abs(2.3 / 3 - 2 / 3) * 0.5, rtol=1e-6)
    close(centers, [0.25, 0.75]); close(acc, [0.0, 2 / 3])


def test_snr_pruning_restatement():
    """F4 oracle (parity UNPINNED): half of the weights go, the survivors are exactly those above the median SNR, a
    pruned weight is left at rho = 0 (weight_pruning.py:106-107 multiplies rho by the mask too)."""
    rs = np.random.RandomState(3)
    layers = [tuple(t(rs.uniform(lo, hi, sh).astype(np.float32)) for lo, hi, sh in
                    ((-0.2, 0.2, (6, 5)), (-5, -4, (6, 5)), (-0.2, 0.2, (6,)), (-5, -4, (6,))))]
    mus = np.concatenate([layers[0][0].numpy().ravel(), layers[0][2].numpy().ravel()]).astype(np.float64)
    sig = np.log(1 + np.exp(np.concatenate([layers[0][1].numpy().ravel(), layers[0][3].numpy().ravel()]).astype(np.float64)))
    snrs = O.compute_snr(mus, sig)
    (wm, wr, bm, br), thr = O.prune_weights(layers, snrs, 0.5)[0][0], O.prune_weights(layers, snrs, 0.5)[1]
    kept = int((wm != 0).sum() + (bm != 0).sum())
    assert kept == 18 and thr == float(np.percentile(snrs, 50))
    assert bool(((wm == 0) == (wr == 0)).all()) and bool(((bm == 0) == (br == 0)).all())


def test_bf16_rounding_points_deviation_on_cpu():
    """What the bf16 math mode costs in accuracy, measured on the CPU alone: the oracle with the device's bf16 rounding
    points (matmul operands rounded to bf16, fp32 accumulate, fp32 statistics) against the reference's fp32 arithmetic,
    C2 network, three epsilon draws per variant.  The complexity terms are untouched (they never see a rounded value);
    the NLL moves by up to a few 1e-3 relative -- the bound the GPU parity tests hold the bf16 kernels to against the
    fp32 oracle (BF16_NLL_RTOL 5.5e-3, logits 1.8e-2 of scale) -- which is why the ELBO of the bf16 mode meets rtol 1e-4
    only while beta * complexity >> NLL."""
    worst_nll, worst_lg = 0.0, 0.0
    for lr in (False, True):
        sd = synth.synth_state_dict(784, 1200, 10, lr)
        p = O.NetParams.from_state_dict(sd, "classification", 784, lr, O.Prior.from_init([1.0], False))
        x, y = synth.synth_batch("classification", 128, 784, 10)
        for g in range(3):
            eps = O.philox_eps_for_network(p, 128, 42, 7 + g)
            o32 = O.network_forward(p, torch.from_numpy(x), eps)
            o16 = O.network_forward_bf16(p, torch.from_numpy(x), eps)
            n32 = float(O.nll(o32[0], torch.from_numpy(y), "classification"))
            n16 = float(O.nll(o16[0], torch.from_numpy(y), "classification"))
            assert abs(float(o16[1]) - float(o32[1])) <= 1e-6 * abs(float(o32[1]))          # log p | KL: fp32 either way
            worst_nll = max(worst_nll, abs(n16 - n32) / n32)
            worst_lg = max(worst_lg, float((o16[0] - o32[0]).abs().max()) / float(o32[0].abs().max()))
    assert 1e-5 < worst_nll <= 5.5e-3 and worst_lg <= 1.8e-2, (worst_nll, worst_lg)


def test_split_bf16_rounding_points_meet_the_fp32_tolerance_on_cpu():
    """What the split-bf16 math mode (BNN_MATH_BF16X3: every matmul operand as hi = bf16(v), lo = bf16(v - hi); products as
    x_hi w_hi + x_hi w_lo + x_lo w_hi in fp32) costs in accuracy, measured on the CPU alone: O.network_forward_bf16x3 against
    the reference's fp32 arithmetic, C2 network, three epsilon draws.  The NLL stays within 1e-4 relative -- the north
    star's ELBO tolerance at every beta, the pure-NLL tail included -- and the logits within 1e-4 of their scale: a
    property of the rounding points, two hundred times tighter than plain bf16 operands
    (test_bf16_rounding_points_deviation_on_cpu)."""
    sd = synth.synth_state_dict(784, 1200, 10, False)
    p = O.NetParams.from_state_dict(sd, "classification", 784, False, O.Prior.from_init([1.0], False))
    x, y = synth.synth_batch("classification", 128, 784, 10)
    worst_nll, worst_lg = 0.0, 0.0
    for g in range(3):
        eps = O.philox_eps_for_network(p, 128, 42, 7 + g)
        o32 = O.network_forward(p, torch.from_numpy(x), eps)
        o3 = O.network_forward_bf16x3(p, torch.from_numpy(x), eps)
        n32 = float(O.nll(o32[0], torch.from_numpy(y), "classification"))
        n3 = float(O.nll(o3[0], torch.from_numpy(y), "classification"))
        assert abs(float(o3[1]) - float(o32[1])) <= 1e-6 * abs(float(o32[1])) and abs(float(o3[2]) - float(o32[2])) <= 1e-6 * abs(float(o32[2]))
        worst_nll = max(worst_nll, abs(n3 - n32) / n32)
        worst_lg = max(worst_lg, float((o3[0] - o32[0]).abs().max()) / float(o32[0].abs().max()))
    assert worst_nll <= 3e-5 and worst_lg <= 1e-4, (worst_nll, worst_lg)
    # the pair itself: hi + lo carries v to 2^-16 relative (two RNE roundings of 8 significant bits each)
    v = torch.from_numpy(np.random.RandomState(0).standard_normal(4096).astype(np.float32))
    hi, lo = O._split_bf16(v)
    assert float(((hi + lo) - v).abs().max() / v.abs().max()) <= 2.0 ** -16
