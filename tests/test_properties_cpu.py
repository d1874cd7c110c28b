"""Property tests (hypothesis) of the host logic and of the oracle's building blocks: no GPU.
They complement the golden-vector tests: the vectors pin values, these pin the structure the GPU
parity tests lean on (sample sharding is a partition, the Philox map does not depend on the shape
it is asked for, the ELBO pieces obey their identities)."""
import math

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from bnn_hip import engine
from oracle import bnn_oracle as O


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 5000), st.integers(1, 64))
def test_shard_range_partitions_the_samples(n, world):
    blocks = [engine.shard_range(n, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
    for (f0, c0), (f1, _c1) in zip(blocks, blocks[1:]):
        assert f1 == f0 + c0                                     # contiguous, in rank order
    counts = [c for _, c in blocks]
    assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 63 - 1), st.integers(0, 11), st.integers(0, 2 ** 31), st.integers(1, 9), st.integers(1, 40),
       st.integers(1, 9))
def test_philox_map_is_independent_of_the_requested_shape(seed, tid, sample, rows, cols, extra_rows):
    """eps[row, col] depends on (seed, tensor, sample, row, cols) only: asking for more rows returns the
    same leading rows (what makes results independent of tiling, launch shape and the number of GPUs)."""
    a = O.philox_normal(seed, tid, sample, rows, cols)
    b = O.philox_normal(seed, tid, sample, rows + extra_rows, cols)
    assert a.shape == (rows, cols) and a.dtype == np.float32
    assert np.array_equal(a, b[:rows])
    assert np.all(np.isfinite(a)) and float(np.abs(a).max()) < 7.0   # Box-Muller on (0,1] uniforms of 32 bits
    c = O.philox_normal(seed, tid, sample + 1, rows, cols)
    assert not np.array_equal(a, c) or a.size < 2                   # another sample: another subsequence


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 500))
def test_beta_schedule_is_a_partition_of_unity(m):
    """class_task.py:70: beta_i = 2^(M-i) / (2^M - 1), i = 1..M, sums to 1 and halves every minibatch."""
    betas = [O.beta_schedule(m, i) for i in range(m)]
    assert abs(sum(betas) - 1.0) < 1e-12
    for b0, b1 in zip(betas, betas[1:]):
        assert b1 == pytest.approx(b0 / 2, rel=1e-12)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.floats(-3, 3), min_size=1, max_size=16), st.lists(st.floats(-6, 2), min_size=1, max_size=16),
       st.floats(0.05, 3.0))
def test_closed_form_kl_is_nonnegative_and_zero_at_the_prior(mus, rhos, sigma_p):
    n = min(len(mus), len(rhos))
    mu = torch.tensor(mus[:n], dtype=torch.float64)
    sig = O.softplus_naive(torch.tensor(rhos[:n], dtype=torch.float64))
    kl = O.kl_closed_form(mu, sig, 0.0, sigma_p)
    assert float(kl) >= -1e-9
    zero = O.kl_closed_form(torch.zeros(n, dtype=torch.float64), torch.full((n,), sigma_p, dtype=torch.float64), 0.0, sigma_p)
    assert abs(float(zero)) < 1e-9


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 6), st.integers(1, 5), st.integers(0, 2 ** 31 - 1))
def test_log_q_of_a_reparameterised_draw_depends_on_eps_and_sigma_only(rows, cols, seed):
    """networks.py:46 at w = mu + sigma * eps: log q = sum(c0 - log sigma - eps^2 / 2), the identity the
    kernels use (they never form (w - mu) / sigma)."""
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(rows, cols, generator=g, dtype=torch.float64)
    rho = torch.randn(rows, cols, generator=g, dtype=torch.float64) - 2.0
    eps = torch.randn(rows, cols, generator=g, dtype=torch.float64)
    w = O.sample_gaussian(mu, rho, eps)
    sig = O.softplus_naive(rho)
    want = (-0.5 * math.log(2 * math.pi) - torch.log(sig) - 0.5 * eps * eps).sum()
    assert float(O.log_q(w, mu, rho)) == pytest.approx(float(want), rel=1e-9, abs=1e-9)


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 4), st.integers(2, 12), st.integers(2, 6), st.integers(0, 2 ** 31 - 1))
def test_sample_elbo_is_the_mean_over_its_samples(samples, batch, hidden, seed):
    """networks.py:199-208: the S-sample ELBO terms are the means of S one-sample evaluations on the same
    eps (the fact the engine uses to batch samples per launch and to shard them over ranks)."""
    from bnn_hip import synth
    sd = synth.synth_state_dict(3, hidden, 2, False)
    p = O.NetParams.from_state_dict(sd, "classification", 3, False, O.Prior.from_init([1.0], False))
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, generator=g)
    y = torch.randint(0, 2, (batch,), generator=g)
    eps = [[torch.randn(s_, generator=g) for s_ in p.eps_shapes(batch)] for _ in range(samples)]
    loss, lp, lq, nll = O.sample_elbo(p, x, y, 0.3, samples, eps=eps)
    singles = [O.sample_elbo(p, x, y, 0.3, 1, eps=[e]) for e in eps]
    assert float(lp) == pytest.approx(float(sum(s_[1] for s_ in singles)) / samples, rel=1e-5)
    assert float(lq) == pytest.approx(float(sum(s_[2] for s_ in singles)) / samples, rel=1e-5)
    assert float(nll) == pytest.approx(float(sum(s_[3] for s_ in singles)) / samples, rel=1e-5)
    assert float(loss) == pytest.approx(0.3 * float(lq) - 0.3 * float(lp) + float(nll), rel=1e-5)
