"""CPU oracle for the Bayes-by-backprop hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, as plain functions over torch CPU tensors, the arithmetic of the
reference's hot path so that the HIP kernels have something to be checked against on a
box where the reference itself cannot travel.  It is NOT part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / the timed CPU baseline.  The product path
(``bayesian-neural-network_amd/``) never imports anything from ``oracle/``.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the real reference
(``/root/reference/networks.py``) in the build container, injects identical epsilon
through the reference's own ``.normal`` attribute seam and stores inputs/outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below against
those vectors (fp32, rel 1e-6).

Each function cites the reference lines whose arithmetic it follows (paths relative to
/root/reference).  The op ORDER inside each expression follows the reference so that
fp32 rounding and CPU cost are the same; the code structure (functional, explicit
epsilon arguments, no nn.Module) is this repository's own.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

C0 = -math.log(math.sqrt(2.0 * math.pi))  # -log sqrt(2 pi), networks.py:46


# --------------------------------------------------------------------------------------
# a1: softplus, naive form (networks.py:39, :118-119)
# --------------------------------------------------------------------------------------
def softplus_naive(rho: torch.Tensor) -> torch.Tensor:
    """sigma = log1p(exp(rho)) with no threshold trick (networks.py:39).

    Overflows to +inf for rho >~ 88.7 and underflows to 0 for rho <~ -104 exactly as the
    reference does (golden G8)."""
    return torch.log1p(torch.exp(rho))


# --------------------------------------------------------------------------------------
# a2: reparameterised sample (networks.py:41-43)
# --------------------------------------------------------------------------------------
def sample_gaussian(mu: torch.Tensor, rho: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """w = mu + sigma * eps (networks.py:43): fp32 multiply, then add."""
    return mu + softplus_naive(rho) * eps


# --------------------------------------------------------------------------------------
# a3: log q(w | mu, rho) (networks.py:45-46)
# --------------------------------------------------------------------------------------
def log_q(w: torch.Tensor, mu: torch.Tensor, rho: torch.Tensor) -> torch.Tensor:
    """sum( c0 - log(sigma) - (w-mu)^2 / (2 sigma^2) ) (networks.py:46).

    sigma is recomputed for each use, as the reference's property does."""
    return (C0 - torch.log(softplus_naive(rho))
            - ((w - mu) ** 2) / (2 * softplus_naive(rho) ** 2)).sum()


# --------------------------------------------------------------------------------------
# a4: Gaussian prior log p(w) (networks.py:67-68 used at :82)
# --------------------------------------------------------------------------------------
def log_p_gauss(w: torch.Tensor, sigma_p: float) -> torch.Tensor:
    """sum of Normal(0, sigma_p).log_prob(w).

    torch.distributions.Normal.log_prob evaluates
    -((v - loc)^2) / (2 var) - log(scale) - log(sqrt(2 pi)); restated in that order."""
    var = sigma_p ** 2
    return (-((w - 0.0) ** 2) / (2 * var) - math.log(sigma_p) - math.log(math.sqrt(2 * math.pi))).sum()


# --------------------------------------------------------------------------------------
# a5: scale-mixture prior (networks.py:24-27)
# --------------------------------------------------------------------------------------
def _normal_logpdf(w: torch.Tensor, scale: float) -> torch.Tensor:
    return -((w - 0.0) ** 2) / (2 * scale ** 2) - math.log(scale) - math.log(math.sqrt(2 * math.pi))


def log_p_mixture(w: torch.Tensor, pi: float, sigma1: float, sigma2: float) -> torch.Tensor:
    """sum log( pi N(w;0,s1) + (1-pi) N(w;0,s2) ), NOT log-sum-exp stabilised
    (networks.py:25-27)."""
    p1 = torch.exp(_normal_logpdf(w, sigma1))
    p2 = torch.exp(_normal_logpdf(w, sigma2))
    return torch.log(pi * p1 + (1 - pi) * p2).sum()


@dataclass
class Prior:
    """Prior description: Gaussian (sigma_p) or scale mixture (pi, sigma1, sigma2).

    ``from_init`` follows the constructor convention of networks.py:61-68: a mixture takes
    ``[pi, log sigma1, log sigma2]``, a Gaussian takes ``[sigma_p]``."""
    mixture: bool
    sigma_p: float = 1.0
    pi: float = 0.5
    sigma1: float = 1.0
    sigma2: float = 1.0

    @staticmethod
    def from_init(prior_init: Sequence[float], mixture: bool) -> "Prior":
        if mixture:
            assert len(prior_init) == 3
            return Prior(True, pi=float(prior_init[0]), sigma1=math.exp(prior_init[1]),
                         sigma2=math.exp(prior_init[2]))
        assert len(prior_init) == 1
        return Prior(False, sigma_p=float(prior_init[0]))

    def log_prob(self, w: torch.Tensor) -> torch.Tensor:
        if self.mixture:
            return log_p_mixture(w, self.pi, self.sigma1, self.sigma2)
        return log_p_gauss(w, self.sigma_p)


# --------------------------------------------------------------------------------------
# a6: BayesianLinear.forward (networks.py:73-88)
# --------------------------------------------------------------------------------------
def bbb_linear(x: torch.Tensor, w_mu: torch.Tensor, w_rho: torch.Tensor, b_mu: torch.Tensor,
               b_rho: torch.Tensor, eps_w: Optional[torch.Tensor], eps_b: Optional[torch.Tensor],
               prior: Prior, want_log_probs: bool = True):
    """One weight-sampling layer.  ``eps_w``/``eps_b`` None means w = mu (the
    eval, sample=False row of the mode table, networks.py:78-79).

    Returns (y, log_prior, log_variational_posterior); the two scalars are Python int 0
    when ``want_log_probs`` is False (networks.py:86)."""
    if eps_w is not None:
        w = sample_gaussian(w_mu, w_rho, eps_w)
        b = sample_gaussian(b_mu, b_rho, eps_b)
    else:
        w, b = w_mu, b_mu
    if want_log_probs:
        lp = prior.log_prob(w).sum() + prior.log_prob(b).sum()       # networks.py:82
        lq = log_q(w, w_mu, w_rho).sum() + log_q(b, b_mu, b_rho).sum()  # networks.py:83
    else:
        lp, lq = 0, 0
    y = torch.nn.functional.linear(x, w, b)                          # networks.py:88
    return y, lp, lq


def bbb_linear_bf16(x: torch.Tensor, w_mu: torch.Tensor, w_rho: torch.Tensor, b_mu: torch.Tensor,
                    b_rho: torch.Tensor, eps_w: torch.Tensor, eps_b: torch.Tensor, prior: Prior):
    """bbb_linear with the device's bf16 rounding points and nothing else changed: the matmul operands (x, the sampled w)
    rounded to bf16 (RNE), fp32 accumulation, fp32 bias, the statistics from the un-rounded fp32 weights -- what the bf16
    kernels of a layer are pinned against (their own fp32 summation order is all that differs)."""
    w = sample_gaussian(w_mu, w_rho, eps_w)
    b = sample_gaussian(b_mu, b_rho, eps_b)
    lp = prior.log_prob(w).sum() + prior.log_prob(b).sum()
    lq = log_q(w, w_mu, w_rho).sum() + log_q(b, b_mu, b_rho).sum()
    rnd = lambda t: t.to(torch.bfloat16).to(torch.float32)
    return torch.nn.functional.linear(rnd(x), rnd(w), b), lp, lq


# --------------------------------------------------------------------------------------
# a7: BayesianLinearLR.forward + compute_kl_cost (networks.py:109-138)
# --------------------------------------------------------------------------------------
def kl_closed_form(q_mu: torch.Tensor, q_sigma: torch.Tensor, p_mu: float, p_sigma: float) -> torch.Tensor:
    """0.5 * sum( 2 log(sp/sq) - 1 + (sq/sp)^2 + ((mp - mq)/sp)^2 ) (networks.py:113)."""
    return 0.5 * (2 * torch.log(p_sigma / q_sigma) - 1 + (q_sigma / p_sigma).pow(2)
                  + ((p_mu - q_mu) / p_sigma).pow(2)).sum()


def lr_linear(x: torch.Tensor, w_mu: torch.Tensor, w_rho: torch.Tensor, b_mu: torch.Tensor,
              b_rho: torch.Tensor, eps_act: torch.Tensor, eps_b: torch.Tensor, sigma_p: float,
              want_kl: bool = True):
    """Local-reparameterisation layer; weights are [in, out] (networks.py:95-96).

    Returns (activation, weight_kl, bias_kl); the KLs are None when ``want_kl`` is False
    (the reference then leaves its attributes stale, networks.py:133)."""
    w_sigma = softplus_naive(w_rho)
    b_sigma = softplus_naive(b_rho)
    act_mu = torch.mm(x, w_mu)                                         # networks.py:120
    act_sigma = torch.sqrt(torch.mm(x.pow(2), w_sigma.pow(2)))         # networks.py:121
    act_w = act_mu + act_sigma * eps_act                               # networks.py:125
    act_b = b_mu + b_sigma * eps_b                                     # networks.py:126
    act = act_w + act_b.unsqueeze(0).expand(x.shape[0], -1)            # networks.py:128
    if want_kl:
        return act, kl_closed_form(w_mu, w_sigma, 0.0, sigma_p), kl_closed_form(b_mu, b_sigma, 0.0, sigma_p)
    return act, None, None


def lr_linear_mean(x: torch.Tensor, w_mu: torch.Tensor, b_mu: torch.Tensor) -> torch.Tensor:
    """Deterministic LR forward x.M + b_mu: the evident intent of networks.py:131 (the
    reference raises AttributeError there)."""
    return torch.mm(x, w_mu) + b_mu


# --------------------------------------------------------------------------------------
# a10: NLL (networks.py:183-190)
# --------------------------------------------------------------------------------------
def nll(outputs: torch.Tensor, target: torch.Tensor, mode: str, sigma: float = 1.0) -> torch.Tensor:
    if mode == "regression":
        # -Normal(outputs, sigma).log_prob(target).sum()   (networks.py:185)
        var = sigma ** 2
        lp = -((target - outputs) ** 2) / (2 * var) - math.log(sigma) - math.log(math.sqrt(2 * math.pi))
        return -lp.sum()
    if mode == "classification":
        return torch.nn.functional.cross_entropy(outputs, target, reduction="sum")  # networks.py:187
    raise Exception("Training mode must be either 'regression' or 'classification'")


# --------------------------------------------------------------------------------------
# a8-a12: three-layer network, MC loop, ELBO assembly (networks.py:166-225)
# --------------------------------------------------------------------------------------
@dataclass
class NetParams:
    """Twelve fp32 tensors in state_dict order l1..l3 x (weight_mu, weight_rho, bias_mu,
    bias_rho); BBB weights are [out,in], LR weights [in,out] (networks.py:53 vs :95)."""
    layers: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]]
    mode: str
    input_shape: int
    local_reparam: bool
    prior: Prior

    @staticmethod
    def from_state_dict(sd, mode: str, input_shape: int, local_reparam: bool, prior: Prior) -> "NetParams":
        layers = []
        for l in ("l1", "l2", "l3"):
            layers.append(tuple(torch.as_tensor(np.asarray(sd[f"{l}.{n}"]), dtype=torch.float32)
                                if not torch.is_tensor(sd[f"{l}.{n}"]) else sd[f"{l}.{n}"].detach().float().cpu()
                                for n in ("weight_mu", "weight_rho", "bias_mu", "bias_rho")))
        return NetParams(layers, mode, input_shape, local_reparam, prior)

    def eps_shapes(self, batch: int) -> List[Tuple[int, ...]]:
        """Epsilon draw order for ONE forward: per layer the weight-shaped
        (BBB) or activation-shaped (LR) draw, then the bias draw."""
        out = []
        for (wm, _, bm, _) in self.layers:
            if self.local_reparam:
                out += [(batch, wm.shape[1]), tuple(bm.shape)]
            else:
                out += [tuple(wm.shape), tuple(bm.shape)]
        return out

    def n_stochastic(self) -> int:
        return sum(wm.numel() + bm.numel() for (wm, _, bm, _) in self.layers)


def draw_eps(p: NetParams, batch: int, gen: Optional[torch.Generator] = None) -> List[torch.Tensor]:
    """Standard-normal draws in the reference's order on the CPU generator
    (Normal(0,1).sample(shape) == torch.randn(shape); networks.py:42, :123-124)."""
    return [torch.randn(s, generator=gen) for s in p.eps_shapes(batch)]


def network_forward(p: NetParams, x: torch.Tensor, eps: Optional[Sequence[torch.Tensor]],
                    want_log_probs: bool = True):
    """BayesianNetwork.forward (networks.py:166-172) for one MC sample.

    Returns (logits, log_prior, log_q) for BBB or (logits, kl, None) for LR: the sums over
    the three layers of networks.py:174-181."""
    if p.mode == "classification":
        x = x.view(-1, p.input_shape)                                  # networks.py:168
    a, b = 0, 0
    for i, (wm, wr, bm, br) in enumerate(p.layers):
        ew = eps[2 * i] if eps is not None else None
        eb = eps[2 * i + 1] if eps is not None else None
        if p.local_reparam:
            if eps is None:
                x = lr_linear_mean(x, wm, bm)
            else:
                x, kw, kb = lr_linear(x, wm, wr, bm, br, ew, eb, p.prior.sigma_p, want_log_probs)
                if want_log_probs:
                    a = a + (kw + kb)                                  # networks.py:136, :181
        else:
            x, lp, lq = bbb_linear(x, wm, wr, bm, br, ew, eb, p.prior, want_log_probs)
            a = a + lp
            b = b + lq
        if i < 2:
            x = torch.relu(x)                                          # networks.py:161,163
    return (x, a, None) if p.local_reparam else (x, a, b)


# --------------------------------------------------------------------------------------
# The SAME network with the rounding points of the device's bf16 math mode (include/bnn_hip.h, BNN_MATH_BF16).
# NOT a reference function: the reference computes in fp32 throughout (network_forward above is its restatement).  This
# variant exists so that the bf16 kernels can be pinned TIGHTLY (accumulation order is then the only difference, ~1e-6),
# independently of how large the bf16-vs-fp32 deviation itself is; that deviation is a property of the precision choice
# and is measured on this CPU pair (network_forward vs network_forward_bf16), see tests/test_oracle_golden.py.
# Rounding points (all round-to-nearest-even to bf16, everything else fp32):
#   BBB: the matmul operands -- bf16(x) (the input batch and every hidden activation AFTER ReLU) and bf16(w) with
#        w = mu + softplus(rho) * eps in fp32; products exact, fp32 accumulation, the fp32 bias sample added in fp32;
#        log p / log q from the UN-rounded fp32 w.
#   LR:  bf16(x) . bf16(M) and x2 . bf16(sigma^2) with x2 = bf16(bf16(x)^2), the square of the rounded activation, rounded
#        (one definition for every kernel form, include/bnn_hip.h bnn_math); sqrt, activation noise, bias and the KL in fp32.
# --------------------------------------------------------------------------------------
def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def network_forward_bf16(p: NetParams, x: torch.Tensor, eps: Sequence[torch.Tensor]):
    """(logits, log_prior | kl, log_q | None) like network_forward, with the device's bf16 rounding points."""
    if p.mode == "classification":
        x = x.view(-1, p.input_shape)
    a, b = 0, 0
    for i, (wm, wr, bm, br) in enumerate(p.layers):
        ew, eb = eps[2 * i], eps[2 * i + 1]
        if p.local_reparam:
            w_sigma, b_sigma = softplus_naive(wr), softplus_naive(br)
            xr = _bf16(x)
            x2 = _bf16(xr * xr)
            act_mu = torch.mm(xr, _bf16(wm))
            act_sigma = torch.sqrt(torch.mm(x2, _bf16(w_sigma * w_sigma)))
            x = act_mu + act_sigma * ew + (bm + b_sigma * eb).unsqueeze(0)
            a = a + (kl_closed_form(wm, w_sigma, 0.0, p.prior.sigma_p) + kl_closed_form(bm, b_sigma, 0.0, p.prior.sigma_p))
        else:
            w = sample_gaussian(wm, wr, ew)
            bb = sample_gaussian(bm, br, eb)
            a = a + (p.prior.log_prob(w).sum() + p.prior.log_prob(bb).sum())
            b = b + (log_q(w, wm, wr).sum() + log_q(bb, bm, br).sum())
            x = torch.nn.functional.linear(_bf16(x), _bf16(w), bb)
        if i < 2:
            x = torch.relu(x)
    return (x, a, None) if p.local_reparam else (x, a, b)


def _split_bf16(t: torch.Tensor):
    """The split pair of the device's BNN_MATH_BF16X3 mode (include/bnn_hip.h): hi = bf16(v), lo = bf16(v - hi), both RNE."""
    hi = _bf16(t)
    return hi, _bf16(t - hi)


def linear_bf16x3(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """F.linear(x, w, b) (networks.py:88) with the rounding points of the split-bf16 mode: both operands as (hi, lo) pairs,
    x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in fp32 (the lo . lo term is dropped), fp32 bias."""
    xh, xl = _split_bf16(x)
    wh, wl = _split_bf16(w)
    return torch.nn.functional.linear(xh, wh) + torch.nn.functional.linear(xh, wl) + torch.nn.functional.linear(xl, wh) + b


def network_forward_bf16x3(p: NetParams, x: torch.Tensor, eps: Sequence[torch.Tensor]):
    """(logits, log_prior | kl, log_q | None) like network_forward, with the rounding points of the device's split-bf16 math
    mode and nothing else changed (statistics from the un-rounded fp32 parameters).  Local reparameterisation (the stacked-
    minibatch path of engine.GraphedElbo): hidden layers take the MEAN product x . M in split-bf16 and the VARIANCE product on
    bf16 operands -- bf16 of the fp32 squares of the layer's fp32 input, bf16(sigma^2) --; the output layer is exact fp32."""
    if p.mode == "classification":
        x = x.view(-1, p.input_shape)
    a, b = 0, 0
    if p.local_reparam:
        last = len(p.layers) - 1
        for i, (wm, wr, bm, br) in enumerate(p.layers):
            ew, eb = eps[2 * i], eps[2 * i + 1]
            w_sigma, b_sigma = softplus_naive(wr), softplus_naive(br)
            if i < last:
                act_mu = linear_bf16x3(x, wm.t(), torch.zeros(wm.shape[1]))
                act_sigma = torch.sqrt(torch.mm(_bf16(x * x), _bf16(w_sigma * w_sigma)))
            else:
                act_mu = torch.mm(x, wm)
                act_sigma = torch.sqrt(torch.mm(x * x, w_sigma * w_sigma))
            x = act_mu + act_sigma * ew + (bm + b_sigma * eb).unsqueeze(0)
            a = a + (kl_closed_form(wm, w_sigma, 0.0, p.prior.sigma_p) + kl_closed_form(bm, b_sigma, 0.0, p.prior.sigma_p))
            if i < last:
                x = torch.relu(x)
        return x, a, None
    for i, (wm, wr, bm, br) in enumerate(p.layers):
        w = sample_gaussian(wm, wr, eps[2 * i])
        bb = sample_gaussian(bm, br, eps[2 * i + 1])
        a = a + (p.prior.log_prob(w).sum() + p.prior.log_prob(bb).sum())
        b = b + (log_q(w, wm, wr).sum() + log_q(bb, bm, br).sum())
        x = linear_bf16x3(x, w, bb)
        if i < 2:
            x = torch.relu(x)
    return x, a, b


def sample_elbo(p: NetParams, x: torch.Tensor, target: torch.Tensor, beta: float, samples: int,
                sigma: float = 1.0, eps: Optional[Sequence[Sequence[torch.Tensor]]] = None,
                gen: Optional[torch.Generator] = None):
    """networks.py:192-209.  Returns (loss[1], mean log p [], mean log q [], nll[1])."""
    assert not p.local_reparam
    lps = torch.zeros(samples)
    lqs = torch.zeros(samples)
    nl = torch.zeros(1)
    batch = x.shape[0]
    for i in range(samples):
        e = eps[i] if eps is not None else draw_eps(p, batch, gen)
        out, lp, lq = network_forward(p, x, e)
        lps[i] = lp
        lqs[i] = lq
        nl += nll(out, target, p.mode, sigma)
    log_prior = beta * lps.mean()                                      # networks.py:205
    log_var_post = beta * lqs.mean()                                   # networks.py:206
    nl = nl / samples                                                  # networks.py:207
    loss = log_var_post - log_prior + nl                               # networks.py:208
    return loss, lps.mean(), lqs.mean(), nl


def sample_elbo_lr(p: NetParams, x: torch.Tensor, target: torch.Tensor, beta: float, samples: int,
                   sigma: float = 1.0, eps: Optional[Sequence[Sequence[torch.Tensor]]] = None,
                   gen: Optional[torch.Generator] = None):
    """networks.py:211-225.  Returns (loss[1], mean kl [], nll[1])."""
    assert p.local_reparam
    kls = torch.zeros(samples)
    nl = torch.zeros(1)
    batch = x.shape[0]
    for i in range(samples):
        e = eps[i] if eps is not None else draw_eps(p, batch, gen)
        out, kl, _ = network_forward(p, x, e)
        kls[i] = kl
        nl += nll(out, target, p.mode, sigma)
    kl_cost = beta * kls.mean()                                        # networks.py:222
    nl = nl / samples
    loss = kl_cost + nl                                                # networks.py:224
    return loss, kls.mean(), nl


def beta_schedule(num_batches: int, idx: int) -> float:
    """beta = 2^(M-(idx+1)) / (2^M - 1): exact Python integers, then true division
    (classification/class_task.py:70, regression/reg_task.py:63)."""
    return 2 ** (num_batches - (idx + 1)) / (2 ** num_batches - 1)


# --------------------------------------------------------------------------------------
# CPU restatement of the DEVICE epsilon generator (not a reference function: the reference
# draws eps from torch's CPU generator).  Philox4x32 (Salmon et al., SC'11) with PHILOX_ROUNDS rounds + Box-Muller
# with the counter->element map of include/bnn_hip.h (BNN_EPS_MAP_VERSION 2: 7 rounds; version 1 ran 10).  Used by tests to check that
# the kernels' on-chip eps is what the header says it is.
# --------------------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


PHILOX_ROUNDS = 7      # include/bnn_hip.h: BNN_PHILOX_ROUNDS


def philox4x32(c0, c1, c2, c3, k0, k1, rounds: int = PHILOX_ROUNDS):
    """Vectorised Philox4x32-`rounds`.  All arguments broadcastable uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(v, dtype=np.uint32) for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = c0.astype(np.uint64) * _PHILOX_M0
            p1 = c2.astype(np.uint64) * _PHILOX_M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """The 10-round form of the same code: what the Random123 known-answer vectors exist for."""
    return philox4x32(c0, c1, c2, c3, k0, k1, rounds=10)


def _u01(r: np.ndarray) -> np.ndarray:
    """uint32 -> (0,1]: the device's v_cvt_f32_u32 (round to nearest even) followed by one
    fp32 fma(r, 2^-32, 2^-33): the product+sum is exact in float64, then rounded once."""
    return (r.astype(np.float32).astype(np.float64) * 2.0 ** -32 + 2.0 ** -33).astype(np.float32)


def box_muller(r0, r1):
    u1 = _u01(r0).astype(np.float64)
    u2 = _u01(r1).astype(np.float64)
    rad = np.sqrt(-2.0 * np.log(u1))
    return (rad * np.cos(2 * np.pi * u2)).astype(np.float32), (rad * np.sin(2 * np.pi * u2)).astype(np.float32)


def philox_normal(seed: int, tensor_id: int, sample: int, rows: int, cols: int) -> np.ndarray:
    """eps[rows, cols] for global MC sample ``sample`` of tensor ``tensor_id``.

    Element (r, c) lives in group g = r * ceil(cols/4) + (c >> 2), slot c & 3.
    Counter = (g, sample, tensor_id, 0), key = (seed_lo, seed_hi).  The four outputs map
    to slots 0..3 as (BoxMuller(r0,r1).cos, .sin, BoxMuller(r2,r3).cos, .sin)."""
    gpr = (cols + 3) // 4
    g = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(gpr)
         + np.arange(gpr, dtype=np.uint64)[None, :]).astype(np.uint32)
    r0, r1, r2, r3 = philox4x32(g, np.uint32(sample), np.uint32(tensor_id), np.uint32(0),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    n0, n1 = box_muller(r0, r1)
    n2, n3 = box_muller(r2, r3)
    out = np.stack([n0, n1, n2, n3], axis=-1).reshape(rows, gpr * 4)
    return np.ascontiguousarray(out[:, :cols])


def tensor_id(layer: int, kind: int) -> int:
    """kind: 0 weight eps (BBB), 1 bias eps, 2 activation eps (LR)."""
    return layer * 4 + kind


def philox_eps_for_network(p: NetParams, batch: int, seed: int, sample: int) -> List[torch.Tensor]:
    """The eps list (reference draw order) that the device generator produces for global
    MC sample ``sample``."""
    out = []
    for li, (wm, _, bm, _) in enumerate(p.layers):
        if p.local_reparam:
            out.append(torch.from_numpy(philox_normal(seed, tensor_id(li, 2), sample, batch, wm.shape[1])))
        else:
            out.append(torch.from_numpy(philox_normal(seed, tensor_id(li, 0), sample, wm.shape[0], wm.shape[1])))
        out.append(torch.from_numpy(philox_normal(seed, tensor_id(li, 1), sample, 1, bm.shape[0]))[0])
    return out


# --------------------------------------------------------------------------------------
# F3 / F4 restatements.  compute_ece.py and weight_pruning.py cannot be imported in the build container (both
# import seaborn at module scope; nothing can be installed), so these follow the SOURCE TEXT line by line and are
# NOT pinned by reference-recorded vectors: parity UNPINNED for these two rows.
# --------------------------------------------------------------------------------------
def get_one_hot(targets: np.ndarray, nb_classes: int) -> np.ndarray:
    """compute_ece.py:59-61."""
    res = np.eye(nb_classes)[np.array(targets).reshape(-1)]
    return res.reshape(list(targets.shape) + [nb_classes])


def ece_reference(probs: np.ndarray, labels: np.ndarray, bin_step: float = 0.1, num_classes: int = 10):
    """ECELoss.forward, compute_ece.py:22-57, statement by statement.  Returns (ece, bin_centers[have_data], bin_acc).
    Like the reference it is only meaningful when every bin holds data: an empty bin makes np.mean NaN and the
    compressed bin_acc array misaligned with the loop index (IndexError in the reference; NaN here)."""
    pred_class = np.argmax(probs, axis=1)                                          # :23
    expanded_preds = np.reshape(probs, -1)                                         # :26
    pred_class_OH = np.reshape(get_one_hot(pred_class, num_classes), -1)           # :27
    target_class_OH = np.reshape(get_one_hot(labels, num_classes), -1)             # :28
    correct_vec = (target_class_OH * (pred_class_OH == target_class_OH)).astype(int)   # :29
    bins = np.arange(0, 1.1, bin_step)                                             # :32
    bin_idxs = np.digitize(expanded_preds, bins, right=True)                       # :33
    bin_idxs = bin_idxs - 1                                                        # :34
    bin_centers = bins[1:] - bin_step / 2                                          # :36
    bin_counts = np.ones(len(bin_centers))                                         # :37
    bin_corrects = np.zeros(len(bin_centers))                                      # :38
    bin_confidence = np.zeros(len(bin_centers))                                    # :39
    for nbin in range(len(bin_centers)):                                           # :45-48
        bin_counts[nbin] = np.sum((bin_idxs == nbin).astype(int))
        bin_corrects[nbin] = np.sum(correct_vec[bin_idxs == nbin])
        with np.errstate(invalid="ignore"), __import__("warnings").catch_warnings():
            __import__("warnings").simplefilter("ignore")
            bin_confidence[nbin] = np.mean(expanded_preds[bin_idxs == nbin])
    have_data = bin_counts > 0                                                     # :50
    bin_acc = bin_corrects[have_data] / bin_counts[have_data]                      # :51
    ece = 0                                                                        # :53-55
    try:
        for i in range(len(bin_confidence)):
            ece += np.absolute(bin_confidence[i] - bin_acc[i]) * bin_counts[i] / np.sum(bin_counts)
    except IndexError:               # an empty bin: the reference's loop walks off the compressed bin_acc array here
        ece = float("nan")
    return ece, bin_centers[have_data], bin_acc, (bin_counts, bin_corrects, bin_confidence)


def compute_snr(mu, sigma):
    """weight_pruning.py:85-87: signal-to-noise ratio in decibels."""
    return 10 * np.log10(abs(mu) / sigma)


def prune_weights(layers: Sequence[Sequence[torch.Tensor]], snrs, drop_percentage: float = 0.5):
    """weight_pruning.py:89-115 on a list of (weight_mu, weight_rho, bias_mu, bias_rho) fp32 tensors: returns the
    pruned copies and the threshold.  Same torch ops, same order."""
    snr_threshold = np.percentile(snrs, 100 * drop_percentage)                     # :92
    out = []
    for weight_mus, weight_rhos, bias_mus, bias_rhos in layers:
        weight_sigmas = torch.log1p(torch.exp(weight_rhos))                        # :98
        bias_sigmas = torch.log1p(torch.exp(bias_rhos))                            # :99
        s = 10 * torch.log10(torch.abs(weight_mus) / weight_sigmas)                # :102
        mask = (s > snr_threshold).long()                                          # :104-105
        wm, wr = weight_mus * mask, weight_rhos * mask                             # :106-107
        s = 10 * torch.log10(torch.abs(bias_mus) / bias_sigmas)                    # :110
        mask = (s > snr_threshold).long()
        out.append((wm, wr, bias_mus * mask, bias_rhos * mask))                    # :113-114
    return out, float(snr_threshold)
