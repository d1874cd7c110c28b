"""Drop-in `networks` module for tennisonliu/bayesian-neural-network on AMD MI355X.

Same nn.Module surface as the reference's networks.py (class names, constructor
arguments, parameter names/shapes, forward signatures, side-effect attributes, return
tuples — SURVEY §8(b1)), so `main.py`, the task wrappers, `load_model_utils.py`,
`logger_utils.py` and `weight_pruning.py` run unchanged when this directory precedes the
reference's on sys.path.  The arithmetic of BayesianLinear / BayesianLinearLR /
BayesianNetwork runs in hand-written gfx950 kernels (bnn_hip/libbnn_hip.so); there is no
CPU fallback: a forward on CPU tensors raises.

Differences that are deliberate:
  * the serial MC loop of sample_elbo / sample_elbo_lr (reference networks.py:199, :217)
    is one launch per layer over all samples;
  * epsilon comes from the on-chip Philox stream unless a layer's `.normal` attribute was
    replaced or BNN_HIP_EPS=host (then the reference's CPU draw order is reproduced);
  * BayesianLinearLR in eval mode without `sample` returns x.M + bias_mu (the reference
    raises AttributeError at networks.py:131);
  * the LR closed-form KL is computed once per ELBO evaluation, not once per MC sample.
"""
import math

import torch
from torch import nn

from config import *  # noqa: F401,F403  (DEVICE, RegConfig, RLConfig, ClassConfig)
from bnn_hip import _lib as _L
from bnn_hip import engine as _engine
from bnn_hip.functional import BBBLinearFn as _BBBLinearFn
from bnn_hip.functional import LayerCall as _LayerCall
from bnn_hip.functional import LRLinearFn as _LRLinearFn
from bnn_hip.ops import PriorSpec as _PriorSpec
from bnn_hip.runtime import state as _state
from bnn_hip.runtime import take_samples as _take_samples

_L.load()   # fail at import, loudly, if libbnn_hip.so is missing: there is no fallback path


class _StandardNormal:
    """Default value of the `.normal` seam (reference networks.py:35, :100).  While a layer
    still holds this object its epsilon is generated on chip; replacing the attribute with
    anything exposing `.sample(size)` switches that layer to injected epsilon."""

    def sample(self, size):
        return torch.randn(tuple(size))


class ScaleMixtureGaussian:
    """pi N(0, sigma1) + (1 - pi) N(0, sigma2) (reference networks.py:14-27)."""

    def __init__(self, pi, sigma1, sigma2):
        self.pi, self.sigma1, self.sigma2 = pi, sigma1, sigma2
        self.gaussian1 = torch.distributions.Normal(0, sigma1)
        self.gaussian2 = torch.distributions.Normal(0, sigma2)

    def log_prob(self, input):
        # helper surface only; the layer's forward evaluates this inside the fused kernel
        mix = self.pi * self.gaussian1.log_prob(input).exp() + (1 - self.pi) * self.gaussian2.log_prob(input).exp()
        return mix.log().sum()


class GaussianNode:
    """View of one (mu, rho) parameter pair (reference networks.py:29-46).  Holds references
    to the layer's nn.Parameters, so optimiser updates and .to() are seen."""

    def __init__(self, mu, rho):
        self.mu, self.rho = mu, rho
        self.normal = _StandardNormal()

    @property
    def sigma(self):
        return torch.log1p(torch.exp(self.rho))

    def sample(self):
        epsilon = self.normal.sample(self.rho.size()).to(self.rho.device)
        return self.mu + self.sigma * epsilon

    def log_prob(self, input):
        s = self.sigma
        return (-math.log(math.sqrt(2 * math.pi)) - torch.log(s) - (input - self.mu) ** 2 / (2 * s ** 2)).sum()


def _uniform_param(lo_hi, *shape):
    return nn.Parameter(torch.empty(*shape).uniform_(*lo_hi))


class BayesianLinear(nn.Module):
    """Weight-sampling Bayesian FC layer (reference networks.py:48-88) on kernel K1."""

    def __init__(self, in_features, out_features, mu_init, rho_init, prior_init, mixture_prior=True):
        super().__init__()
        self.weight_mu = _uniform_param(mu_init, out_features, in_features)
        self.weight_rho = _uniform_param(rho_init, out_features, in_features)
        self.weight = GaussianNode(self.weight_mu, self.weight_rho)
        self.bias_mu = _uniform_param(mu_init, out_features)
        self.bias_rho = _uniform_param(rho_init, out_features)
        self.bias = GaussianNode(self.bias_mu, self.bias_rho)
        self._prior_spec = _PriorSpec.from_init(prior_init, bool(mixture_prior))
        if mixture_prior:
            self.weight_prior = ScaleMixtureGaussian(prior_init[0], math.exp(prior_init[1]), math.exp(prior_init[2]))
            self.bias_prior = ScaleMixtureGaussian(prior_init[0], math.exp(prior_init[1]), math.exp(prior_init[2]))
        else:
            self.weight_prior = torch.distributions.Normal(0, prior_init[0])
            self.bias_prior = torch.distributions.Normal(0, prior_init[0])
        self.log_prior = 0
        self.log_variational_posterior = 0
        self._layer_id = 0

    # ---- epsilon seam
    def _eps_stubbed(self):
        return not (isinstance(self.weight.normal, _StandardNormal) and isinstance(self.bias.normal, _StandardNormal))

    def _draw_eps(self, w_shape, b_shape):
        return [self.weight.normal.sample(torch.Size(w_shape)), self.bias.normal.sample(torch.Size(b_shape))]

    def forward(self, input, sample=False, calculate_log_probs=False):
        do_sample = self.training or sample
        want = self.training or calculate_log_probs
        injected = None
        if do_sample:
            injected = _engine.collect_injected([_engine.LayerSpec(self, self._layer_id, False, False)],
                                                input.shape[0], 1, input.device)
        eps_mode = _L.EPS_ZERO if not do_sample else (_L.EPS_MEMORY if injected is not None else _L.EPS_PHILOX)
        call = _LayerCall(n_samples=1, prior=self._prior_spec, math_mode=_state.math, relu=False, eps_mode=eps_mode,
                          seed=_state.seed, layer_id=self._layer_id,
                          sample_offset=_take_samples(1) if eps_mode == _L.EPS_PHILOX else 0, want_stats=want)
        e_w, e_b = (injected[0], injected[1]) if injected is not None else (None, None)
        y, lp, lq = _BBBLinearFn.apply(input, self.weight_mu, self.weight_rho, self.bias_mu, self.bias_rho,
                                       e_w, e_b, call)
        if want:
            self.log_prior, self.log_variational_posterior = lp[0], lq[0]
        else:
            self.log_prior, self.log_variational_posterior = 0, 0
        return y[0]


class BayesianLinearLR(nn.Module):
    """Local-reparameterisation Bayesian FC layer (reference networks.py:90-138) on kernel
    K3; weights are [in, out]."""

    def __init__(self, in_features, out_features, mu_init, rho_init, prior_init, mixture_prior=False):
        super().__init__()
        self.weight_mu = _uniform_param(mu_init, in_features, out_features)
        self.weight_rho = _uniform_param(rho_init, in_features, out_features)
        self.bias_mu = _uniform_param(mu_init, out_features)
        self.bias_rho = _uniform_param(rho_init, out_features)
        self.normal = _StandardNormal()
        assert len(prior_init) == 1, "Gaussian Prior requires one value in prior initialisation"
        self._prior_spec = _PriorSpec.from_init(prior_init, False)
        self.weight_prior = [0, prior_init[0]]
        self.bias_prior = [0, prior_init[0]]
        self.weight_kl_cost = 0
        self.bias_kl_cost = 0
        self.kl_cost = 0
        self._layer_id = 0

    def _eps_stubbed(self):
        return not isinstance(self.normal, _StandardNormal)

    def _draw_eps(self, act_shape, b_shape):
        return [self.normal.sample(torch.Size(act_shape)), self.normal.sample(torch.Size(b_shape))]

    def compute_kl_cost(self, p_params, q_params):
        """Closed-form KL between two Gaussians (reference networks.py:109-114) on kernel K2.
        q_params = [q_mu, q_sigma] is accepted for signature compatibility; when q_sigma is
        softplus of one of this layer's rho tensors the kernel reads rho directly."""
        from bnn_hip import ops as _ops
        [p_mu, p_sigma] = p_params
        [q_mu, q_sigma] = q_params
        for mu_p, rho_p in ((self.weight_mu, self.weight_rho), (self.bias_mu, self.bias_rho)):
            if q_mu is mu_p and p_mu == 0:
                return _ops.gauss_kl(mu_p.detach(), rho_p.detach(), float(p_sigma))[0]
        return 0.5 * (2 * torch.log(p_sigma / q_sigma) - 1 + (q_sigma / p_sigma).pow(2)
                      + ((p_mu - q_mu) / p_sigma).pow(2)).sum()

    def forward(self, input, sample=False, calculate_log_probs=False):
        do_sample = self.training or sample
        want = self.training or calculate_log_probs
        injected = None
        if do_sample:
            injected = _engine.collect_injected([_engine.LayerSpec(self, self._layer_id, True, False)],
                                                input.shape[0], 1, input.device)
        eps_mode = _L.EPS_ZERO if not do_sample else (_L.EPS_MEMORY if injected is not None else _L.EPS_PHILOX)
        call = _LayerCall(n_samples=1, prior=self._prior_spec, math_mode=_state.math, relu=False, eps_mode=eps_mode,
                          seed=_state.seed, layer_id=self._layer_id,
                          sample_offset=_take_samples(1) if eps_mode == _L.EPS_PHILOX else 0, want_stats=want)
        e_a, e_b = (injected[0], injected[1]) if injected is not None else (None, None)
        y, kl3 = _LRLinearFn.apply(input, self.weight_mu, self.weight_rho, self.bias_mu, self.bias_rho, e_a, e_b, call)
        if want:   # otherwise the attributes stay as they were (reference networks.py:133)
            self.kl_cost, self.weight_kl_cost, self.bias_kl_cost = kl3[0], kl3[1], kl3[2]
        return y[0]


class BayesianNetwork(nn.Module):
    """Three stochastic layers + ReLU and the ELBO assembly (reference networks.py:140-225)."""

    def __init__(self, model_params):
        super().__init__()
        self.input_shape = model_params['input_shape']
        self.classes = model_params['classes']
        self.batch_size = model_params['batch_size']
        self.hidden_units = model_params['hidden_units']
        self.mode = model_params['mode']
        self.mu_init = model_params['mu_init']
        self.rho_init = model_params['rho_init']
        self.prior_init = model_params['prior_init']
        self.mixture_prior = model_params['mixture_prior']
        self.local_reparam = model_params['local_reparam']
        layer = BayesianLinearLR if self.local_reparam else BayesianLinear
        dims = [(self.input_shape, self.hidden_units), (self.hidden_units, self.hidden_units),
                (self.hidden_units, self.classes)]
        built = [layer(i, o, self.mu_init, self.rho_init, self.prior_init, self.mixture_prior) for i, o in dims]
        for idx, l in enumerate(built):
            l._layer_id = idx
        # registered in the reference's order (networks.py:160-164): children() yields l1, l1_act, l2, l2_act, l3
        self.l1 = built[0]
        self.l1_act = nn.ReLU()
        self.l2 = built[1]
        self.l2_act = nn.ReLU()
        self.l3 = built[2]

    def _specs(self):
        lr = bool(self.local_reparam)
        return [_engine.LayerSpec(self.l1, 0, lr, True), _engine.LayerSpec(self.l2, 1, lr, True),
                _engine.LayerSpec(self.l3, 2, lr, False)]

    def _flat(self, x):
        if self.mode == 'classification':
            x = x.view(-1, self.input_shape)
        return x

    def forward(self, x, sample=False):
        """One forward (reference networks.py:166-172): three launches, ReLU fused."""
        x = self._flat(x)
        specs = self._specs()
        training = self.training
        do_sample = training or sample
        differentiable = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        injected = _engine.collect_injected(specs, x.shape[0], 1, x.device) if do_sample else None
        first = _take_samples(1) if (do_sample and injected is None) else 0
        out, stats = _engine.run_layers(specs, x, 1, first, want_stats=training, sample=do_sample, injected=injected,
                                        differentiable=differentiable or training)
        for sp, st in zip(specs, stats):
            if self.local_reparam:
                if training:
                    sp.m.kl_cost, sp.m.weight_kl_cost, sp.m.bias_kl_cost = st[0], st[1], st[2]
            else:
                sp.m.log_prior, sp.m.log_variational_posterior = (st[0][0], st[1][0]) if training else (0, 0)
        return out[0]

    def forward_mc(self, x, samples):
        """Extension (not in the reference): the outputs of `samples` stochastic forward passes,
        [samples, batch, classes], with the samples batched into one launch per layer — what
        regression/reg_task.py:76-83 collects by calling net(x, sample=True) in a Python loop."""
        return _engine.mc_forward(self._specs(), self._flat(x), int(samples))

    def predict_mc(self, x, samples):
        """Extension (not in the reference): classification/class_task.py:81-87 in one call —
        (preds[batch], probs[batch, classes]) with probs = mean over `samples` stochastic passes of
        softmax(net(x, sample=True)).  Same eps order as that loop when eps is injected."""
        return _engine.mc_predict(self._specs(), self._flat(x), int(samples))

    def predictor(self, x, samples, capture=True):
        """Extension (not in the reference): predict_mc for this minibatch shape as a captured evaluation --
        `p = net.predictor(x, samples)`, then `p.x.copy_(next_minibatch); preds, probs = p.replay()` per minibatch (static
        buffers, one hipGraph replay, fresh epsilon each time: bnn_hip.engine.GraphedPredict).  `capture="calls"`: a recorded
        launch list instead of a hipGraph (a few us less per replay at up to ~16 samples)."""
        return _engine.GraphedPredict(self, x, int(samples), capture=capture)

    def elbo_many(self, inputs, targets, samples, sigma=1.):
        """Extension (not in the reference): the forward-only ELBO terms of G independent minibatches -- inputs
        [G, batch, ...], targets [G, batch] -- in one launch per layer instead of G sample_elbo calls under
        no_grad (the reference walks minibatches one at a time, classification/class_task.py:89-103).  Returns
        float32 [G, 4]: per minibatch (sum_s log p | sum_s KL, sum_s log q | 0, sum_s nll, samples); divide by
        `samples` for the means sample_elbo / sample_elbo_lr return (networks.py:205-208, :222-224)."""
        if self.mode not in ('regression', 'classification'):
            raise Exception("Training mode must be either 'regression' or 'classification'")
        return _engine.elbo_many(self, inputs, targets, int(samples), sigma=float(sigma))

    def log_prior(self):
        return self.l1.log_prior + self.l2.log_prior + self.l3.log_prior

    def log_variational_posterior(self):
        return self.l1.log_variational_posterior + self.l2.log_variational_posterior + self.l3.log_variational_posterior

    def kl_cost(self):
        return self.l1.kl_cost + self.l2.kl_cost + self.l3.kl_cost

    def get_nll(self, outputs, target, sigma=1.):
        if self.mode not in ('regression', 'classification'):
            raise Exception("Training mode must be either 'regression' or 'classification'")
        from bnn_hip.functional import NLLFn as _NLLFn
        return _NLLFn.apply(outputs.unsqueeze(0), target, self.mode, float(sigma))[0]

    def sample_elbo(self, input, target, beta, samples, sigma=1.):
        ''' Sample ELBO for BNN w/o Local Reparameterisation '''
        assert self.local_reparam == False, 'sample_elbo() method returns loss for BNNs without local reparameterisation, alternatively use sample_elbo_lr()'
        if self.mode not in ('regression', 'classification'):
            raise Exception("Training mode must be either 'regression' or 'classification'")
        slp, slq, snll, n = _engine.elbo_terms(self._specs(), self._flat(input), target, samples, mode=self.mode,
                                               sigma=sigma, local_reparam=False)
        log_prior_mean, log_q_mean = slp / n, slq / n
        negative_log_likelihood = (snll / n).reshape(1)
        loss = beta * log_q_mean - beta * log_prior_mean + negative_log_likelihood
        return loss, log_prior_mean, log_q_mean, negative_log_likelihood

    def sample_elbo_lr(self, input, target, beta, samples, sigma=1.):
        ''' Sample ELBO for BNN w/ Local Reparameterisation '''
        assert self.local_reparam == True, 'sample_elbo_lr() method returns loss for BNNs with local reparameterisation, alternatively use sample_elbo()'
        if self.mode not in ('regression', 'classification'):
            raise Exception("Training mode must be either 'regression' or 'classification'")
        skl, _, snll, n = _engine.elbo_terms(self._specs(), self._flat(input), target, samples, mode=self.mode,
                                             sigma=sigma, local_reparam=True)
        kl_mean = skl / n
        negative_log_likelihood = (snll / n).reshape(1)
        loss = beta * kl_mean + negative_log_likelihood
        return loss, kl_mean, negative_log_likelihood


class _PlainMLP(nn.Module):
    """Deterministic baselines (reference networks.py:227-285): stock nn.Linear stacks, out
    of the hot path; kept so `from networks import MLP, MLP_Dropout` keeps working."""
    _p_drop = None

    def __init__(self, model_params):
        super().__init__()
        self.input_shape = model_params['input_shape']
        self.classes = model_params['classes']
        self.batch_size = model_params['batch_size']
        self.hidden_units = model_params['hidden_units']
        self.mode = model_params['mode']
        widths = [self.input_shape, self.hidden_units, self.hidden_units]
        mods = []
        for a, b in zip(widths[:-1], widths[1:]):
            mods += [nn.Linear(a, b), nn.ReLU()] + ([nn.Dropout(self._p_drop)] if self._p_drop else [])
        self.net = nn.Sequential(*mods, nn.Linear(self.hidden_units, self.classes))

    def forward(self, x):
        if self.mode == 'classification':
            assert len(x.shape) == 4, "Input dimensions incorrect, expected shape = (batch_size, sample, x_dim[0], x_dim[1])"
            x = x.view(-1, self.input_shape)
        else:
            assert len(x.shape) == 2, "Input dimensions incorrect, expected shape = (batch_size, sample,...)"
        return self.net(x)


class MLP(_PlainMLP):
    pass


class MLP_Dropout(_PlainMLP):
    _p_drop = 0.5

    def enable_dropout(self):
        ''' Enable the dropout layers during test-time '''
        for m in self.modules():
            if m.__class__.__name__.startswith('Dropout'):
                m.train()
