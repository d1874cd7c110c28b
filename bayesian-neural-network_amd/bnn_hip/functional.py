"""autograd bridges for the hot path.

Forward = the HIP kernels (K1/K3/K4).  Backward = the HIP backward kernels (F1: bnn_bbb_linear_bwd,
bnn_lr_linear_bwd, bnn_nll_bwd) implementing the closed-form gradients of SURVEY A.5 on eps REGENERATED from the
Philox counter map (nothing weight-sized is kept alive between forward and backward except in the identical-eps
parity mode).  There is no other backward: activations that the kernels cannot take (anything but fp32 on the
differentiable path) raise.  The same formulas evaluated with tensor ops live in tests/tensor_op_grads.py as the
cross-check the GPU tests compare the kernels with (golden G5 / G6 pin both against the real reference's autograd).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L
from . import ops
from .ops import PriorSpec


@dataclass(frozen=True)
class LayerCall:
    """Static (non-tensor) description of one layer launch."""
    n_samples: int
    prior: PriorSpec
    math_mode: int
    relu: bool
    eps_mode: int
    seed: int
    layer_id: int
    sample_offset: int
    want_stats: bool
    y_dtype: torch.dtype = torch.float32
    sample_counter: Optional[torch.Tensor] = None   # device int32[1] added to sample_offset at run time


def _need_f32(name, *ts):
    for t_ in ts:
        if t_ is not None and t_.dtype != torch.float32:
            raise ops.BnnHipError(f"{name}: the HIP backward kernels take fp32 activations (got {t_.dtype}); the differentiable "
                                  f"path keeps hidden activations in fp32 -- cast the input batch to float32")


class BBBLinearFn(torch.autograd.Function):
    """(x, w_mu, w_rho, b_mu, b_rho, eps_w|None, eps_b|None) -> (y[S,B,N], log_prior[S], log_q[S])."""

    @staticmethod
    def forward(ctx, x, w_mu, w_rho, b_mu, b_rho, eps_w, eps_b, call: LayerCall):
        out = ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, n_samples=call.n_samples, prior=call.prior,
                                 math_mode=call.math_mode, relu=call.relu, y_dtype=call.y_dtype,
                                 eps_mode=call.eps_mode, eps_w=eps_w, eps_b=eps_b, seed=call.seed,
                                 layer_id=call.layer_id, sample_offset=call.sample_offset,
                                 sample_counter=call.sample_counter,
                                 want_stats=call.want_stats, want_scalars=call.want_stats)
        y = out["y"]
        ctx.call = call
        ctx.save_for_backward(x, w_mu, w_rho, b_mu, b_rho, eps_w, eps_b, y if call.relu else None)
        if call.want_stats:
            return y, out["log_prior"], out["log_q"]
        z1 = torch.zeros(call.n_samples, dtype=torch.float32, device=y.device)
        z2 = torch.zeros(call.n_samples, dtype=torch.float32, device=y.device)
        ctx.mark_non_differentiable(z1, z2)
        return y, z1, z2

    @staticmethod
    def backward(ctx, gy, glp, glq):
        call: LayerCall = ctx.call
        x, w_mu, w_rho, b_mu, b_rho, eps_w, eps_b, y = ctx.saved_tensors
        S = call.n_samples
        N, K = w_mu.shape
        dev = w_mu.device
        _need_f32("BBBLinearFn.backward", x, y)
        # F1: hand-written backward kernels, eps regenerated on chip
        g_wmu, g_wrho, g_bmu, g_brho, gx = ops.bbb_linear_bwd(
            x, gy.float(), y, w_mu, w_rho, b_mu, b_rho, n_samples=S, prior=call.prior, math_mode=call.math_mode,
            relu=call.relu, eps_mode=call.eps_mode, eps_w=eps_w, eps_b=eps_b, seed=call.seed,
            layer_id=call.layer_id, sample_offset=call.sample_offset, sample_counter=call.sample_counter,
            g_log_prior=glp if call.want_stats else None, g_log_q=glq if call.want_stats else None,
            want_gx=ctx.needs_input_grad[0])
        if gx is not None and x.dim() == 2:
            gx = gx.sum(0)
        return gx, g_wmu, g_wrho, g_bmu, g_brho, None, None, None


class LRLinearFn(torch.autograd.Function):
    """(x, M[in,out], rho, b_mu, b_rho, eps_act|None, eps_b|None) -> (y[S,B,N], kl3[3])."""

    @staticmethod
    def forward(ctx, x, w_mu, w_rho, b_mu, b_rho, eps_act, eps_b, call: LayerCall):
        out = ops.lr_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, n_samples=call.n_samples,
                                sigma_p=call.prior.sigma_p, math_mode=call.math_mode, relu=call.relu,
                                y_dtype=call.y_dtype, eps_mode=call.eps_mode, eps_act=eps_act, eps_b=eps_b,
                                seed=call.seed, layer_id=call.layer_id, sample_offset=call.sample_offset,
                                sample_counter=call.sample_counter,
                                want_kl=call.want_stats, want_scalars=call.want_stats, want_v=True)
        y = out["y"]
        ctx.call = call
        ctx.save_for_backward(x, w_mu, w_rho, b_mu, b_rho, eps_act, eps_b, y if call.relu else None, out["v"])
        if call.want_stats:
            return y, out["kl3"]
        z = torch.zeros(3, dtype=torch.float32, device=y.device)
        ctx.mark_non_differentiable(z)
        return y, z

    @staticmethod
    def backward(ctx, gy, gkl3):
        call: LayerCall = ctx.call
        x, w_mu, w_rho, b_mu, b_rho, eps_act, eps_b, y, v = ctx.saved_tensors
        S = call.n_samples
        K, N = w_mu.shape
        dev = w_mu.device
        _need_f32("LRLinearFn.backward", x, y)
        # F1: hand-written backward kernels, eps regenerated on chip, v saved by the forward
        g_wmu, g_wrho, g_bmu, g_brho, gx = ops.lr_linear_bwd(
            x, gy.float(), y, v, w_mu, w_rho, b_mu, b_rho, n_samples=S, sigma_p=call.prior.sigma_p, relu=call.relu,
            eps_mode=call.eps_mode, eps_act=eps_act, eps_b=eps_b, seed=call.seed, layer_id=call.layer_id,
            sample_offset=call.sample_offset, sample_counter=call.sample_counter,
            g_kl=gkl3 if call.want_stats else None, want_gx=ctx.needs_input_grad[0], math_mode=call.math_mode)
        if gx is not None and x.dim() == 2:
            gx = gx.sum(0)
        return gx, g_wmu, g_wrho, g_bmu, g_brho, None, None, None


@dataclass(frozen=True)
class NetCall:
    """Static description of one ELBO evaluation of a whole stack of layers (ElboFn)."""
    layers: tuple                 # per layer: (local_reparam, relu, layer_id, PriorSpec, in_features, out_features)
    n_samples: int                # local MC samples
    sample_offset: int
    mode: str
    sigma: float
    math_mode: int
    seed: int
    sample_counter: Optional[torch.Tensor] = None


class ElboFn(torch.autograd.Function):
    """(x, target, *[w_mu, w_rho, b_mu, b_rho per layer]) -> float32[3] = the local samples' sums
    {sum_s log p | sum_s KL, sum_s log q | 0, sum_s nll}: the whole network of networks.py:166-190 as ONE
    autograd node.  Forward = layer kernels + bnn_elbo_finalize; backward = bnn_nll_bwd + the layer
    backward kernels chained by hand (what train.GraphedTrainStep captures), so an unchanged
    `loss.backward()` loop pays ~15 launches per step instead of autograd's ~60.  On-chip eps only."""

    @staticmethod
    def forward(ctx, x, target, call: NetCall, *params):
        S = call.n_samples
        h = x
        saved, wss = [], []
        for i, (lr, relu, layer_id, prior, _fin, _fout) in enumerate(call.layers):
            p = tuple(t.detach() for t in params[4 * i:4 * i + 4])
            common = dict(n_samples=S, math_mode=call.math_mode, relu=relu, y_dtype=torch.float32, eps_mode=L.EPS_PHILOX,
                          seed=call.seed, layer_id=layer_id, sample_offset=call.sample_offset,
                          sample_counter=call.sample_counter)
            if lr:
                out = ops.lr_linear_fwd(h, *p, sigma_p=prior.sigma_p, want_kl=True, want_v=True, **common)
            else:
                out = ops.bbb_linear_fwd(h, *p, prior=prior, want_stats=True, **common)
            saved += [h, out["y"], out.get("v")]
            wss.append(out["workspace"])
            h = out["y"]
        lr_net = bool(call.layers[0][0])
        sums = torch.empty(4, dtype=torch.float32, device=h.device)
        ops.elbo_finalize(workspaces=wss, layer_in=[l[4] for l in call.layers], layer_out=[l[5] for l in call.layers],
                          local_reparam=lr_net, prior=call.layers[0][3], n_samples=S, logits=h, target=target, mode=call.mode,
                          nll_sigma=call.sigma, sums=sums,
                          ticket=torch.zeros(1, dtype=torch.int32, device=h.device) if S > 1 else None)
        ctx.call = call
        ctx.n_saved = len(saved)
        ctx.save_for_backward(target, *params, *saved)
        return sums[:3].clone()

    @staticmethod
    def backward(ctx, gsums):
        call: NetCall = ctx.call
        S = call.n_samples
        nl = len(call.layers)
        tensors = ctx.saved_tensors
        target, params, saved = tensors[0], tensors[1:1 + 4 * nl], tensors[1 + 4 * nl:]
        lr_net = bool(call.layers[0][0])
        gsums = gsums.float()
        g_a = gsums[0].expand(S).contiguous()
        g_b = gsums[1].expand(S).contiguous()
        g_nll = gsums[2].expand(S).contiguous()
        # LR: the first sum is S times the layers' total KL, so each layer's KL sees S * g
        g_kl3 = torch.stack([gsums[0] * S, torch.zeros_like(gsums[0]), torch.zeros_like(gsums[0])]) if lr_net else None
        logits = saved[3 * (nl - 1) + 1]
        g = ops.nll_bwd(logits, target, g_nll, call.mode, call.sigma)
        grads = [None] * (4 * nl)
        gx = None
        for i in reversed(range(nl)):
            lr, relu, layer_id, prior, _fin, _fout = call.layers[i]
            xin, y, v = saved[3 * i:3 * i + 3]
            p = params[4 * i:4 * i + 4]
            kw = dict(n_samples=S, relu=relu, eps_mode=L.EPS_PHILOX, seed=call.seed, layer_id=layer_id,
                      sample_offset=call.sample_offset, sample_counter=call.sample_counter,
                      want_gx=(i > 0) or ctx.needs_input_grad[0])
            if lr:
                out = ops.lr_linear_bwd(xin, g, y if relu else None, v, *p, sigma_p=prior.sigma_p, g_kl=g_kl3,
                                        math_mode=call.math_mode, **kw)
            else:
                out = ops.bbb_linear_bwd(xin, g, y if relu else None, *p, prior=prior, math_mode=call.math_mode,
                                         g_log_prior=g_a, g_log_q=g_b, **kw)
            grads[4 * i:4 * i + 4] = out[:4]
            g = out[4]
        if ctx.needs_input_grad[0] and g is not None:
            gx = g.sum(0) if g.dim() == 3 and tensors[1 + 4 * nl].dim() == 2 else g
        return (gx, None, None, *grads)


class NLLFn(torch.autograd.Function):
    """logits[S,B,C], target -> nll[S] via K4 (networks.py:183-190)."""

    @staticmethod
    def forward(ctx, logits, target, mode: str, sigma: float):
        S = logits.shape[0]
        out = ops.elbo_finalize(workspaces=[], layer_in=[], layer_out=[], local_reparam=False,
                                prior=PriorSpec(), n_samples=S, logits=logits, target=target, mode=mode,
                                nll_sigma=sigma)
        ctx.mode, ctx.sigma = mode, sigma
        ctx.save_for_backward(logits, target)
        return out["nll"]

    @staticmethod
    def backward(ctx, gnll):
        logits, target = ctx.saved_tensors
        S = logits.shape[0]
        _need_f32("NLLFn.backward", logits)
        return ops.nll_bwd(logits, target, gnll, ctx.mode, ctx.sigma), None, None, None


class AllReduceSumFn(torch.autograd.Function):
    """Sum all-reduce of a small tensor over the default process group (RCCL on GPUs).  The
    adjoint of y = sum_r x_r w.r.t. the local x_r is the identity on the (replicated) grad."""

    @staticmethod
    def forward(ctx, t):
        import torch.distributed as dist
        out = t.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        return out

    @staticmethod
    def backward(ctx, g):
        return g
