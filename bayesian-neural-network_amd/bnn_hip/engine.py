"""Batched-samples launcher: what replaces the reference's serial MC loop
(networks.py:199-203, :217-220).  All locally owned MC samples of an ELBO evaluation go
through ONE launch per layer plus one finalize launch; with sample sharding enabled the
S global samples are split over the ranks of the default process group and the only
collective is a sum all-reduce of three scalars (RCCL over xGMI on a GPU node).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import ops
from .functional import AllReduceSumFn, BBBLinearFn, ElboFn, LayerCall, LRLinearFn, NetCall, NLLFn
from .runtime import state, take_samples


CAST_INPUT_MIN_SAMPLES = 4    # BBB: from here on the first layer can take a block-GEMM form (bf16 x through LDS-DMA);
                              # below, the extra launch costs more than it saves
# LR: from one sample on.  Its first layer squares every x fragment it loads, and fp32 x doubles the bytes each block
# pulls through its CU's L1 (21.5 us for that layer against 14.7 us for the wider second one): casting first wins
# even with the extra launch in the chain (one-sample evaluation 59.4 -> 55.3 us)
CAST_INPUT_MIN_SAMPLES_LR = 1
LR_SQUARES_MIN_SAMPLES = 7    # LR: carry x^2 (bf16) between layers from here on (the block-GEMM form streams it)
# BBB: the K-sliced GEMM form needs a scratch for its fp32 partial tiles; the library's plan (bnn_bbb_plan) decides
# whether and how finely to slice (bbb_linear.hip: kslices) -- here only the cases it can never take are left without
SPLIT_MIN_SAMPLES = 4
SPLIT_MIN_WEIGHTS = 250_000
SPLIT_MAX_UNITS = 1024        # (64-feature group x sample x batch block) units: 256 KiB of scratch each

# differentiable sample_elbo*: the whole network as one autograd node (functional.ElboFn) when eps is drawn on
# chip; the identical-eps parity path keeps one node per layer
FUSED_ELBO_NODE = True


# BBB, bf16 math: evaluations of at most this many (minibatch, MC sample) pairs draw the OUTPUT layer's weights beside the
# layer before it (a sampling job riding on that launch, bnn_bbb_fwd_args.rider) and run the output layer + finalize in
# the row-split matmul-only form (K1r): the tail of the dependent chain is then ~5 us instead of ~13
FINAL_ROWS_MAX_SAMPLES = 16
# (Letting the FIRST layer's launch carry the sampling of every later layer -- hidden layers matmul-only too -- was
# built and measured slower at every size: one evaluation 36.8 against 35.4 us at one sample, 58.8 against 43.6 at three:
# the rider blocks inherit the layer's 768-thread / 98 KB-LDS block shape, one per CU, and queue behind the layer's own.
# PRESAMPLE_HIDDEN_MAX_SAMPLES = 0 keeps that form off; tools/few_sample_sweep.py re-measures it.)
PRESAMPLE_HIDDEN_MAX_SAMPLES = 0


def final_rows_ok(specs, n_samples: int, batch: int, hidden_dtype) -> bool:
    """The output layer of this evaluation can take the pre-sampled row-split form."""
    if len(specs) < 2 or any(sp.lr for sp in specs) or hidden_dtype != torch.bfloat16 or state.form != L.FORM_AUTO or \
            state.math != L.MATH_BF16:                  # (the pre-sampled forms carry bf16 weights: plain bf16 math only)
        return False
    k_last, n_last = specs[-1].in_out
    return 0 < n_samples <= FINAL_ROWS_MAX_SAMPLES and n_last <= 16 and batch <= 128 and k_last % 8 == 0


# BBB, bf16 math, forward-only: layers fed with at least this many batch rows sample their weights once per launch
# (bnn_bbb_sample_weights: 8 B read + 2 B written per weight and sample, statistics included) and run the matmul over
# them in the library's 256 x 256 block form (K1g, csrc/bbb_block_gemm.h: bnn_bbb_linear_fwd takes it for w_sampled and
# >= 512 batch rows): 2 * batch flops per sampled weight make the matrix cores the bound, and the fused kernels would
# redo the sampling for every 128-row batch block
BLOCK_GEMM_MIN_BATCH = 512
# GraphedElbo: those layers' sampling launches on a side stream, beside the earlier layers' matmuls.  Measured and left OFF
# (tools/side_stream_ab.py, profiles/r04_side_stream_ab.log: 580 against 573 us per 4-sample evaluation at batch 1024, 1640
# against 1617 at batch 4096): a K1g block takes 2 x 216 of a SIMD's 512 registers and 128 KiB of LDS, so a sampling block
# only ever runs where a matmul block has not started yet -- the launches take turns on a CU instead of sharing it, and the
# cross-stream edges cost more than the little that overlaps.  The bits are the same either way (tests).
SAMPLE_BESIDE_MATMUL = False
# LR, one minibatch: the prepare launch (fragments of the layers after the first) depends on no activation and the first layer (K3s
# on the shared input, 152 blocks) leaves a hundred CUs idle -- it can run on a side stream beside that layer (a fork / join inside
# the evaluation, so a captured graph keeps the edges).  Measured SLOWER by 9 us at every sample count (7 ... 64 samples: 88 against
# 79 us at 8; profiles/r04_prepare_side_stream.log): a cross-stream edge inside a hipGraph costs more than the 5 us launch it
# hides -- the same lesson as SAMPLE_BESIDE_MATMUL.  Off.
PREPARE_BESIDE_FIRST_LAYER = False


def use_block_gemm(sp, batch: int, hidden_dtype) -> bool:
    return (not sp.lr) and hidden_dtype == torch.bfloat16 and batch >= BLOCK_GEMM_MIN_BATCH and sp.in_out[0] % 8 == 0 and \
        state.form == L.FORM_AUTO and state.math == L.MATH_BF16


# BBB, bf16 math: from this many (minibatch, sample) pairs per launch on, the output layer + finalize can run in the row-split form
# (K1r) behind a sampling launch of their own (K1s for the output layer's few weights) instead of one block per pair that samples,
# multiplies and finalizes (K1c: 28.6 us at 256 pairs + 4 us for the sums launch).  Built, oracle-checked at (256, 1), and measured
# SLOWER: 621 against 613 us per launch group of 256 minibatches (tools/rows_alone_ab.py, profiles/r04_rows_alone_ab.log) -- the
# sampling launch and 2304 small blocks cost more than the one chain per pair they replace.  Off (set to 64 to try it).
FINAL_ROWS_ALONE_MIN_SAMPLES = 10 ** 9
FINAL_ROWS_ALONE_MAX_SAMPLES = 4096


def final_rows_alone(specs, n_samples: int, batch: int, hidden_dtype) -> bool:
    if len(specs) < 2 or any(sp.lr for sp in specs) or hidden_dtype != torch.bfloat16 or state.form != L.FORM_AUTO or \
            state.math != L.MATH_BF16:
        return False
    k_last, n_last = specs[-1].in_out
    return FINAL_ROWS_ALONE_MIN_SAMPLES <= n_samples <= FINAL_ROWS_ALONE_MAX_SAMPLES and n_last <= 16 and batch <= 128 and k_last % 8 == 0


def presample_from(specs, n_samples: int, batch: int, hidden_dtype) -> int:
    """Index of the layer whose launch carries the sampling job of all layers after it (they run matmul-only), or -1:
    the layer before the output layer, or the first layer for very few (minibatch, sample) pairs."""
    if not final_rows_ok(specs, n_samples, batch, hidden_dtype):
        return -1
    last = len(specs) - 1
    if n_samples <= PRESAMPLE_HIDDEN_MAX_SAMPLES and all(sp.in_out[0] % 8 == 0 for sp in specs[1:]):
        return 0
    return last - 1


def use_split(fin: int, fout: int, n_samples: int, batch: int = 128) -> bool:
    units = ((fout + 63) // 64) * n_samples * ((batch + 127) // 128)
    return fout > 16 and fin % 8 == 0 and fin * fout >= SPLIT_MIN_WEIGHTS and n_samples >= SPLIT_MIN_SAMPLES and \
        units <= SPLIT_MAX_UNITS and state.math == L.MATH_BF16


def hoist_sigma(fin: int, fout: int, n_samples: int, batch: int = 128) -> bool:
    """sigma = softplus(rho) once per evaluation (part of the one prepare launch, ops.eval_prepare) instead of per
    sampled weight: pays when the layer runs in a block-GEMM form (K-sliced from SPLIT_MIN_SAMPLES, plain from 450
    units) and enough samples share it -- a quarter of the generator work of every sample against one 8 B/weight pass."""
    units = ((fout + 63) // 64) * n_samples * ((batch + 127) // 128)
    enough = n_samples >= (SIGMA_HOIST_MIN_SAMPLES if fin * fout < SIGMA_HOIST_BIG_LAYER else SIGMA_HOIST_MIN_SAMPLES_BIG)
    return fout > 16 and fin % 8 == 0 and fin * fout >= SPLIT_MIN_WEIGHTS and enough and \
        (units >= 450 or (n_samples >= SPLIT_MIN_SAMPLES and state.math == L.MATH_BF16))      # (bf16x3: no K-sliced form)


# measured: the pass costs ~1.6 us per million weights; what it saves per sample shrinks as more waves per SIMD cover the
# softplus (1200 x 1200, 8 samples: 31.5 -> 27.6 us; 4096 x 4096, 4 samples: 113.7 -> 107.5 us against a 25 us pass)
SIGMA_HOIST_BIG_LAYER = 4_000_000
SIGMA_HOIST_MIN_SAMPLES_BIG = 24
SIGMA_HOIST_MIN_SAMPLES = 4   # BBB: precompute sigma = softplus(rho) once per evaluation from here on (8 until round 4: the
                              # evaluation of one minibatch at 4 / 5 / 6 / 7 samples 58.2 / 59.6 / 63.6 / 68.8 -> 56.5 / 58.0 / 60.9 /
                              # 67.7 us, tools/hoist_threshold_sweep.py, profiles/r04_hoist_threshold.log)
LR_PREPARE_MIN_SAMPLES = 7    # LR: prepare bf16 (M, sigma^2) fragments once per evaluation from here on (tools/lr_mid_sweep.py)


LR_FINAL_ROWS_MAX_SAMPLES = 4096   # the output layer's fragments ride in the prepare launch and its row-split form (K3r) runs up to here
LR_SHARED_MAX_SAMPLES = 64    # csrc/lr_linear.hip: kLrsMaxShared (tools/lr_shared_sweep.py: faster than K3b + prepare + cast up to there)


def lr_unit_samples(samples: int, shared: bool) -> int:
    """Samples the K3s units count: all samples of a launch on ONE input (the first layer of sample_elbo_lr / predict: the
    reference runs forward(x) `samples` times on the same x, networks.py:211-225) share the unit's two products -- the
    kernel makes them once and runs the epilogue (bias, activation noise, stores) per sample."""
    return 1 if (shared and 2 <= samples <= LR_SHARED_MAX_SAMPLES) else samples


def lr_use_split(out_features: int, samples: int, batch: int, shared: bool = False) -> bool:
    """Worth handing bnn_lr_linear_fwd a split scratch (K3s: at most 160 (32-feature group, sample, batch block) units)."""
    samples = lr_unit_samples(samples, shared)
    return out_features >= 64 and out_features % 4 == 0 and ((out_features + 31) // 32) * samples * ((batch + 127) // 128) <= 160


def lr_kslice_expected(in_features: int, out_features: int, samples: int, batch: int, shared: bool = False) -> bool:
    """Mirror of the library's plan for K3s (csrc/lr_linear.hip: lr_plan): True where a launch with a split scratch and bf16
    math takes the K-sliced form -- used to decide whether a rider is worth attaching (elsewhere it costs a launch)."""
    if not lr_use_split(out_features, samples, batch, shared) or in_features % 8 or in_features < 64:
        return False
    samples = lr_unit_samples(samples, shared)
    units = ((out_features + 31) // 32) * samples * ((batch + 127) // 128)
    ksteps = (in_features + 31) // 32
    ksl = min(8, max(1, 160 // units))
    while ksl < 8 and (ksteps + ksl - 1) // ksl > 13:
        ksl += 1
    nst = (ksteps + ksl - 1) // ksl
    ksl = (ksteps + nst - 1) // nst
    # (two rounds' worth of blocks where the tile form would itself need a second round: lr_kslice_plan)
    tile_blocks = ((out_features + 15) // 16) * samples * ((batch + 127) // 128)
    return nst <= 13 and units * ksl <= (512 if (not shared and tile_blocks > 256) else 256)


def lr_use_prepare(n_out: int, n_samples: int, batch: int) -> bool:
    """True when the LR throughput kernel (block GEMM) will run for this layer and enough samples share the
    prepared weights to pay for the extra pass (mirrors the launcher's geometry rule)."""
    return n_samples >= LR_PREPARE_MIN_SAMPLES and ((n_out + 63) // 64) * n_samples * ((batch + 127) // 128) >= 130


def wide_nll(specs, batch: int) -> bool:
    """The finalize of this network sums its NLL by row blocks spread over the chip (bnn_elbo_finalize with a scratch):
    wide outputs, where one block per sample would walk batch x outputs elements through a single CU."""
    n_out = specs[-1].in_out[1]
    return n_out > 32 and batch * n_out >= 32768


def effective_math(any_lr: bool) -> int:
    """The math mode the launches of a network run in: the split-bf16 mode exists for the BBB forward kernels; the
    local-reparameterisation layers run it as exact fp32 (the mode's promise is the reference's fp32 arithmetic)."""
    return L.MATH_F32 if (state.math == L.MATH_BF16X3 and any_lr) else state.math


def shard_range(n_samples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of global sample indices owned by `rank`: [first, first+count).
    The first (n_samples % world) ranks own one extra sample."""
    base, extra = divmod(int(n_samples), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def dist_info() -> Tuple[int, int]:
    """(rank, world) of the default group when sample sharding is on, else (0, 1)."""
    if state.shard_samples:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _host_eps(shapes: Sequence[Tuple[int, ...]], n_samples: int, device) -> List[torch.Tensor]:
    """eps drawn from torch's CPU generator in the reference's order (per sample: layer
    1..3, weight-shaped then bias-shaped; networks.py:42, :75-76, :123-124), stacked per
    tensor over samples and copied H2D."""
    per = [[torch.randn(s) for s in shapes] for _ in range(n_samples)]
    return [torch.stack([per[i][j] for i in range(n_samples)]).to(device) for j in range(len(shapes))]


class LayerSpec:
    """What the engine needs to know about one stochastic layer."""

    def __init__(self, module, layer_id: int, local_reparam: bool, relu: bool):
        self.m, self.layer_id, self.lr, self.relu = module, layer_id, local_reparam, relu

    @property
    def in_out(self) -> Tuple[int, int]:
        w = self.m.weight_mu
        return (w.shape[0], w.shape[1]) if self.lr else (w.shape[1], w.shape[0])


def run_layers(layers: Sequence[LayerSpec], x: torch.Tensor, n_local: int, first_sample: int, *,
               want_stats: bool, sample: bool, injected: Optional[List[torch.Tensor]] = None,
               differentiable: bool, fin_kw: Optional[dict] = None):
    """Push `n_local` MC samples through the stack.  Returns (logits[S,B,C] fp32,
    per-layer stats).  Stats are (log_prior[S], log_q[S]) or kl3[3] tensors on the
    differentiable path and raw workspaces on the forward-only path."""
    any_lr = any(sp.lr for sp in layers)
    math_mode = effective_math(any_lr)
    x3 = math_mode == L.MATH_BF16X3
    hidden_dtype = torch.float32 if (differentiable or math_mode == L.MATH_F32) else torch.bfloat16
    h, h_sq, h_lo = x, None, None                  # h_lo: the low plane of a bf16 activation in split-bf16 math
    lr_sq = (not differentiable) and hidden_dtype == torch.bfloat16 and n_local >= LR_SQUARES_MIN_SAMPLES and any_lr
    # everything that depends on no activation, in one launch: the input batch in bf16 (every layer then streams 2-byte x;
    # LR: and its elementwise square) and sigma = softplus(rho) of the layers that will run a block-GEMM form
    want_cast = hidden_dtype == torch.bfloat16 and x.dtype == torch.float32 and n_local >= CAST_INPUT_MIN_SAMPLES  # (eager
    # calls are host-bound: the one-sample LR cast of GraphedElbo would only add a launch here)
    # LR, 2 .. 64 samples on one minibatch: the first layer's K3s makes its products once for all of them and reads the fp32
    # minibatch itself
    first_shared = (layers[0].lr and x.dim() == 2 and not differentiable and hidden_dtype == torch.bfloat16 and len(layers) > 1 and
                    lr_unit_samples(n_local, True) == 1 and lr_kslice_expected(*layers[0].in_out, n_local, x.shape[-2], True))
    if first_shared and all(sp.lr for sp in layers):
        want_cast = False
    hoisted = {}
    if hidden_dtype == torch.bfloat16 and sample and not differentiable and (x.dtype == torch.bfloat16 or want_cast):
        hoisted = {i: None for i, sp in enumerate(layers)
                   if not sp.lr and hoist_sigma(*sp.in_out, n_local, x.shape[-2]) and not use_block_gemm(sp, x.shape[-2], hidden_dtype)}
    if want_cast or hoisted:
        c16lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if (want_cast and x3) else None
        sig, c16, c16sq = ops.eval_prepare([layers[i].m.weight_rho.detach() for i in hoisted],
                                           cast=x if want_cast else None, want_sq=lr_sq and want_cast, cast_out_lo=c16lo)
        hoisted = dict(zip(hoisted, sig))
        if want_cast:
            h, h_sq, h_lo = c16, c16sq, c16lo
    stats = []
    # forward-only ELBO with on-chip eps: the output layer may take the pre-sampled row-split form (final_rows_ok)
    pre_from = presample_from(layers, n_local, x.shape[-2], hidden_dtype) \
        if (not differentiable and fin_kw is not None and want_stats and sample and injected is None) else -1
    presampled = {}                                   # layer index -> dict(w, b, workspace) drawn by an earlier launch
    for i, sp in enumerate(layers):
        last = i == len(layers) - 1
        if not sample:
            eps_mode, e_w, e_b = L.EPS_ZERO, None, None
        elif injected is not None:
            eps_mode, e_w, e_b = L.EPS_MEMORY, injected[2 * i], injected[2 * i + 1]
        else:
            eps_mode, e_w, e_b = L.EPS_PHILOX, None, None
        call = LayerCall(n_samples=n_local, prior=sp.m._prior_spec, math_mode=math_mode, relu=sp.relu,
                         eps_mode=eps_mode, seed=state.seed, layer_id=sp.layer_id, sample_offset=first_sample,
                         want_stats=want_stats, y_dtype=torch.float32 if last else hidden_dtype,
                         sample_counter=state.device_counter if (differentiable and eps_mode == L.EPS_PHILOX) else None)
        p = (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho)
        if differentiable:
            if sp.lr:
                h, kl3 = LRLinearFn.apply(h, *p, e_w, e_b, call)
                stats.append(kl3)
            else:
                h, lp, lq = BBBLinearFn.apply(h, *p, e_w, e_b, call)
                stats.append((lp, lq))
        else:
            pd = tuple(t.detach() for t in p)
            if sp.lr:
                want_sq = lr_sq and not last
                wfrag, ws_pre = (None, None)
                shared_i = i == 0 and first_shared
                if lr_sq and lr_use_prepare(sp.in_out[1], n_local, h.shape[-2]) and not shared_i:
                    wfrag, ws_pre = ops.lr_prepare(*pd)
                if last and fin_kw is not None and want_stats and eps_mode == L.EPS_PHILOX and fin_kw.get("scratch") is not None \
                        and not wide_nll(layers, h.shape[-2]) and all(q.lr for q in layers):
                    # output layer + finalize through bnn_lr_final_fwd: one launch when its row-split form applies
                    ws_last = ws_pre if ws_pre is not None else ops.lr_workspace(sp.in_out[1], h.device)
                    out, fin = ops.lr_final_fwd((h,) + pd, dict(w_frag=wfrag, workspace=ws_last, form=state.form, n_samples=n_local,
                                                                sigma_p=call.prior.sigma_p, math_mode=math_mode, relu=sp.relu,
                                                                y_dtype=call.y_dtype, eps_mode=eps_mode, seed=state.seed,
                                                                layer_id=sp.layer_id, sample_offset=first_sample, want_kl=True,
                                                                x_sq=h_sq if lr_sq else None),
                                                dict(workspaces=stats + [ws_last], **fin_kw))
                    return out["y"], fin
                out = ops.lr_linear_fwd(h, *pd, w_frag=wfrag, workspace=ws_pre, form=state.form, n_samples=n_local, sigma_p=call.prior.sigma_p, math_mode=math_mode,
                                        relu=sp.relu, y_dtype=call.y_dtype, eps_mode=eps_mode, eps_act=e_w, eps_b=e_b,
                                        seed=state.seed, layer_id=sp.layer_id, sample_offset=first_sample,
                                        want_kl=want_stats, x_sq=h_sq if lr_sq else None,
                                        out_sq=torch.empty((n_local, h.shape[-2], sp.in_out[1]), dtype=torch.bfloat16,
                                                           device=h.device) if want_sq else None,
                                        split_scratch=ops.lr_split_scratch_cached(lr_unit_samples(n_local, shared_i), h.shape[-2], sp.in_out[1], h.device)
                                        if (wfrag is None and not last and lr_use_split(sp.in_out[1], n_local, h.shape[-2], shared_i)) else None)
                h_sq = out["y_sq"]
            else:
                kw = dict(n_samples=n_local, prior=call.prior, math_mode=math_mode, relu=sp.relu, y_dtype=call.y_dtype,
                          eps_mode=eps_mode, eps_w=e_w, eps_b=e_b, seed=state.seed, layer_id=sp.layer_id,
                          sample_offset=first_sample, want_stats=want_stats, form=state.form)
                if x3 and h.dtype == torch.bfloat16:
                    kw["x_lo"] = h_lo
                if eps_mode == L.EPS_PHILOX and use_block_gemm(sp, h.shape[-2], hidden_dtype):
                    if h.dtype != torch.bfloat16:
                        h = ops.cast_bf16(h)
                    sm = ops.bbb_sample_weights([dict(w_mu=pd[0], w_rho=pd[1], b_mu=pd[2], b_rho=pd[3], prior=call.prior,
                                                       layer_id=sp.layer_id)], n_samples=n_local, seed=state.seed,
                                                sample_offset=first_sample)[0]
                    h = ops.bbb_sampled_matmul(h, sm["w"], sm["b"], n_samples=n_local, relu=sp.relu, y_dtype=call.y_dtype)
                    stats.append(sm["workspace"])
                    continue
                if i in hoisted and h.dtype == torch.bfloat16:
                    kw["w_sigma"] = hoisted[i]                 # consumed by the block-GEMM forms only
                if h.dtype == torch.bfloat16 and use_split(sp.in_out[0], sp.in_out[1], n_local, h.shape[-2]):
                    kw["split_scratch"] = ops.split_scratch_cached(n_local, h.shape[-2], sp.in_out[1], h.device)
                if last and fin_kw is not None and want_stats:
                    # last layer + finalize in one launch (when the layer is a single feature tile)
                    if i in presampled:
                        ps = presampled[i]
                        out, fin = ops.bbb_final_fwd((h, None, None, None, None),
                                                     dict(n_samples=n_local, prior=call.prior, math_mode=math_mode, relu=sp.relu,
                                                          y_dtype=call.y_dtype, eps_mode=L.EPS_ZERO, want_stats=False,
                                                          w_sampled=ps["w"], b_sampled=ps["b"]),
                                                     dict(workspaces=stats + [ps["workspace"]], **fin_kw))
                    else:
                        out, fin = ops.bbb_final_fwd((h,) + pd, kw, dict(workspaces=stats, **fin_kw))
                    return out["y"], fin
                if i in presampled:                   # a hidden layer whose weights an earlier launch has drawn
                    ps = presampled[i]
                    h = ops.bbb_sampled_matmul(h, ps["w"], ps["b"], n_samples=n_local, relu=sp.relu, y_dtype=call.y_dtype)
                    stats.append(ps["workspace"])
                    continue
                if i == pre_from:
                    # the later layers' weights are drawn beside this layer (a sampling job riding on its launch)
                    kw["rider"] = ops.build_sample_job(
                        [dict(w_mu=q.m.weight_mu.detach(), w_rho=q.m.weight_rho.detach(), b_mu=q.m.bias_mu.detach(),
                              b_rho=q.m.bias_rho.detach(), prior=q.m._prior_spec, layer_id=q.layer_id)
                         for q in layers[i + 1:]], n_samples=n_local, seed=state.seed, sample_offset=first_sample)
                    presampled = {i + 1 + j: r for j, r in enumerate(kw["rider"][1])}
                out = ops.bbb_linear_fwd(h, *pd, **kw)
                h_lo = out.get("y_lo")
            h = out["y"]
            stats.append(out["workspace"])
    return h, stats


def eps_shapes(layers: Sequence[LayerSpec], batch: int) -> List[Tuple[int, ...]]:
    shapes = []
    for sp in layers:
        fin, fout = sp.in_out
        shapes += [(batch, fout) if sp.lr else (fout, fin), (fout,)]
    return shapes


def collect_injected(layers: Sequence[LayerSpec], batch: int, n_samples: int, device):
    """Identical-eps seam.  If any layer's `.normal` attribute was replaced (the way the
    reference's draws are stubbed, networks.py:35/:100) call it in the reference's order and
    stack per tensor; if BNN_HIP_EPS=host draw from torch's CPU generator; else None
    (on-chip Philox)."""
    stubbed = any(sp.m._eps_stubbed() for sp in layers)
    if not stubbed and not state.host_eps:
        return None
    shapes = eps_shapes(layers, batch)
    if not stubbed:
        return _host_eps(shapes, n_samples, device)
    per_sample = []
    for _ in range(n_samples):
        row = []
        for i, sp in enumerate(layers):
            row += sp.m._draw_eps(shapes[2 * i], shapes[2 * i + 1])
        per_sample.append(row)
    return [torch.stack([per_sample[s][j].float() for s in range(n_samples)]).to(device).contiguous()
            for j in range(len(shapes))]


def elbo_terms(layers: Sequence[LayerSpec], x: torch.Tensor, target: torch.Tensor, samples: int, *, mode: str,
               sigma: float, local_reparam: bool):
    """The per-evaluation sums every ELBO variant needs, sharded over ranks when enabled.

    Returns (sum_a, sum_b, sum_nll, n_total) as 0-dim fp32 tensors with
    sum_a = sum_s log p (BBB) or sum_s KL (LR), sum_b = sum_s log q (BBB) or 0."""
    rank, world = dist_info()
    first_global = take_samples(samples)
    lo, n_local = shard_range(samples, rank, world)
    dev = x.device
    differentiable = torch.is_grad_enabled() and any(
        p.requires_grad for sp in layers for p in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
    if differentiable and world > 1:
        # each rank's autograd graph covers its own samples only and nothing here all-reduces parameter gradients:
        # a per-rank optimizer.step() would make the replicas diverge silently
        raise ops.BnnHipError(
            "sample-sharded sample_elbo* is forward-only (use torch.no_grad()); for multi-GPU training shard the "
            "minibatches instead: bnn_hip.train.GraphedTrainStep(data_parallel=True) sum-all-reduces the gradients")
    zero = torch.zeros((), dtype=torch.float32, device=dev)
    if n_local == 0:
        sums = torch.stack([zero, zero, zero])
    else:
        B = x.shape[0]
        injected = collect_injected(layers, B, samples, dev)
        if injected is not None and world > 1:
            injected = [t[lo:lo + n_local].contiguous() for t in injected]
        fin_kw = None
        if not differentiable:
            fin_kw = dict(layer_in=[sp.in_out[0] for sp in layers], layer_out=[sp.in_out[1] for sp in layers],
                          local_reparam=local_reparam, prior=layers[0].m._prior_spec, n_samples=n_local, target=target,
                          mode=mode, nll_sigma=float(sigma),
                          ticket=torch.zeros(1, dtype=torch.int32, device=dev) if n_local > 1 else None,
                          scratch=ops.final_scratch(n_local, dev)
                          if (not local_reparam or wide_nll(layers, x.shape[0]) or layers[-1].in_out[1] <= 16) else None)
        fused_node = differentiable and injected is None and FUSED_ELBO_NODE and x.dtype == torch.float32 and \
            all(bool(sp.lr) == bool(local_reparam) for sp in layers)
        if fused_node:
            # the whole network as one autograd node (functional.ElboFn): ~15 launches per step instead of ~60
            call = NetCall(layers=tuple((bool(sp.lr), bool(sp.relu), sp.layer_id, sp.m._prior_spec, *sp.in_out) for sp in layers),
                           n_samples=n_local, sample_offset=first_global + lo, mode=mode, sigma=float(sigma),
                           math_mode=state.math, seed=state.seed, sample_counter=state.device_counter)
            params = [t for sp in layers for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho)]
            sums = ElboFn.apply(x, target, call, *params)
            if world > 1:
                sums = AllReduceSumFn.apply(sums)
            return sums[0], sums[1], sums[2], samples
        logits, stats = run_layers(layers, x, n_local, first_global + lo, want_stats=True, sample=True,
                                   injected=injected, differentiable=differentiable,
                                   fin_kw=fin_kw)
        if differentiable:
            nll = NLLFn.apply(logits, target, mode, float(sigma))
            if local_reparam:
                kl = stats[0][0]
                for k3 in stats[1:]:
                    kl = kl + k3[0]                       # networks.py:181: l1 + l2 + l3
                a, b = kl * n_local, zero
            else:
                lp = stats[0][0]
                lq = stats[0][1]
                for (p_, q_) in stats[1:]:
                    lp, lq = lp + p_, lq + q_              # networks.py:174-178
                a, b = lp.sum(), lq.sum()
            sums = torch.stack([a, b, nll.sum()])
        else:
            fin = stats if isinstance(stats, dict) else ops.elbo_finalize(workspaces=stats, logits=logits, **fin_kw)
            if local_reparam:
                sums = torch.stack([fin["kl"].sum(), zero, fin["nll"].sum()])
            else:
                sums = torch.stack([fin["log_prior"].sum(), fin["log_q"].sum(), fin["nll"].sum()])
    if world > 1:
        sums = AllReduceSumFn.apply(sums)
    return sums[0], sums[1], sums[2], samples


def mc_forward(layers: Sequence[LayerSpec], x: torch.Tensor, samples: int) -> torch.Tensor:
    """Outputs of `samples` stochastic forward passes in ONE launch per layer: what the reference
    collects by calling net(x, sample=True) in a Python loop (regression/reg_task.py:76-83,
    classification/class_task.py:83-85).  Returns this rank's [n_local, B, out] fp32 block."""
    rank, world = dist_info()
    first_global = take_samples(samples)
    lo, n_local = shard_range(samples, rank, world)
    if n_local == 0:
        return torch.empty((0, x.shape[0], layers[-1].in_out[1]), dtype=torch.float32, device=x.device)
    injected = collect_injected(layers, x.shape[0], samples, x.device)
    if injected is not None and world > 1:
        injected = [t[lo:lo + n_local].contiguous() for t in injected]
    with torch.no_grad():
        logits, _ = run_layers(layers, x, n_local, first_global + lo, want_stats=False, sample=True, injected=injected,
                               differentiable=False)
    return logits


def mc_predict(layers: Sequence[LayerSpec], x: torch.Tensor, samples: int):
    """F3: (preds[B], probs[B,C]) with probs = mean_s softmax(net(x, sample=True)) (class_task.py:81-87),
    the samples batched per launch and, when sharding is on, split over the ranks."""
    logits = mc_forward(layers, x, samples)
    rank, world = dist_info()
    if world > 1:
        import torch.distributed as dist
        if logits.shape[0]:
            probs, _ = ops.mc_softmax_mean(logits, 1.0 / samples, want_preds=False)
        else:
            probs = torch.zeros((x.shape[0], layers[-1].in_out[1]), dtype=torch.float32, device=x.device)
        dist.all_reduce(probs, op=dist.ReduceOp.SUM)
        return torch.argmax(probs, dim=1), probs
    probs, preds = ops.mc_softmax_mean(logits, 1.0 / samples)
    return preds, probs


class GraphedElbo:
    """Forward-only ELBO evaluations with static buffers: one launch per layer for ALL local MC samples (the last
    BBB layer carries the finalize), captured once as a hipGraph and replayed.  The Philox sample index has a
    device-resident part (`counter`) that the finalize kernel advances, so every replay draws fresh epsilon.

    `x` is one minibatch [B, ...] or a stack of G independent minibatches [G, B, ...] (`target` likewise [B] /
    [G, B]): what the reference evaluates one after the other (class_task.py:89-103 walks the test loader) runs as
    one launch per layer over the G x samples (minibatch, MC sample) pairs, exactly as the MC samples of one
    minibatch do.  Every pair draws its own weights (its own global sample index).

    `capture`: True = a hipGraph (replayed on the stream current at replay time, or on `stream`); "calls" = the evaluation's C-ABI
    launches recorded once and called again per replay, on the stream they were recorded on -- no graph, so none of the ~8 us a
    graph replay spends around its nodes: the faster form for evaluations of up to ~16 (minibatch, sample) pairs; False = eager.

    `replay()` returns the static float32 tensor [G, 4] (or [4] for one minibatch) of
    {sum log p | sum KL, sum log q | 0, sum nll, local sample count} per minibatch: the vector(s) a sharded job
    all-reduces.  `out` holds the per-(minibatch, sample) scalars, `logits` the outputs [G * S_local, B, C]."""

    def __init__(self, net, x: torch.Tensor, target: torch.Tensor, samples: int, sigma: float = 1.0,
                 capture: bool = True, stream: Optional[torch.cuda.Stream] = None, evals_per_replay: int = 1,
                 stacked: bool = False):
        """`stacked`: x / target carry a leading minibatch dimension G.
        `evals_per_replay` E > 1: one replay runs E consecutive evaluations of the same inputs with fresh
        epsilon (one graph of E times the kernels; a hipGraph launch costs the host ~10 us + ~1 us per node, so
        short evaluations are launch-bound one at a time); `sums` / `out` / `logits` then hold the LAST one."""
        self.net, self.samples, self.sigma = net, int(samples), float(sigma)
        self.per_replay = max(1, int(evals_per_replay))
        self.stream = stream
        self.rank, self.world = dist_info()
        self.lo, self.s_local = shard_range(self.samples, self.rank, self.world)
        if self.s_local <= 0:
            raise ops.BnnHipError("GraphedElbo: this rank owns no MC sample (samples < world size)")
        self.specs = net._specs()
        self.lr = bool(net.local_reparam)
        dev = x.device
        self.G = int(x.shape[0]) if stacked else 1
        if stacked:
            if target.shape[0] != self.G:
                raise ops.BnnHipError("GraphedElbo: stacked x and target must agree in their leading dimension")
            xf = torch.stack([net._flat(x[g]) for g in range(self.G)]) if self.G > 1 else net._flat(x[0])
            target = target if self.G > 1 else target[0]
        else:
            xf = net._flat(x)
        self.x = xf.contiguous()
        self.target = target.contiguous()
        self.n_local = self.G * self.s_local                 # (minibatch, MC sample) pairs in a launch
        self.group = self.s_local if self.G > 1 else 0       # samples per minibatch, as the library's group fields
        self.total_samples = self.G * self.samples           # global sample indices one evaluation spans
        first = take_samples(0)
        self.counter = torch.tensor([first], dtype=torch.int32, device=dev)
        S = self.n_local
        B = self.x.shape[-2]
        # split-bf16 math on a local-reparameterisation network: a stream of stacked minibatches whose hidden layers all take the
        # block form over prepared fragments (K3b<X3>) runs them in that mode -- the layer below the output layer hands it fp32
        # activations and the narrow output layer runs exact fp32; any other LR evaluation runs exact fp32 throughout
        self.lr_x3 = (self.lr and state.math == L.MATH_BF16X3 and self.G > 1 and len(self.specs) >= 2 and S >= LR_SQUARES_MIN_SAMPLES and
                      self.x.dtype == torch.float32 and
                      all(sp.in_out[0] % 8 == 0 and lr_use_prepare(sp.in_out[1], S, B) for sp in self.specs[:-1]))
        math_mode = self.math = L.MATH_BF16X3 if self.lr_x3 else effective_math(self.lr)
        self.x3 = math_mode == L.MATH_BF16X3
        hid = torch.float32 if math_mode == L.MATH_F32 else torch.bfloat16
        self.bufs, self.ws, self.bufs_lo = [], [], []
        nl_ = len(self.specs)
        for i, sp in enumerate(self.specs):
            fin, fout = sp.in_out
            last = i == nl_ - 1
            f32_out = last or (self.lr_x3 and i == nl_ - 2)
            self.bufs.append(torch.empty((S, B, fout), dtype=torch.float32 if f32_out else hid, device=dev))
            # split-bf16 math: the low plane of every bf16 activation
            self.bufs_lo.append(torch.empty((S, B, fout), dtype=torch.bfloat16, device=dev) if (self.x3 and not f32_out) else None)
            self.ws.append(ops.lr_workspace(fout, dev) if self.lr else ops.bbb_workspace(S, fout, dev))
        keys = ("kl",) if self.lr else ("log_prior", "log_q")
        self.out = {k: torch.zeros(S, dtype=torch.float32, device=dev) for k in keys + ("nll",)}
        self._sums = torch.zeros((self.G, 4), dtype=torch.float32, device=dev)
        self.sums = self._sums if self.G > 1 else self._sums.view(4)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scratch = ops.final_scratch(S, dev) if (not self.lr or wide_nll(self.specs, B) or self.specs[-1].in_out[1] <= 16) else None
        # large batches: sample once per launch, then the 256 x 256 block form of the matmul (use_block_gemm)
        self.lib = [use_block_gemm(sp, B, hid) for sp in self.specs]
        self.lib_w = [torch.empty((S, sp.in_out[1], sp.in_out[0]), dtype=torch.bfloat16, device=dev) if lb else None
                      for sp, lb in zip(self.specs, self.lib)]
        self.lib_b = [torch.empty((S, sp.in_out[1]), dtype=torch.float32, device=dev) if lb else None
                      for sp, lb in zip(self.specs, self.lib)]
        for i, (sp, lb) in enumerate(zip(self.specs, self.lib)):
            if lb:
                self.ws[i] = ops.sample_workspace(S, sp.in_out[0], sp.in_out[1], dev)
        # LR, 1-2 samples: the first layer's K-sliced form (K3s) reads the fp32 minibatch itself -- no cast launch ahead of it
        # (2 .. 64 samples of one minibatch share that layer's products: lr_unit_samples)
        shared0 = self.lr and not self.G > 1 and lr_unit_samples(S, True) == 1
        k3s_first = (self.lr and hid == torch.bfloat16 and len(self.specs) > 1 and self.specs[0].in_out[0] % 8 == 0 and
                     lr_use_split(self.specs[0].in_out[1], S, B, shared0) and not self.G > 1)
        shared0 = shared0 and k3s_first
        self.x16 = (torch.empty(self.x.shape, dtype=torch.bfloat16, device=dev)
                    if (hid == torch.bfloat16 and self.x.dtype == torch.float32 and
                        (self.lib[0] or S >= (CAST_INPUT_MIN_SAMPLES_LR if self.lr else CAST_INPUT_MIN_SAMPLES)) and
                        not k3s_first) else None)
        self.x16_lo = torch.empty(self.x.shape, dtype=torch.bfloat16, device=dev) if (self.x3 and self.x16 is not None) else None
        self.lr_sq = self.lr and (self.x16 is not None or (k3s_first and hid == torch.bfloat16)) and S >= LR_SQUARES_MIN_SAMPLES
        self.x16_sq = torch.empty(self.x.shape, dtype=torch.bfloat16, device=dev) if (self.lr_sq and self.x16 is not None) else None
        self.bufs_sq = [torch.empty(b.shape, dtype=torch.bfloat16, device=dev)
                        if (self.lr_sq and i < len(self.bufs) - 1 and b.dtype == torch.bfloat16) else None for i, b in enumerate(self.bufs)]
        self.split = [ops.split_scratch(S, B, sp.in_out[1], dev)
                      if (not self.lr and hid == torch.bfloat16 and not self.lib[i] and
                          use_split(sp.in_out[0], sp.in_out[1], S, B)) else None
                      for i, sp in enumerate(self.specs)]
        # K3s (1-3 samples on a wide LR layer): the K slices of a 32-feature group meet through this scratch; the library's
        # plan decides whether a launch uses it
        self.lr_split = [ops.lr_split_scratch(lr_unit_samples(S, shared0 and i == 0), B, sp.in_out[1], dev)
                         if (self.lr and hid == torch.bfloat16 and i < len(self.specs) - 1 and
                             lr_use_split(sp.in_out[1], S, B, shared0 and i == 0)) else None
                         for i, sp in enumerate(self.specs)]
        self.wsigma = [torch.empty_like(sp.m.weight_rho.detach())
                      if (not self.lr and hid == torch.bfloat16 and not lb and hoist_sigma(*sp.in_out, S, B)) else None
                      for sp, lb in zip(self.specs, self.lib)]
        self.wfrag = [None] * len(self.specs)
        if self.lr_sq:
            frag_bytes = L.load().bnn_lr_prepare_x3_bytes if self.lr_x3 else L.load().bnn_lr_prepare_bytes
            self.wfrag = [torch.empty(frag_bytes(*sp.in_out) // 4, dtype=torch.float32, device=dev)
                          if (lr_use_prepare(sp.in_out[1], S, B) and not (shared0 and i == 0) and not (self.lr_x3 and i == nl_ - 1)) else None
                          for i, sp in enumerate(self.specs)]
        # LR, few samples: the narrow output layer's operands are prepared by a rider of the previous layer's launch
        # (bnn_lr_rider), so that the row-split final launch (K3r) parks nothing
        nl = len(self.specs)
        self.lr_rider = None
        # ... and where the hidden layers' fragments come from a prepare launch anyway (>= 7 pairs: no K3s launch to ride on), the
        # narrow output layer's ride in that SAME launch (bnn_lr_prepare_many): K3r then fetches ready fragments instead of
        # every row block parking the whole layer
        if (self.lr_sq and not self.lr_x3 and self.wfrag[nl - 1] is None and any(w is not None for w in self.wfrag) and
                self.scratch is not None and not wide_nll(self.specs, B) and self.specs[-1].in_out[1] <= 16 and S <= LR_FINAL_ROWS_MAX_SAMPLES):
            self.wfrag[nl - 1] = torch.empty(L.load().bnn_lr_prepare_bytes(*self.specs[-1].in_out) // 4, dtype=torch.float32, device=dev)
        if (self.lr and hid == torch.bfloat16 and nl > 1 and self.wfrag[nl - 1] is None and self.scratch is not None and
                not wide_nll(self.specs, B) and self.specs[-1].in_out[1] <= 16 and S <= 16 and self.lr_split[nl - 2] is not None and
                lr_kslice_expected(*self.specs[-2].in_out, S, B)):
            sp = self.specs[-1]
            self.lr_rider = dict(w_mu=sp.m.weight_mu.detach(), w_rho=sp.m.weight_rho.detach(), b_mu=sp.m.bias_mu.detach(),
                                 b_rho=sp.m.bias_rho.detach(), workspace=self.ws[nl - 1],
                                 w_frag=torch.empty(L.load().bnn_lr_prepare_bytes(*sp.in_out) // 4, dtype=torch.float32, device=dev))
        self.pre_from = presample_from(self.specs, S, B, hid)      # the layer whose launch samples all layers after it
        self.rows = self.pre_from >= 0
        self.w_pre, self.b_pre = [None] * len(self.specs), [None] * len(self.specs)
        # many pairs: the output layer's weights from a sampling launch of its own, then the row-split final form
        self.rows_alone = (not self.rows) and self.scratch is not None and not self.lib[-1] and final_rows_alone(self.specs, S, B, hid)
        if self.rows_alone:
            k_i, n_i = self.specs[-1].in_out
            self.w_pre[-1] = torch.empty((S, n_i, k_i), dtype=torch.bfloat16, device=dev)
            self.b_pre[-1] = torch.empty((S, n_i), dtype=torch.float32, device=dev)
            self.ws[-1] = ops.sample_workspace(S, k_i, n_i, dev)
        if self.rows:
            for i in range(self.pre_from + 1, len(self.specs)):
                k_i, n_i = self.specs[i].in_out
                self.w_pre[i] = torch.empty((S, n_i, k_i), dtype=torch.bfloat16, device=dev)
                self.b_pre[i] = torch.empty((S, n_i), dtype=torch.float32, device=dev)
                self.ws[i] = ops.sample_workspace(S, k_i, n_i, dev)
        # large-batch layers (K1s + K1g): the sampling launches depend on no activation -- they run on a SIDE stream beside the
        # matmuls of the layers before them (vector / memory work next to matrix-core work on the same CUs: layer l's K1g
        # waits for layer l's K1s only), forked and joined inside the evaluation so that a captured graph keeps the edges
        self.side = torch.cuda.Stream(device=dev) if (SAMPLE_BESIDE_MATMUL and sum(self.lib) >= 1 and not self.lr) else None
        self.prep_side = (torch.cuda.Stream(device=dev)
                          if (PREPARE_BESIDE_FIRST_LAYER and self.lr and self.wfrag[0] is None and any(w is not None for w in self.wfrag))
                          else None)
        self.graph = None
        self.calls = None
        if capture == "calls":
            # The evaluation as a recorded list of C-ABI launches, replayed by calling them again: the argument structures are baked
            # exactly as a hipGraph bakes them (static buffers, the device-resident sample counter), but the launches go to the stream
            # one by one -- no graph, so none of the ~8 us a hipGraph replay spends around its nodes; the host pays ~3-5 us per
            # launch instead, hidden as long as an evaluation's kernels take longer than that.
            # (the stream is baked into the recorded calls: `stream` if given, else the stream current now)
            rec_stream = self.stream if self.stream is not None else torch.cuda.current_stream()
            with torch.cuda.stream(rec_stream):
                self._enqueue()                  # warm-up (also validates arguments eagerly)
            take_samples(self.total_samples)
            torch.cuda.synchronize()
            before = torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0)
            with L.recording() as calls, torch.cuda.stream(rec_stream):
                self._eager()
            take_samples(self.total_samples * self.per_replay)
            torch.cuda.synchronize()
            if torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0) != before:
                raise ops.BnnHipError("GraphedElbo(capture='calls'): the evaluation allocated device memory while it was recorded -- "
                                      "a replay would launch on freed buffers; use capture=True (hipGraph) for this configuration")
            self.calls = list(calls)
        elif capture:
            self._enqueue()                      # warm-up (also validates arguments eagerly)
            take_samples(self.total_samples)
            torch.cuda.synchronize()
            side = self.stream if self.stream is not None else torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    self._eager()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = g

    def _enqueue(self):
        """One evaluation of all local (minibatch, MC sample) pairs."""
        math_mode = self.math
        h_sq, h_lo = None, None
        # one launch for everything that depends on no activation: the bf16 input batch (+ squares), the hoisted sigmas
        hoist = [i for i, w in enumerate(self.wsigma) if w is not None]
        h = self.x
        if self.x16 is not None or hoist:
            ops.eval_prepare([self.specs[i].m.weight_rho.detach() for i in hoist], [self.wsigma[i] for i in hoist],
                             cast=self.x if self.x16 is not None else None, cast_out=self.x16,
                             cast_out_sq=self.x16_sq, cast_out_lo=self.x16_lo)
            if self.x16 is not None:
                h, h_sq, h_lo = self.x16, self.x16_sq, self.x16_lo
        last = len(self.specs) - 1
        prepared = None                                  # event: the side stream's prepare launch has been enqueued
        if self.lr and any(w is not None for w in self.wfrag):
            # the prepared operands of every layer that takes them, in ONE launch (they depend on no activation)
            jobs = [dict(w_mu=sp.m.weight_mu.detach(), w_rho=sp.m.weight_rho.detach(), b_mu=sp.m.bias_mu.detach(),
                         b_rho=sp.m.bias_rho.detach(), workspace=self.ws[i], out=self.wfrag[i])
                    for i, sp in enumerate(self.specs) if self.wfrag[i] is not None]
            if self.prep_side is not None:
                main = torch.cuda.current_stream()
                self.prep_side.wait_stream(main)         # fork: behind the previous evaluation's finalize (it read these buffers)
                with torch.cuda.stream(self.prep_side):
                    ops.lr_prepare_many(jobs, x3=self.lr_x3)
                    prepared = torch.cuda.Event()
                    prepared.record(self.prep_side)
            else:
                ops.lr_prepare_many(jobs, x3=self.lr_x3)
        grp = dict(sample_group=self.group, sample_group_stride=self.samples) if self.G > 1 else {}
        sampled = {}                                     # layer -> event: its K1s launch (side stream) has been enqueued
        if self.side is not None:
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)                  # fork: behind everything this stream has enqueued (the last finalize)
            with torch.cuda.stream(self.side):
                for i, sp in enumerate(self.specs):
                    if self.lib[i]:
                        self._sample_layer(i, grp)
                        sampled[i] = torch.cuda.Event()
                        sampled[i].record(self.side)
        fin_kw = dict(layer_in=[sp.in_out[0] for sp in self.specs], layer_out=[sp.in_out[1] for sp in self.specs],
                      local_reparam=self.lr, prior=self.specs[0].m._prior_spec, n_samples=self.n_local,
                      target=self.target, mode=self.net.mode, nll_sigma=self.sigma, sample_counter=self.counter,
                      sample_counter_inc=self.total_samples, out=self.out, sums=self._sums,
                      ticket=self.ticket, scratch=self.scratch, group_samples=self.group)
        for i, sp in enumerate(self.specs):
            p = tuple(t.detach() for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
            if i == last and self.side is not None:
                torch.cuda.current_stream().wait_stream(self.side)      # join: ahead of the launch that finalizes (it advances
                                                                        # the sample counter the sampling launches read)
            common = dict(n_samples=self.n_local, math_mode=math_mode, relu=sp.relu, y_dtype=self.bufs[i].dtype,
                          eps_mode=L.EPS_PHILOX, seed=state.seed, layer_id=sp.layer_id, sample_offset=self.lo,
                          sample_counter=self.counter, workspace=self.ws[i], out=self.bufs[i], form=state.form, **grp)
            if self.lr:
                if prepared is not None and self.wfrag[i] is not None:
                    torch.cuda.current_stream().wait_event(prepared)      # join: the first layer that reads prepared operands
                    prepared = None
                if self.lr_x3 and i == last:
                    common["math_mode"] = L.MATH_F32          # the narrow output layer: exact fp32 on the fp32 activations
                if i == last and self.scratch is not None and not wide_nll(self.specs, self.x.shape[-2]):
                    # output layer + finalize in one launch when the library's row-split form applies (else it issues both)
                    ops.lr_final_fwd((h,) + p, dict(sigma_p=sp.m._prior_spec.sigma_p, want_kl=True, x_sq=h_sq,
                                                    w_frag=self.lr_rider["w_frag"] if self.lr_rider is not None else self.wfrag[i],
                                                    **common), dict(workspaces=self.ws, **fin_kw))
                    return
                ops.lr_linear_fwd(h, *p, sigma_p=sp.m._prior_spec.sigma_p, want_kl=True, x_sq=h_sq,
                                  out_sq=self.bufs_sq[i], w_frag=self.wfrag[i], split_scratch=self.lr_split[i],
                                  rider=self.lr_rider if i == last - 1 else None,
                                  x_lo=h_lo if self.lr_x3 else None, out_lo=self.bufs_lo[i] if self.lr_x3 else None, **common)
                h_sq = self.bufs_sq[i]
            elif self.lib[i]:
                if i in sampled:
                    torch.cuda.current_stream().wait_event(sampled[i])
                else:
                    self._sample_layer(i, grp)
                ops.bbb_sampled_matmul(h, self.lib_w[i], self.lib_b[i], n_samples=self.n_local, relu=sp.relu,
                                       y_dtype=self.bufs[i].dtype, out=self.bufs[i])
                if i == last:
                    ops.elbo_finalize(workspaces=self.ws, logits=self.bufs[i], **fin_kw)
            else:
                kw = dict(prior=sp.m._prior_spec, want_stats=True, split_scratch=self.split[i], w_sigma=self.wsigma[i], **common)
                if self.x3:
                    kw.update(x_lo=h_lo if h.dtype == torch.bfloat16 else None, out_lo=self.bufs_lo[i])
                if i == last and self.rows_alone:
                    ops.bbb_sample_weights([dict(w_mu=p[0], w_rho=p[1], b_mu=p[2], b_rho=p[3], prior=sp.m._prior_spec, layer_id=sp.layer_id,
                                                 workspace=self.ws[i], w_out=self.w_pre[i], b_out=self.b_pre[i])],
                                           n_samples=self.n_local, seed=state.seed, sample_offset=self.lo, sample_counter=self.counter, **grp)
                if i == last and (self.rows or self.rows_alone):
                    ops.bbb_final_fwd((h, None, None, None, None),
                                      dict(n_samples=self.n_local, prior=sp.m._prior_spec, math_mode=math_mode, relu=sp.relu,
                                           y_dtype=self.bufs[i].dtype, eps_mode=L.EPS_ZERO, want_stats=False, out=self.bufs[i],
                                           w_sampled=self.w_pre[i], b_sampled=self.b_pre[i]),
                                      dict(workspaces=self.ws, **fin_kw))
                elif i == last:
                    ops.bbb_final_fwd((h,) + p, kw, dict(workspaces=self.ws[:last], **fin_kw))
                elif self.w_pre[i] is not None:              # a hidden layer whose weights an earlier launch has drawn
                    ops.bbb_sampled_matmul(h, self.w_pre[i], self.b_pre[i], n_samples=self.n_local, relu=sp.relu,
                                           y_dtype=self.bufs[i].dtype, out=self.bufs[i])
                else:
                    if i == self.pre_from:                   # the later layers' weights are drawn beside this layer
                        kw["rider"] = ops.build_sample_job(
                            [dict(w_mu=q.m.weight_mu.detach(), w_rho=q.m.weight_rho.detach(), b_mu=q.m.bias_mu.detach(),
                                  b_rho=q.m.bias_rho.detach(), prior=q.m._prior_spec, layer_id=q.layer_id,
                                  workspace=self.ws[j], w_out=self.w_pre[j], b_out=self.b_pre[j])
                             for j, q in enumerate(self.specs) if j > i],
                            n_samples=self.n_local, seed=state.seed, sample_offset=self.lo, sample_counter=self.counter, **grp)
                    ops.bbb_linear_fwd(h, *p, **kw)
            h, h_lo = self.bufs[i], self.bufs_lo[i]
        if self.lr:
            ops.elbo_finalize(workspaces=self.ws, logits=h, **fin_kw)

    def _sample_layer(self, i, grp):
        """K1s of layer i into its static buffers (on the current stream)."""
        sp = self.specs[i]
        p = tuple(t.detach() for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
        ops.bbb_sample_weights([dict(w_mu=p[0], w_rho=p[1], b_mu=p[2], b_rho=p[3], prior=sp.m._prior_spec, layer_id=sp.layer_id,
                                     workspace=self.ws[i], w_out=self.lib_w[i], b_out=self.lib_b[i])],
                               n_samples=self.n_local, seed=state.seed, sample_offset=self.lo, sample_counter=self.counter, **grp)

    def _eager(self):
        for _ in range(self.per_replay):
            self._enqueue()

    def replay(self) -> torch.Tensor:
        if self.calls is not None:
            for fn, args, name in self.calls:
                rc = fn(*args)
                if rc:
                    L.check(rc, name)
            take_samples(self.total_samples * self.per_replay)
            return self.sums
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self.graph.replay() if self.graph is not None else self._eager()
        elif self.graph is not None:
            self.graph.replay()
        else:
            self._eager()
        take_samples(self.total_samples * self.per_replay)     # keep the host-side counter in step
        return self.sums

    @property
    def logits(self) -> torch.Tensor:
        return self.bufs[-1]


class GraphedPredict(GraphedElbo):
    """F3 as a captured evaluation: the MC-averaged prediction of one minibatch (classification/class_task.py:81-87 --
    probs = mean_s softmax(net(x, sample=True)), preds = argmax) with static buffers, one hipGraph, FRESH epsilon on every
    replay (the device-resident sample counter of GraphedElbo).  It is GraphedElbo's launch chain for `samples` MC samples of
    the minibatch -- the forms a forward-only evaluation takes at that sample count, output layer in its row-split form -- with
    the softmax-mean launch (bnn_mc_softmax_mean) behind it; the ELBO scalars it also produces are computed against all-zero
    labels and mean nothing.  `replay()` returns the static (preds [B] int64, probs [B, C] float32); with sample sharding on,
    every rank runs its share of the samples and replay() sum-all-reduces the probabilities (outside the graph)."""

    def __init__(self, net, x: torch.Tensor, samples: int, capture: bool = True, stream: Optional[torch.cuda.Stream] = None):
        if net.mode != "classification":
            raise ops.BnnHipError("GraphedPredict: the MC-averaged class prediction exists for classification networks")
        xf = net._flat(x)
        self.probs = torch.empty((xf.shape[0], net._specs()[-1].in_out[1]), dtype=torch.float32, device=x.device)
        self.preds = torch.empty(xf.shape[0], dtype=torch.int64, device=x.device)
        super().__init__(net, x, torch.zeros(xf.shape[0], dtype=torch.int64, device=x.device), samples, capture=capture, stream=stream)

    def _enqueue(self):
        super()._enqueue()
        ops.mc_softmax_mean(self.logits, 1.0 / self.samples, out_probs=self.probs, out_preds=self.preds if self.world == 1 else None,
                            want_preds=self.world == 1)

    def replay(self):
        super().replay()
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.probs, op=dist.ReduceOp.SUM)
            self.preds.copy_(torch.argmax(self.probs, dim=1))
        return self.preds, self.probs


def elbo_many(net, x: torch.Tensor, target: torch.Tensor, samples: int, sigma: float = 1.0) -> torch.Tensor:
    """Forward-only ELBO terms of G independent minibatches in ONE launch per layer: x [G, B, ...], target [G, B]
    (or [G, B, out] for regression).  Returns float32 [G, 4] = per minibatch {sum_s log p | sum_s KL, sum_s log q | 0,
    sum_s nll, samples}, all-reduced over the ranks when sample sharding is on -- what G calls of
    sample_elbo / sample_elbo_lr under no_grad would give before their division by `samples`
    (networks.py:199-208, :217-224), each (minibatch, sample) pair with its own epsilon."""
    with torch.no_grad():
        ev = GraphedElbo(net, x, target, samples, sigma=sigma, capture=False, stacked=True)
        sums = ev.replay().clone().view(ev.G, 4)
    if ev.world > 1:
        import torch.distributed as dist
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums
