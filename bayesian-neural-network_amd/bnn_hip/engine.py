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


CAST_INPUT_MIN_SAMPLES = 8    # BBB: below this the extra launch costs more than it saves
# LR: from one sample on.  Its first layer squares every x fragment it loads, and fp32 x doubles the bytes each block
# pulls through its CU's L1 (21.5 us for that layer against 14.7 us for the wider second one): casting first wins
# even with the extra launch in the chain (one-sample evaluation 59.4 -> 55.3 us alone, 15.7 -> 14.6 us pipelined)
CAST_INPUT_MIN_SAMPLES_LR = 1
LR_SQUARES_MIN_SAMPLES = 8    # LR: carry x^2 (bf16) between layers from here on (the block-GEMM form streams it)
# BBB: K-sliced GEMM form (deterministic split-K + reduce kernel) for this range of samples per launch.
# Measured on MI355X it only ties the K-split kernel on the 1200-wide layers (29+5 us vs 32 us per layer
# at 8 samples) but wins on big layers below the plain GEMM form's threshold (4096x4096, 4 samples:
# 126 vs 175 us), so it is on for layers of >= SPLIT_MIN_WEIGHTS weights, everywhere with BNN_HIP_SPLITK=1.
SPLIT_MIN_SAMPLES, SPLIT_MAX_SAMPLES = 4, 24
SPLIT_MIN_WEIGHTS = 0 if os.environ.get("BNN_HIP_SPLITK", "0") == "1" else 4_000_000


# BBB, bf16 math: evaluations of at most this many MC samples take the split form -- ONE streaming launch samples the
# hidden layers' weights (bnn_bbb_sample_weights, the input cast riding on it), the hidden layers are then matmul-only
# launches over the sampled bf16 weights, the output layer + finalize stay fused.  0 = never.
PRESAMPLE_MAX_SAMPLES = int(os.environ.get("BNN_HIP_PRESAMPLE", "0"))


# BBB, bf16 math, one MC sample per evaluation, several evaluations per graph launch: the output layer + finalize of
# evaluation j share ONE launch with the first layer of evaluation j + 1 (bnn_bbb_final_next_fwd) -- the output layer
# is a few latency-bound blocks that otherwise hold the stream's chain of dependent launches for ~10 us.
PIPELINE_EVALS = os.environ.get("BNN_HIP_PIPELINE_EVALS", "1") != "0"
PIPELINE_DEPTH3 = os.environ.get("BNN_HIP_PIPELINE_DEPTH", "3") != "2"
PIPE_MAX_S = 3            # BBB evaluations of up to this many MC samples are pipelined (2: +11 %, 3: +2 %, 4: -1 %, 8: -24 %)


# differentiable sample_elbo*: the whole network as one autograd node (functional.ElboFn) when eps is drawn on
# chip; BNN_HIP_FUSED_ELBO=0 keeps one node per layer (the form the identical-eps parity path always uses)
FUSED_ELBO_NODE = os.environ.get("BNN_HIP_FUSED_ELBO", "1") != "0"


def use_split(fin: int, fout: int, n_samples: int) -> bool:
    return fout > 16 and fin * fout >= SPLIT_MIN_WEIGHTS and SPLIT_MIN_SAMPLES <= n_samples < SPLIT_MAX_SAMPLES
SIGMA_HOIST_MIN_SAMPLES = 24  # BBB: precompute sigma = softplus(rho) once per evaluation from here on
LR_PREPARE_MIN_SAMPLES = 24   # LR: prepare bf16 (M, sigma^2) fragments once per evaluation from here on


def lr_use_prepare(n_out: int, n_samples: int, batch: int) -> bool:
    """True when the LR throughput kernel (block GEMM) will run for this layer and enough samples share the
    prepared weights to pay for the extra pass (mirrors the launcher's geometry rule)."""
    return n_samples >= LR_PREPARE_MIN_SAMPLES and ((n_out + 63) // 64) * n_samples * ((batch + 127) // 128) >= 300


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
    math_mode = state.math
    hidden_dtype = torch.float32 if (differentiable or math_mode == L.MATH_F32) else torch.bfloat16
    h, h_sq = x, None
    any_lr = any(sp.lr for sp in layers)
    lr_sq = (not differentiable) and hidden_dtype == torch.bfloat16 and n_local >= LR_SQUARES_MIN_SAMPLES and any_lr
    if hidden_dtype == torch.bfloat16 and x.dtype == torch.float32 and n_local >= CAST_INPUT_MIN_SAMPLES:   # (eager calls
        # are host-bound: the one-sample LR cast of GraphedElbo would only add a launch here)
        # once per evaluation: every layer then streams 2-byte x (LR: and its elementwise square)
        if lr_sq:
            h, h_sq = ops.cast_bf16(x, want_sq=True)
        else:
            h = ops.cast_bf16(x)
    stats = []
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
                if lr_sq and want_stats and lr_use_prepare(sp.in_out[1], n_local, h.shape[-2]):
                    wfrag, ws_pre = ops.lr_prepare(*pd)
                out = ops.lr_linear_fwd(h, *pd, w_frag=wfrag, workspace=ws_pre, n_samples=n_local, sigma_p=call.prior.sigma_p, math_mode=math_mode,
                                        relu=sp.relu, y_dtype=call.y_dtype, eps_mode=eps_mode, eps_act=e_w, eps_b=e_b,
                                        seed=state.seed, layer_id=sp.layer_id, sample_offset=first_sample,
                                        want_kl=want_stats, x_sq=h_sq if lr_sq else None,
                                        out_sq=torch.empty((n_local, h.shape[-2], sp.in_out[1]), dtype=torch.bfloat16,
                                                           device=h.device) if want_sq else None)
                h_sq = out["y_sq"]
            else:
                kw = dict(n_samples=n_local, prior=call.prior, math_mode=math_mode, relu=sp.relu, y_dtype=call.y_dtype,
                          eps_mode=eps_mode, eps_w=e_w, eps_b=e_b, seed=state.seed, layer_id=sp.layer_id,
                          sample_offset=first_sample, want_stats=want_stats)
                if h.dtype == torch.bfloat16 and n_local >= SIGMA_HOIST_MIN_SAMPLES and eps_mode != L.EPS_ZERO and \
                        ((sp.in_out[1] + 63) // 64) * n_local >= 450:
                    kw["w_sigma"] = ops.softplus(pd[1])        # consumed by the throughput (GEMM) form only
                if h.dtype == torch.bfloat16 and use_split(sp.in_out[0], sp.in_out[1], n_local):
                    kw["split_scratch"] = ops.split_scratch(n_local, h.shape[-2], sp.in_out[1], h.device)
                if last and fin_kw is not None and want_stats:
                    # last layer + finalize in one launch (when the layer is a single feature tile)
                    out, fin = ops.bbb_final_fwd((h,) + pd, kw, dict(workspaces=stats, **fin_kw))
                    return out["y"], fin
                out = ops.bbb_linear_fwd(h, *pd, **kw)
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
                          scratch=None if local_reparam else ops.final_scratch(n_local, dev))
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
                                   fin_kw=fin_kw if not local_reparam else None)
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
    """One forward-only ELBO evaluation (all local MC samples: one launch per layer + the
    finalize launch) captured once as a hipGraph and replayed.  The Philox sample index has
    a device-resident part (`counter`) that the finalize kernel advances by the GLOBAL
    sample count, so every replay draws fresh epsilon without re-capturing.

    `replay()` returns the static float32[4] tensor {sum log p | sum KL, sum log q | 0,
    sum nll, n_local}: the vector a sharded job all-reduces."""

    def __init__(self, net, x: torch.Tensor, target: torch.Tensor, samples: int, sigma: float = 1.0,
                 capture: bool = True, counter_stride: int = 1, stream: Optional[torch.cuda.Stream] = None,
                 sums_ring=None, evals_per_replay: int = 1):
        """`evals_per_replay` E > 1: one replay runs E consecutive evaluations (one graph of E times
        the kernels; a hipGraph launch costs the host ~10 us + ~1 us per node, so short evaluations
        are launch-bound one at a time); `sums`/`out`/`logits` then hold the LAST one, the ring all.
        `counter_stride` > 1: this evaluator is one of several that run concurrently on their
        own streams and interleave the global MC sample index space (evaluator j of n starts j
        evaluations in and advances by n evaluations per replay).
        `sums_ring` = (base, ring_len, stride_floats): replay k deposits its 4-vector at
        base.view(-1)[(k % ring_len) * stride : +4] instead of a fixed tensor (device-side cursor), so
        a sharded job can all-reduce many evaluations' scalars with one collective."""
        self.net, self.samples, self.sigma = net, int(samples), float(sigma)
        self.stride = int(counter_stride)
        self.per_replay = max(1, int(evals_per_replay))
        self.stream = stream
        self.rank, self.world = dist_info()
        self.lo, self.n_local = shard_range(self.samples, self.rank, self.world)
        if self.n_local <= 0:
            raise ops.BnnHipError("GraphedElbo: this rank owns no MC sample (samples < world size)")
        self.specs = net._specs()
        self.lr = bool(net.local_reparam)
        dev = x.device
        self.x = net._flat(x).contiguous()
        self.target = target.contiguous()
        first = take_samples(0)
        self.counter = torch.tensor([first], dtype=torch.int32, device=dev)
        S = self.n_local
        B = self.x.shape[0]
        math_mode = state.math
        hid = torch.float32 if math_mode == L.MATH_F32 else torch.bfloat16
        self.bufs, self.ws = [], []
        for i, sp in enumerate(self.specs):
            fin, fout = sp.in_out
            last = i == len(self.specs) - 1
            self.bufs.append(torch.empty((S, B, fout), dtype=torch.float32 if last else hid, device=dev))
            self.ws.append(ops.lr_workspace(fout, dev) if self.lr else ops.bbb_workspace(S, fout, dev))
        keys = ("kl",) if self.lr else ("log_prior", "log_q")
        self.out = {k: torch.zeros(S, dtype=torch.float32, device=dev) for k in keys + ("nll",)}
        self.sums = torch.zeros(4, dtype=torch.float32, device=dev) if sums_ring is None else sums_ring[0]
        self.ring = None
        if sums_ring is not None:
            self.ring = (torch.zeros(1, dtype=torch.int32, device=dev), int(sums_ring[1]), int(sums_ring[2]))
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scratch = None if self.lr else ops.final_scratch(S, dev)
        nl = len(self.specs)
        self.presample = (not self.lr and hid == torch.bfloat16 and 0 < S <= PRESAMPLE_MAX_SAMPLES and nl >= 2 and
                          all(sp.in_out[0] % 8 == 0 for sp in self.specs[:-1]))
        self.wsamp = self.bsamp = None
        if self.presample:
            self.wsamp = [torch.empty((S, sp.in_out[1], sp.in_out[0]), dtype=torch.bfloat16, device=dev) for sp in self.specs[:-1]]
            self.bsamp = [torch.empty((S, sp.in_out[1]), dtype=torch.float32, device=dev) for sp in self.specs[:-1]]
            for i, sp in enumerate(self.specs[:-1]):
                self.ws[i] = ops.sample_workspace(S, sp.in_out[0], sp.in_out[1], dev)
        self.x16 = (torch.empty(self.x.shape, dtype=torch.bfloat16, device=dev)
                    if (hid == torch.bfloat16 and self.x.dtype == torch.float32 and
                        (self.presample or S >= (CAST_INPUT_MIN_SAMPLES_LR if self.lr else CAST_INPUT_MIN_SAMPLES))) else None)
        self.lr_sq = self.lr and self.x16 is not None and S >= LR_SQUARES_MIN_SAMPLES
        self.x16_sq = torch.empty(self.x.shape, dtype=torch.bfloat16, device=dev) if self.lr_sq else None
        self.bufs_sq = [torch.empty(b.shape, dtype=torch.bfloat16, device=dev) if (self.lr_sq and i < len(self.bufs) - 1)
                        else None for i, b in enumerate(self.bufs)]
        self.split = [ops.split_scratch(S, B, sp.in_out[1], dev)
                      if (not self.lr and hid == torch.bfloat16 and use_split(sp.in_out[0], sp.in_out[1], S)) else None
                      for i, sp in enumerate(self.specs)]
        self.wsigma = [torch.empty_like(sp.m.weight_rho.detach())
                      if (not self.lr and hid == torch.bfloat16 and S >= SIGMA_HOIST_MIN_SAMPLES and
                          ((sp.in_out[1] + 63) // 64) * S >= 450) else None for sp in self.specs]
        self.wfrag = [None] * len(self.specs)
        if self.lr_sq:
            self.wfrag = [torch.empty(L.load().bnn_lr_prepare_bytes(*sp.in_out) // 4, dtype=torch.float32, device=dev)
                          if lr_use_prepare(sp.in_out[1], S, B) else None for sp in self.specs]
        # software pipeline over the evaluations of one graph launch (see PIPELINE_EVALS): the first layer's statistics
        # workspace alternates, every evaluation has its own static sample offset and only the last finalize of a
        # replay advances the device counter, so an evaluation's first layer depends on nothing its predecessor writes
        self.pipelined = (PIPELINE_EVALS and self.per_replay > 1 and not self.lr and not self.presample and S <= PIPE_MAX_S and
                          hid == torch.bfloat16 and nl >= 2 and self.specs[-1].in_out[1] <= 16 and B <= 128 and
                          self.split[0] is None and self.wsigma[0] is None and self.x16 is None)
        self.ws0_alt = ops.bbb_workspace(S, self.specs[0].in_out[1], dev) if self.pipelined else None
        # three-layer nets go one step further: first layer of evaluation j+2, hidden layer of j+1 and output layer of j
        # in ONE launch (bnn_bbb_stage_fwd), activations and statistics of the two hidden layers buffered three deep
        self.pipe3 = self.pipelined and nl == 3 and PIPELINE_DEPTH3 and self.split[1] is None and self.wsigma[1] is None
        if self.pipe3:
            self.slot_bufs = [[self.bufs[i]] + [torch.empty_like(self.bufs[i]) for _ in range(2)] for i in range(2)]
            self.slot_ws = [[self.ws[i]] + [ops.bbb_workspace(S, self.specs[i].in_out[1], dev) for _ in range(2)]
                            for i in range(2)]
        # LR, three layers: the same three-deep pipeline (bnn_lr_stage_fwd); the finalize stays a launch of its own and
        # carries the input cast of a later evaluation
        # (one sample per evaluation only: from two samples on the stage cannot carry the finalize and measured 3-6 % slower
        # than one launch per layer with the cast riding on the finalize)
        self.lr_pipe3 = (PIPELINE_EVALS and PIPELINE_DEPTH3 and self.lr and self.per_replay > 1 and nl == 3 and S == 1 and
                         hid == torch.bfloat16 and self.x16 is not None and not self.lr_sq and
                         all(w is None for w in self.wfrag))
        if self.lr_pipe3:
            self.slot_bufs = [[self.bufs[i]] + [torch.empty_like(self.bufs[i]) for _ in range(2)] for i in range(3)]
            self.slot_ws = [[self.ws[i]] + [ops.lr_workspace(self.specs[i].in_out[1], dev) for _ in range(2)] for i in range(3)]
            self.x16_alt = torch.empty_like(self.x16)     # the cast of evaluation t + 1 rides beside the first layer of t
            self._last_slot = 0
        self.graph = None
        self._enqueue()                      # warm-up (also validates arguments eagerly)
        take_samples(self.samples)
        torch.cuda.synchronize()
        if self.stride > 1:                  # undo the warm-up's stride-sized advance: next index = first + S
            self.counter.fill_(first + self.samples)
            torch.cuda.synchronize()
        if self.ring is not None:            # the warm-up used slot 0
            self.ring[0].zero_()
            torch.cuda.synchronize()
        if capture:
            side = self.stream if self.stream is not None else torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    self._eager()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = g

    def _enqueue(self, skip_cast: bool = False, ride_cast: bool = False):
        """One evaluation.  LR with several evaluations per graph launch: the input cast of evaluation j + 1 rides on the
        finalize launch of evaluation j (`ride_cast`; the next call then passes `skip_cast`), one launch less on the chain."""
        math_mode = state.math
        h_sq = None
        if self.presample:
            hidden = self.specs[:-1]
            ops.bbb_sample_weights(
                [dict(w_mu=sp.m.weight_mu.detach(), w_rho=sp.m.weight_rho.detach(), b_mu=sp.m.bias_mu.detach(),
                      b_rho=sp.m.bias_rho.detach(), prior=sp.m._prior_spec, layer_id=sp.layer_id, workspace=self.ws[i],
                      w_out=self.wsamp[i], b_out=self.bsamp[i]) for i, sp in enumerate(hidden)],
                n_samples=self.n_local, seed=state.seed, sample_offset=self.lo, sample_counter=self.counter,
                cast=(self.x, self.x16) if self.x16 is not None else None)
            h = self.x16 if self.x16 is not None else self.x
        elif self.x16 is None:
            h = self.x
        elif skip_cast:
            h, h_sq = self.x16, (self.x16_sq if self.lr_sq else None)
        elif self.lr_sq:
            h, h_sq = ops.cast_bf16(self.x, out=self.x16, out_sq=self.x16_sq)
        else:
            h = ops.cast_bf16(self.x, out=self.x16)
        last = len(self.specs) - 1
        fin_kw = dict(layer_in=[sp.in_out[0] for sp in self.specs], layer_out=[sp.in_out[1] for sp in self.specs],
                      local_reparam=self.lr, prior=self.specs[0].m._prior_spec, n_samples=self.n_local,
                      target=self.target, mode=self.net.mode, nll_sigma=self.sigma, sample_counter=self.counter,
                      sample_counter_inc=self.samples * self.stride, out=self.out, sums=self.sums,
                      ticket=self.ticket, scratch=self.scratch, sums_ring=self.ring)
        pending = None
        for i, sp in enumerate(self.specs):
            p = tuple(t.detach() for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
            common = dict(n_samples=self.n_local, math_mode=math_mode, relu=sp.relu, y_dtype=self.bufs[i].dtype,
                          eps_mode=L.EPS_PHILOX, seed=state.seed, layer_id=sp.layer_id, sample_offset=self.lo,
                          sample_counter=self.counter, workspace=self.ws[i], out=self.bufs[i],
                          concurrency=self.stride)      # evaluators that run side by side: size launches for a share of the chip
            if self.presample and i < last:
                ops.bbb_sampled_matmul(h, self.wsamp[i], self.bsamp[i], n_samples=self.n_local, relu=sp.relu,
                                       y_dtype=self.bufs[i].dtype, out=self.bufs[i], concurrency=self.stride)
            elif self.lr:
                if self.wfrag[i] is not None:
                    ops.lr_prepare(*p, workspace=self.ws[i], out=self.wfrag[i])
                ops.lr_linear_fwd(h, *p, sigma_p=sp.m._prior_spec.sigma_p, want_kl=True, x_sq=h_sq,
                                  out_sq=self.bufs_sq[i], w_frag=self.wfrag[i], **common)
                h_sq = self.bufs_sq[i]
            elif i == last and pending is not None:
                # one sample: the last hidden layer, the output layer and the finalize in ONE launch
                ops.bbb_tail2_fwd(pending[0], pending[1], (h,) + p, dict(prior=sp.m._prior_spec, want_stats=True, **common),
                                  dict(workspaces=self.ws[:last - 1], **fin_kw))
            elif i == last:
                if self.wsigma[i] is not None:
                    ops.softplus(p[1], out=self.wsigma[i])
                ops.bbb_final_fwd((h,) + p, dict(prior=sp.m._prior_spec, want_stats=True, split_scratch=self.split[i],
                                                 w_sigma=self.wsigma[i], **common),
                                  dict(workspaces=self.ws[:last], **fin_kw))
            elif i == last - 1 and not self.presample and self.n_local == 1 and self.bufs[i].dtype == torch.bfloat16 and \
                    h.dtype == torch.bfloat16 and self.split[i] is None and self.wsigma[i] is None:
                pending = ((h,) + p, dict(prior=sp.m._prior_spec, want_stats=True, **common))   # launched with the last layer
            else:
                if self.wsigma[i] is not None:
                    ops.softplus(p[1], out=self.wsigma[i])
                ops.bbb_linear_fwd(h, *p, prior=sp.m._prior_spec, want_stats=True, split_scratch=self.split[i],
                                   w_sigma=self.wsigma[i], **common)
            h = self.bufs[i]
        if self.lr:
            ops.elbo_finalize(workspaces=self.ws, logits=h,
                              cast=(self.x, self.x16, self.x16_sq if self.lr_sq else None) if ride_cast else None, **fin_kw)

    def _enqueue_pipelined(self):
        """per_replay one-sample BBB evaluations as L0(e0) L1..(e0) [final(e0) + L0(e1)] L1..(e1) ... final(e_last)."""
        E, last = self.per_replay, len(self.specs) - 1
        inc = self.samples * self.stride                   # global MC indices one evaluation of this evaluator spans
        math_mode = state.math
        if self.pipe3:
            self._enqueue_pipelined3(E, inc, math_mode)
            return

        def layer_call(i, j):
            sp = self.specs[i]
            p = tuple(t.detach() for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
            ws = (self.ws[0], self.ws0_alt)[j & 1] if i == 0 else self.ws[i]
            h = self.x if i == 0 else self.bufs[i - 1]
            kw = dict(n_samples=self.n_local, math_mode=math_mode, relu=sp.relu, y_dtype=self.bufs[i].dtype,
                      eps_mode=L.EPS_PHILOX, seed=state.seed, layer_id=sp.layer_id, sample_offset=self.lo + j * inc,
                      sample_counter=self.counter, workspace=ws, out=self.bufs[i], concurrency=self.stride,
                      prior=sp.m._prior_spec, want_stats=True)
            return (h,) + p, kw

        for j in range(E):
            if j == 0:
                a0, k0 = layer_call(0, 0)
                ops.bbb_linear_fwd(*a0, **k0)
            for i in range(1, last):
                ai, ki = layer_call(i, j)
                ops.bbb_linear_fwd(*ai, **ki)
            al, kl = layer_call(last, j)
            fin_kw = dict(layer_in=[sp.in_out[0] for sp in self.specs], layer_out=[sp.in_out[1] for sp in self.specs],
                          local_reparam=False, prior=self.specs[0].m._prior_spec, n_samples=self.n_local,
                          target=self.target, mode=self.net.mode, nll_sigma=self.sigma, sample_counter=self.counter,
                          sample_counter_inc=E * inc if j == E - 1 else 0, out=self.out, sums=self.sums,
                          ticket=self.ticket, scratch=self.scratch, sums_ring=self.ring,
                          workspaces=[(self.ws[0], self.ws0_alt)[j & 1]] + self.ws[1:last])
            if j < E - 1:
                an, kn = layer_call(0, j + 1)
                ops.bbb_final_next_fwd(al, kl, fin_kw, an, kn)
            else:
                ops.bbb_final_fwd(al, kl, fin_kw)

    def _enqueue_pipelined3(self, E, inc, math_mode):
        """Launch t = {output layer + finalize of evaluation t-2, hidden layer of t-1, first layer of t}: E + 2 launches
        for E evaluations; evaluation j lives in buffer slot j % 3."""
        def layer_call(i, j):
            sp = self.specs[i]
            p = tuple(t.detach() for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
            slot = j % 3
            h = self.x if i == 0 else self.slot_bufs[i - 1][slot]
            out = self.slot_bufs[i][slot] if i < 2 else self.bufs[2]
            ws = self.slot_ws[i][slot] if i < 2 else self.ws[2]
            kw = dict(n_samples=self.n_local, math_mode=math_mode, relu=sp.relu, y_dtype=out.dtype, eps_mode=L.EPS_PHILOX,
                      seed=state.seed, layer_id=sp.layer_id, sample_offset=self.lo + j * inc, sample_counter=self.counter,
                      workspace=ws, out=out, concurrency=self.stride, prior=sp.m._prior_spec, want_stats=True)
            return (h,) + p, kw

        for t in range(E + 2):
            final = mid = first = None
            if t < E:
                first = layer_call(0, t)
            if 0 <= t - 1 < E:
                mid = layer_call(1, t - 1)
            if 0 <= t - 2 < E:
                j = t - 2
                al, kl = layer_call(2, j)
                fin_kw = dict(layer_in=[sp.in_out[0] for sp in self.specs], layer_out=[sp.in_out[1] for sp in self.specs],
                              local_reparam=False, prior=self.specs[0].m._prior_spec, n_samples=self.n_local,
                              target=self.target, mode=self.net.mode, nll_sigma=self.sigma, sample_counter=self.counter,
                              sample_counter_inc=E * inc if j == E - 1 else 0, out=self.out, sums=self.sums,
                              ticket=self.ticket, scratch=self.scratch, sums_ring=self.ring,
                              workspaces=[self.slot_ws[0][j % 3], self.slot_ws[1][j % 3]])
                final = (al, kl, fin_kw)
            ops.bbb_stage_fwd(final=final, mid=mid, first=first)

    def steady_state_stage(self):
        """For measurement (bench.py's roofline): a closure that enqueues ONE steady-state launch of the three-deep
        pipeline -- output layer + finalize of the evaluation in slot 0, hidden layer of the one in slot 1, first layer
        of the one in slot 2 -- without advancing the sample counter or the sums ring."""
        if self.lr_pipe3:
            math_mode = state.math

            def lcall(i, slot):
                sp = self.specs[i]
                p = tuple(t.detach() for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
                h = self.x16 if i == 0 else self.slot_bufs[i - 1][slot]
                out, ws = self.slot_bufs[i][slot], self.slot_ws[i][slot]
                return (h,) + p, dict(n_samples=self.n_local, sigma_p=sp.m._prior_spec.sigma_p, math_mode=math_mode,
                                      relu=sp.relu, y_dtype=out.dtype, eps_mode=L.EPS_PHILOX, seed=state.seed,
                                      layer_id=sp.layer_id, sample_offset=self.lo, sample_counter=self.counter, want_kl=True,
                                      workspace=ws, out=out, concurrency=self.stride)
            last, mid, first = lcall(2, 0), lcall(1, 1), lcall(0, 2)
            sums = torch.zeros(4, dtype=torch.float32, device=self.x.device)
            fin_kw = dict(workspaces=[self.slot_ws[i][1] for i in range(3)], logits=self.slot_bufs[2][1],
                          layer_in=[sp.in_out[0] for sp in self.specs], layer_out=[sp.in_out[1] for sp in self.specs],
                          local_reparam=True, prior=self.specs[0].m._prior_spec, n_samples=self.n_local, target=self.target,
                          mode=self.net.mode, nll_sigma=self.sigma, sample_counter=self.counter, sample_counter_inc=0,
                          out=self.out, sums=sums, ticket=self.ticket, cast=(self.x, self.x16_alt, None))
            return lambda: ops.lr_stage_fwd(last=last, mid=mid, first=first, fin_kw=fin_kw)
        if not self.pipe3:
            raise ops.BnnHipError("steady_state_stage: this evaluator is not three-deep pipelined")
        math_mode = state.math
        sums = torch.zeros(4, dtype=torch.float32, device=self.x.device)

        def call(i, slot):
            sp = self.specs[i]
            p = tuple(t.detach() for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
            h = self.x if i == 0 else self.slot_bufs[i - 1][slot]
            out = self.slot_bufs[i][slot] if i < 2 else self.bufs[2]
            ws = self.slot_ws[i][slot] if i < 2 else self.ws[2]
            return (h,) + p, dict(n_samples=self.n_local, math_mode=math_mode, relu=sp.relu, y_dtype=out.dtype,
                                  eps_mode=L.EPS_PHILOX, seed=state.seed, layer_id=sp.layer_id, sample_offset=self.lo,
                                  sample_counter=self.counter, workspace=ws, out=out, concurrency=self.stride,
                                  prior=sp.m._prior_spec, want_stats=True)

        al, kl = call(2, 0)
        fin_kw = dict(layer_in=[sp.in_out[0] for sp in self.specs], layer_out=[sp.in_out[1] for sp in self.specs],
                      local_reparam=False, prior=self.specs[0].m._prior_spec, n_samples=self.n_local, target=self.target,
                      mode=self.net.mode, nll_sigma=self.sigma, sample_counter=self.counter, sample_counter_inc=0,
                      out=self.out, sums=sums, ticket=self.ticket, scratch=self.scratch,
                      workspaces=[self.slot_ws[0][0], self.slot_ws[1][0]])
        mid, first = call(1, 1), call(0, 2)
        return lambda: ops.bbb_stage_fwd(final=(al, kl, fin_kw), mid=mid, first=first)

    def _enqueue_lr_pipelined(self):
        """LR: launch t = {output layer of evaluation t-2, hidden layer of t-1, first layer of t, finalize of t-3, input
        cast of t+1} (bnn_lr_stage_fwd): E + 3 launches for E evaluations, one per evaluation in steady state.  Evaluation
        j lives in buffer slot j % 3, its bf16 input in x16 buffer j % 2."""
        E = self.per_replay
        inc = self.samples * self.stride
        math_mode = state.math
        x16 = (self.x16, self.x16_alt)

        def layer_call(i, j):
            sp = self.specs[i]
            p = tuple(t.detach() for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
            slot = j % 3
            h = x16[j % 2] if i == 0 else self.slot_bufs[i - 1][slot]
            return (h,) + p, dict(n_samples=self.n_local, sigma_p=sp.m._prior_spec.sigma_p, math_mode=math_mode,
                                  relu=sp.relu, y_dtype=self.slot_bufs[i][slot].dtype, eps_mode=L.EPS_PHILOX, seed=state.seed,
                                  layer_id=sp.layer_id, sample_offset=self.lo + j * inc, sample_counter=self.counter,
                                  want_kl=True, workspace=self.slot_ws[i][slot], out=self.slot_bufs[i][slot],
                                  concurrency=self.stride)

        def fin_call(j, t):
            slot = j % 3
            rider = (self.x, x16[(t + 1) % 2], None) if t + 1 < E else None
            return dict(workspaces=[self.slot_ws[i][slot] for i in range(3)], logits=self.slot_bufs[2][slot],
                        layer_in=[sp.in_out[0] for sp in self.specs], layer_out=[sp.in_out[1] for sp in self.specs],
                        local_reparam=True, prior=self.specs[0].m._prior_spec, n_samples=self.n_local, target=self.target,
                        mode=self.net.mode, nll_sigma=self.sigma, sample_counter=self.counter,
                        sample_counter_inc=E * inc if j == E - 1 else 0, out=self.out, sums=self.sums, ticket=self.ticket,
                        sums_ring=self.ring, cast=rider)

        ops.cast_bf16(self.x, out=x16[0])                     # evaluation 0
        for t in range(E + 3):
            j = t - 3
            fin_kw = fin_call(j, t) if 0 <= j < E else None
            if fin_kw is None and t + 1 < E:                  # no finalize to carry the next cast yet
                ops.cast_bf16(self.x, out=x16[(t + 1) % 2])
            ops.lr_stage_fwd(last=layer_call(2, t - 2) if 0 <= t - 2 < E else None,
                             mid=layer_call(1, t - 1) if 0 <= t - 1 < E else None,
                             first=layer_call(0, t) if t < E else None, fin_kw=fin_kw)
        self._last_slot = (E - 1) % 3

    def _eager(self):
        if self.pipelined:
            self._enqueue_pipelined()
            return
        if self.lr_pipe3:
            self._enqueue_lr_pipelined()
            return
        ride = PIPELINE_EVALS and self.lr and self.x16 is not None and self.per_replay > 1
        for j in range(self.per_replay):
            self._enqueue(skip_cast=ride and j > 0, ride_cast=ride and j < self.per_replay - 1)

    def replay(self) -> torch.Tensor:
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self.graph.replay() if self.graph is not None else self._eager()
        elif self.graph is not None:
            self.graph.replay()
        else:
            self._eager()
        take_samples(self.samples * self.per_replay)     # keep the host-side counter in step
        return self.sums

    @property
    def logits(self) -> torch.Tensor:
        if getattr(self, "lr_pipe3", False):               # of the last evaluation of a replay
            return self.slot_bufs[2][self._last_slot]
        return self.bufs[-1]
