"""One whole training step of the reference's loop (classification/class_task.py:66-79:
zero_grad -> sample_elbo -> loss.backward() -> optimiser.step()) captured as ONE hipGraph.

Eagerly that loop is launch-bound on MI355X (about 1 ms of host work around ~0.2 ms of kernels
at the MNIST configuration); replaying a graph removes the host from the step.  Everything
that varies between steps lives in device memory the graph reads: the minibatch (static
buffers), beta (a 0-dim tensor: class_task.py:70 computes it on the host per minibatch), the
Philox sample counter (so every replay draws fresh eps and the backward kernels regenerate
exactly the eps of that step's forward), Adam's step number and learning rate (FusedAdam
capturable mode; StepLR keeps working through FusedAdam.sync_lr()).

Two forms of the captured step: `autograd=True` records exactly what the eager loop runs
(torch.autograd over the HIP-kernel Functions); the default chains the same kernels by hand —
layer forwards, bnn_elbo_finalize, bnn_elbo_loss_nll_bwd (loss, backward seeds, d nll / d logits), the
layer backwards (each layer's ReLU mask applied by the input-gradient kernel above it), bnn_adam_step — without autograd's bookkeeping kernels (zero fills, gradient
accumulation, per-layer scalar launches): about a third fewer microseconds per step.
"""
from __future__ import annotations

import os

import torch

from . import _lib as L
from . import ops
from .optim import FusedAdam
from .runtime import state, take_samples


# BBB, bf16 math, hand-chained step: sample every layer's weights ONCE per step in one streaming launch
# (bnn_bbb_sample_weights), run the forward as matmul-only launches over them and reuse them in the input-gradient
# launches, which then need no generator either (the transposed generator draws a whole Philox group per weight).
# The weight-gradient kernels still regenerate eps.  BNN_HIP_TRAIN_PRESAMPLE=0: sampling fused into every launch.
TRAIN_PRESAMPLE = os.environ.get("BNN_HIP_TRAIN_PRESAMPLE", "1") != "0"
TRAIN_BF16_FORWARD_INPUTS = True
TRAIN_TRANSPOSED_INPUT_GRAD = True


class GraphedTrainStep:
    def __init__(self, net, optimizer: FusedAdam, x: torch.Tensor, y: torch.Tensor, samples: int, sigma: float = 1.0,
                 warmup: int = 2, autograd: bool = False, data_parallel: bool = False, grad_dtype: torch.dtype = torch.float32):
        """`x`, `y`: an example minibatch (shape/dtype of every later one).  The warm-up steps are real
        updates; parameters, Adam moments, the (host or device) step count and the sample counter are snapshotted
        before and restored afterwards, so building the graph -- also a second one on an already trained
        optimiser -- leaves the model and the optimiser as they were.

        `data_parallel`: every rank of the default torch.distributed group (RCCL on GPUs) holds a replica
        and feeds its own minibatch; the backward kernels write all gradients into ONE flat fp32
        bucket (2 x 2.4 M floats at the MNIST config), sum-all-reduced in TWO pieces between three captured
        graphs: forward + backward down to layer 1 | all-reduce of the upper layers' gradients (asynchronous, on
        the collective's own stream) beside the backward of layer 0 | all-reduce of layer 0's gradients | Adam.
        The backward seeds carry the 1/ranks, rank r draws MC-sample indices [first + r*S, first + (r+1)*S) of
        every step.  Replicas must start identical (`broadcast_parameters`).

        `grad_dtype=torch.bfloat16` (data_parallel only): the bucket is rounded to bf16 (one cast launch per piece, in the
        captured graphs) and ALL-REDUCED IN 2-BYTE ELEMENTS -- 9.6 instead of 19.2 MB per step at the MNIST config, the
        ring's per-link time halves -- and Adam reads the bf16 sums (bnn_adam_args.grad_dtype).  The backward still
        accumulates in fp32; what is rounded is each rank's finished gradient (2^-9 relative) and the collective's sums."""
        if not all(g.get("capturable") for g in optimizer.param_groups):
            raise ops.BnnHipError("GraphedTrainStep needs FusedAdam(capturable=True)")
        if state.host_eps or any(sp.m._eps_stubbed() for sp in net._specs()):
            raise ops.BnnHipError("GraphedTrainStep draws eps on chip; host/injected eps cannot be captured")
        if state.shard_samples:
            raise ops.BnnHipError("GraphedTrainStep: shard MC samples outside the captured step")
        self.net, self.opt, self.samples, self.sigma = net, optimizer, int(samples), float(sigma)
        self.autograd = bool(autograd)
        self.rank, self.world = 0, 1
        if data_parallel:
            import torch.distributed as dist
            if self.autograd:
                raise ops.BnnHipError("data_parallel uses the kernel chain (autograd=False)")
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dp = bool(data_parallel)
        if grad_dtype not in (torch.float32, torch.bfloat16) or (grad_dtype != torch.float32 and not data_parallel):
            raise ops.BnnHipError("grad_dtype: float32, or bfloat16 with data_parallel=True")
        self.grad16 = grad_dtype == torch.bfloat16
        dev = x.device
        self.x, self.y = x.clone(), y.clone()
        # bf16 math: the forward launches read bf16 activations (the staging launch casts the minibatch, each hidden
        # layer also writes its output in bf16), the backward the fp32 ones
        self.x16 = (self.x.to(torch.bfloat16) if (TRAIN_BF16_FORWARD_INPUTS and not autograd and state.math == L.MATH_BF16 and
                                                  x.dtype == torch.float32) else None)
        self.beta = torch.zeros((), dtype=torch.float32, device=dev)
        # ONE device MC-sample counter per optimiser, shared by every step object built on it (full batches and the 96-row
        # last MNIST batch are two objects on one net): a step draws global sample index base + counter (+ rank * S), the
        # counter advances inside Adam's launch, so alternating objects never replay an index.  `mirror` is the host's copy.
        sh = getattr(optimizer, "_sample_words", None)
        if sh is None or sh["counter"].device != dev:
            sh = optimizer._sample_words = dict(counter=torch.zeros(1, dtype=torch.int32, device=dev), base=take_samples(0), mirror=0)
        self._shared = sh
        self.counter, self.base = sh["counter"], sh["base"]
        self.fin_ticket = torch.zeros(1, dtype=torch.int32, device=dev)   # lets the samples be finalized in parallel
        self.fin_scratch = ops.final_scratch(self.samples, dev)   # hand-off words of the row-split / K-sliced output layer
        self._lr_split = {}                                       # layer index -> K3s scratch (owned here: zeroed once, outside capture)
        self._lr_rider_bufs, self._lr_rider_job = None, None      # the output layer's prepared operands + KL workspace (bnn_lr_rider)
        # one flat gradient bucket (each slice 256-byte aligned); p.grad are views of it
        self.params = [p for sp in net._specs() for p in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho)]
        offs, tot = [], 0
        for p in self.params:
            offs.append(tot)
            tot += (p.numel() + 63) // 64 * 64
        self.bucket = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.grad_views = [self.bucket[o:o + p.numel()].view(p.shape) for o, p in zip(offs, self.params)]
        self.bucket_cut = offs[4] if len(offs) > 4 else 0      # first element of layer 1's gradients (layer 0 = 4 tensors)
        self.bucket16 = torch.zeros(tot, dtype=torch.bfloat16, device=dev) if self.grad16 else None
        self.grad16_of = ({p: self.bucket16[o:o + p.numel()].view(p.shape) for o, p in zip(offs, self.params)}
                          if self.grad16 else None)
        self._elbo = net.sample_elbo_lr if net.local_reparam else net.sample_elbo
        specs = net._specs()
        self.presample = (TRAIN_PRESAMPLE and not self.autograd and not net.local_reparam and state.math == L.MATH_BF16 and
                          all(sp.in_out[0] % 8 == 0 for sp in specs) and len(specs) <= L.SAMPLE_MAX_LAYERS)
        if self.presample:
            S = self.samples
            self.wsamp = [torch.empty((S, sp.in_out[1], sp.in_out[0]), dtype=torch.bfloat16, device=dev) for sp in specs]
            self.bsamp = [torch.empty((S, sp.in_out[1]), dtype=torch.float32, device=dev) for sp in specs]
            self.wstat = [ops.sample_workspace(S, sp.in_out[0], sp.in_out[1], dev) for sp in specs]
            # hidden layers above the first: their forward launch also leaves the sampled weights transposed, and their
            # input gradient runs as that same matmul-only launch over them (no 2-byte gathers along the reduction)
            self.wsamp_t = [torch.empty((S, sp.in_out[0], sp.in_out[1]), dtype=torch.bfloat16, device=dev)
                            if (TRAIN_TRANSPOSED_INPUT_GRAD and 0 < i < len(specs) - 1 and sp.in_out[1] % 8 == 0) else None
                            for i, sp in enumerate(specs)]
        self._sync_counter()              # indices other code drew since the last step (evaluations) are skipped, not reused

        # ---- warm-up on a side stream (allocator pools, lazy inits), then undo its effects
        params = [p for g in optimizer.param_groups for p in g["params"]]
        saved_p = [p.detach().clone() for p in params]
        saved_state = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in optimizer.state[p].items()}
                       for p in params if p in optimizer.state and len(optimizer.state[p])}
        # a capturable FusedAdam counts its steps in device words (state[p]["step"] is not written during training):
        # snapshot them, so that a step object built on an already trained optimiser (another batch shape, a
        # re-capture) leaves the bias-correction step where it was
        saved_dev_step = {gi: d[0].clone() for gi, d in getattr(optimizer, "_dev", {}).items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        host_counter = state.counter
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._one_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        state.counter = host_counter
        with torch.no_grad():
            for p, q in zip(params, saved_p):
                p.copy_(q)
            for p in params:
                st = optimizer.state[p]
                if p in saved_state:
                    for k, v in saved_state[p].items():
                        st[k].copy_(v) if torch.is_tensor(v) else st.__setitem__(k, v)
                else:
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()
        for gi in getattr(optimizer, "_dev", {}):
            if gi in saved_dev_step:
                optimizer._dev[gi][0].copy_(saved_dev_step[gi])
            else:                                        # the warm-up created the device words: start from the host-side step
                steps = [int(saved_state[p]["step"]) for p in optimizer.param_groups[gi]["params"] if p in saved_state]
                optimizer._dev[gi][0].fill_(max(steps) if steps else 0)
        self._set_counter(sh["mirror"])    # the warm-up's steps advanced it
        torch.cuda.synchronize()

        # ---- capture
        self.graph = torch.cuda.CUDAGraph()
        self.graph_update = None
        self.graph_bwd0 = None
        state.device_counter = self.counter
        try:
            take_before = state.counter
            state.counter = self.base                # what the autograd path's take_samples bakes into the graph
            if self.dp:
                with torch.cuda.graph(self.graph):
                    with torch.no_grad():
                        self.out = self._chain(stop_above_layer0=True)
                self.graph_bwd0 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_bwd0, pool=self.graph.pool()):
                    with torch.no_grad():
                        self._chain_layer0()
                self.graph_update = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_update, pool=self.graph.pool()):
                    self._update()
            else:
                with torch.cuda.graph(self.graph):
                    self.out = self._one_step()
            state.counter = take_before              # capture ran nothing: the indices are still unused
        finally:
            state.device_counter = None

    def _lr_rider(self, i, specs, hin):
        """bnn_lr_rider of the output layer, attached to the last hidden LR layer's launch when that launch is expected to run
        K-sliced (elsewhere the preparation would be a launch of its own): the final launch (K3r) then parks nothing."""
        from .engine import lr_kslice_expected
        last = specs[-1]
        if (i != len(specs) - 2 or not last.lr or last.in_out[1] > 16 or hin.dtype != torch.bfloat16 or self.samples > 16 or
                hin.shape[-2] > 128 or self._lr_scratch(i, specs[i], hin) is None or
                not lr_kslice_expected(*specs[i].in_out, self.samples, hin.shape[-2])):
            return None
        if self._lr_rider_bufs is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            self._lr_rider_bufs = (torch.empty(L.load().bnn_lr_prepare_bytes(*last.in_out) // 4, dtype=torch.float32, device=hin.device),
                                   ops.lr_workspace(last.in_out[1], hin.device))
        m = last.m
        self._lr_rider_job = dict(w_mu=m.weight_mu.detach(), w_rho=m.weight_rho.detach(), b_mu=m.bias_mu.detach(), b_rho=m.bias_rho.detach(),
                                  w_frag=self._lr_rider_bufs[0], workspace=self._lr_rider_bufs[1])
        return self._lr_rider_job

    def _lr_scratch(self, i, sp, hin):
        """K3s scratch of LR hidden layer i (None when the launch could not use one): allocated and zeroed at the first
        forward outside capture (construction warms the step up before it captures); the kernel leaves its counters zero."""
        from .engine import lr_use_split, lr_unit_samples
        shared = hin.dim() == 2                     # the step's samples on one minibatch: the first layer's products are made once
        if hin.dtype != torch.bfloat16 or not lr_use_split(sp.in_out[1], self.samples, hin.shape[-2], shared):
            return None
        if i not in self._lr_split:
            if torch.cuda.is_current_stream_capturing():
                return None
            self._lr_split[i] = ops.lr_split_scratch(lr_unit_samples(self.samples, shared), hin.shape[-2], sp.in_out[1], hin.device)
        return self._lr_split[i]

    def _chain(self, stop_above_layer0: bool = False):
        """zero_grad -> sample_elbo -> backward, as a hand-made chain of the C-ABI kernels (no autograd).
        Leaves the gradients in p.grad and returns what sample_elbo* returns."""
        net, S = self.net, self.samples
        self._lr_rider_job = None
        specs = net._specs()
        lr = bool(net.local_reparam)
        h = net._flat(self.x)
        h16 = net._flat(self.x16) if self.x16 is not None else None     # what the forward launches read, when present
        first = (self.base + self.rank * S) & 0xFFFFFFFF          # + the shared device counter, added by every launch
        saved, wss = [], []
        self._wt_ready = {}                        # layers whose forward launch of this step left transposed weights
        if self.presample:
            ops.bbb_sample_weights(
                [dict(w_mu=sp.m.weight_mu.detach(), w_rho=sp.m.weight_rho.detach(), b_mu=sp.m.bias_mu.detach(),
                      b_rho=sp.m.bias_rho.detach(), prior=sp.m._prior_spec, layer_id=sp.layer_id, workspace=self.wstat[i],
                      w_out=self.wsamp[i], b_out=self.bsamp[i]) for i, sp in enumerate(specs)],
                n_samples=S, seed=state.seed, sample_offset=first, sample_counter=self.counter)
        fin = None
        fin_kw = dict(layer_in=[sp.in_out[0] for sp in specs], layer_out=[sp.in_out[1] for sp in specs], local_reparam=lr,
                      prior=specs[0].m._prior_spec, n_samples=S, target=self.y, mode=net.mode, nll_sigma=self.sigma,
                      ticket=self.fin_ticket if S > 1 else None)
        for i, sp in enumerate(specs):
            p = tuple(t.detach() for t in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho))
            if not lr and i == len(specs) - 1 and self.presample:
                # output layer + finalize in one launch over the weights the sampling launch drew: split by batch rows
                # when it is a single feature tile (K1r), else matmul + finalize
                res, fin = ops.bbb_final_fwd((h16 if h16 is not None else h, None, None, None, None),
                                             dict(n_samples=S, prior=sp.m._prior_spec, math_mode=state.math, relu=sp.relu,
                                                  y_dtype=torch.float32, eps_mode=L.EPS_ZERO, want_stats=False,
                                                  w_sampled=self.wsamp[i], b_sampled=self.bsamp[i]),
                                             dict(workspaces=wss + [self.wstat[i]], scratch=self.fin_scratch,
                                                  loss=dict(beta=self.beta, total_samples=S, grad_scale=1.0 / self.world), **fin_kw))
                saved.append((h, res["y"], None, p, None))
                h = res["y"]
                continue
            if not lr and i == len(specs) - 1:
                # output layer + finalize in one launch (it samples its own few weights)
                res, fin = ops.bbb_final_fwd((h16 if h16 is not None else h,) + p, dict(n_samples=S, prior=sp.m._prior_spec, math_mode=state.math,
                                                            relu=sp.relu, y_dtype=torch.float32, eps_mode=L.EPS_PHILOX,
                                                            seed=state.seed, layer_id=sp.layer_id, sample_offset=first,
                                                            sample_counter=self.counter, want_stats=True),
                                             dict(workspaces=wss, scratch=self.fin_scratch, **fin_kw))
                saved.append((h, res["y"], None, p, None))
                h = res["y"]
                continue
            if self.presample:
                y = ops.bbb_sampled_matmul(h16 if h16 is not None else h, self.wsamp[i], self.bsamp[i], n_samples=S, relu=sp.relu,
                                           y_dtype=torch.float32, want_y16=self.x16 is not None,
                                           wt_out=self.wsamp_t[i] if h16 is not None else None)
                self._wt_ready[i] = h16 is not None and self.wsamp_t[i] is not None
                y, y16 = y if self.x16 is not None else (y, None)
                saved.append((h, y, None, p, None))
                wss.append(self.wstat[i])
                h, h16 = y, y16
                continue
            common = dict(n_samples=S, math_mode=state.math, relu=sp.relu, y_dtype=torch.float32, eps_mode=L.EPS_PHILOX,
                          seed=state.seed, layer_id=sp.layer_id, sample_offset=first, sample_counter=self.counter)
            hin = h16 if h16 is not None else h
            if sp.lr and i == len(specs) - 1:
                # output layer + finalize + loss tail through bnn_lr_final_fwd: one launch when the layer is narrow and its
                # input is bf16 (K3r; it also saves the variance for the backward), else layer, finalize, loss launches
                rd = self._lr_rider_job
                ws_last = rd["workspace"] if rd is not None else ops.lr_workspace(sp.in_out[1], h.device)
                out, fin = ops.lr_final_fwd((hin,) + p, dict(sigma_p=sp.m._prior_spec.sigma_p, want_kl=True, want_v=True,
                                                             workspace=ws_last, w_frag=rd["w_frag"] if rd is not None else None, **common),
                                            dict(workspaces=wss + [ws_last], scratch=self.fin_scratch,
                                                 loss=dict(beta=self.beta, total_samples=S, grad_scale=1.0 / self.world), **fin_kw))
                saved.append((h, out["y"], out.get("v"), p, None))
                h = out["y"]
                continue
            if sp.lr:
                # a hidden layer's ReLU mask is applied by the input-gradient launch of the layer above, so its backward needs
                # no mask pass: the forward saves eps / (2 sqrt(v)) instead of v and the backward no preparation launch
                hf = i < len(specs) - 1
                out = ops.lr_linear_fwd(hin, *p, sigma_p=sp.m._prior_spec.sigma_p, want_kl=True, want_v=not hf, want_hfac=hf,
                                        want_y16=self.x16 is not None and i < len(specs) - 1,
                                        split_scratch=self._lr_scratch(i, sp, hin) if hf else None,
                                        rider=self._lr_rider(i, specs, hin), **common)
            else:
                out = ops.bbb_linear_fwd(hin, *p, prior=sp.m._prior_spec, want_stats=True,
                                         want_y16=self.x16 is not None and i < len(specs) - 1, **common)
            saved.append((h, out["y"], out.get("v"), p, out.get("hfac")))
            wss.append(out["workspace"])
            h, h16 = out["y"], out.get("y16")
        if fin is None:
            fin = ops.elbo_finalize(workspaces=wss, logits=h, **fin_kw)
        # loss, backward seeds and d nll / d logits in one launch (the NLL seed is the constant 1 / (S ranks))
        if "loss" in fin:                   # the final launch did it
            out4, g_a, g_b, g_kl3, g = fin["loss"]
        else:
            out4, g_a, g_b, g_kl3, g = ops.elbo_loss_nll_bwd(fin["kl"] if lr else fin["log_prior"], None if lr else fin["log_q"],
                                                             fin["nll"], self.beta, S, lr, h, self.y, net.mode, self.sigma,
                                                             grad_scale=1.0 / self.world)
        top = len(specs) - 1
        self._bwd_state = (specs, saved, g_a, g_b, g_kl3, first, lr, top)
        self._bwd_g = g
        self._bwd_g16 = None
        for i in reversed(range(1 if stop_above_layer0 else 0, len(specs))):
            self._backward_layer(i)
        if self.grad16:                                   # what the collective moves: the finished gradients in bf16
            if stop_above_layer0:
                ops.cast_bf16(self.bucket[self.bucket_cut:], out=self.bucket16[self.bucket_cut:])
            else:
                ops.cast_bf16(self.bucket, out=self.bucket16)
        if lr:
            return out4[0:1], out4[1], out4[3:4]
        return out4[0:1], out4[1], out4[2], out4[3:4]

    def _chain_layer0(self):
        """The rest of the backward (layer 0's weight gradients): in data-parallel steps it runs beside the all-reduce of
        the upper layers' gradients."""
        self._backward_layer(0)
        if self.grad16:
            ops.cast_bf16(self.bucket[:self.bucket_cut], out=self.bucket16[:self.bucket_cut])

    def _backward_layer(self, i: int):
        """Backward of layer i (weight gradients into the bucket; input gradient for the layer below)."""
        specs, saved, g_a, g_b, g_kl3, first, lr, top = self._bwd_state
        S, g, sp = self.samples, self._bwd_g, specs[i]
        xin, y, v, p, hfac = saved[i]       # (layer input, output, LR: saved variance | saved eps / (2 sqrt(v)))
        # layer i's ReLU mask is applied by layer i+1's input-gradient kernel (its x IS layer i's output),
        # so only a top layer with a ReLU masks its own gy
        own_relu = sp.relu and i == top
        kw = dict(n_samples=S, relu=own_relu, eps_mode=L.EPS_PHILOX, seed=state.seed, layer_id=sp.layer_id,
                  sample_offset=first, sample_counter=self.counter, want_gx=i > 0, out=self.grad_views[4 * i:4 * i + 4],
                  gx_relu_mask=i > 0 and specs[i - 1].relu)
        if sp.lr:
            grads = ops.lr_linear_bwd(xin, g, y if own_relu else None, v, *p, sigma_p=sp.m._prior_spec.sigma_p, g_kl=g_kl3,
                                      math_mode=state.math, hfac=None if own_relu else hfac, **kw)
        else:
            wt = self.wsamp_t[i] if (self.presample and i > 0 and self._wt_ready.get(i)) else None
            below_t = bool(self.presample and i > 1 and self._wt_ready.get(i - 1))     # the layer below reads g_x in bf16
            grads = ops.bbb_linear_bwd(xin, g, y if own_relu else None, *p, prior=sp.m._prior_spec, math_mode=state.math,
                                       g_log_prior=g_a, g_log_q=g_b,
                                       w_sampled=self.wsamp[i] if (self.presample and i > 0) else None,
                                       w_sampled_t=wt, gy16=self._bwd_g16 if wt is not None else None, want_gx16=below_t, **kw)
            self._bwd_g16 = grads[5] if below_t else None
        sp.m.weight_mu.grad, sp.m.weight_rho.grad, sp.m.bias_mu.grad, sp.m.bias_rho.grad = grads[:4]
        self._bwd_g = grads[4]

    def _set_counter(self, value: int):
        value &= 0xFFFFFFFF
        self.counter.fill_(value - (1 << 32) if value >= (1 << 31) else value)     # the kernels add it as a uint32
        self._shared["mirror"] = value

    def _sync_counter(self):
        """Before a step: the device counter must equal the host's count of indices drawn since `base` (eager evaluations
        between steps draw from the host counter); one small fill when they differ, nothing otherwise."""
        want = (state.counter - self.base) & 0xFFFFFFFF
        if want != self._shared["mirror"]:
            self._set_counter(want)

    def _update(self):
        with torch.no_grad():
            # the step's MC-sample counter advances inside Adam's launch (after the backward re-read it); set per call:
            # another step object of this optimiser may use another increment
            self.opt.bump_after_step(self.counter, self.samples * self.world)
            self.opt.step(grads=self.grad16_of)

    def _allreduce_upper(self):
        """Layers 1..: issued right behind the graph that produced them, asynchronously (the collective runs on its own
        stream and waits for the work queued so far), so that layer 0's backward overlaps it."""
        import torch.distributed as dist
        if self.grad16:
            return dist.all_reduce(self.bucket16[self.bucket_cut:], op=dist.ReduceOp.SUM, async_op=True)
        return dist.all_reduce(self.bucket[self.bucket_cut:], op=dist.ReduceOp.SUM, async_op=True)   # seeds carry 1 / ranks

    def _allreduce_layer0(self):
        import torch.distributed as dist
        if self.grad16:
            return dist.all_reduce(self.bucket16[:self.bucket_cut], op=dist.ReduceOp.SUM, async_op=True)
        return dist.all_reduce(self.bucket[:self.bucket_cut], op=dist.ReduceOp.SUM, async_op=True)

    def _one_step(self):
        if not self.autograd:
            with torch.no_grad():
                out = self._chain()
            if self.dp:
                for w in (self._allreduce_upper(), self._allreduce_layer0()):
                    w.wait()
            self._update()
            return out
        state.device_counter = self.counter
        try:
            self.opt.zero_grad(set_to_none=True)
            out = self._elbo(self.x, self.y, self.beta, self.samples, self.sigma)
            out[0].backward()
            self.opt.bump_after_step(self.counter, self.samples * self.world)
            self.opt.step()                          # also advances self.counter: the next step draws fresh eps
        finally:
            state.device_counter = None
        return tuple(o.detach() for o in out)

    def step(self, x: torch.Tensor, y: torch.Tensor, beta: float):
        """One optimiser step on minibatch (x, y) with KL weight beta.  Returns the tuple
        sample_elbo* returns (static tensors: read them before the next call)."""
        if (x.is_cuda and y.is_cuda and x.dtype == self.x.dtype and y.dtype == self.y.dtype and x.is_contiguous()
                and y.is_contiguous() and x.numel() == self.x.numel() and y.numel() == self.y.numel()):
            cast = self.x16 if (self.x16 is not None and (x.numel() * 4) % 16 == 0 and (y.numel() * y.element_size()) % 16 == 0 and
                                (x.data_ptr() | y.data_ptr()) % 16 == 0) else None
            ops.stage_inputs(x, self.x, y, self.y, self.beta, float(beta), cast0=cast)       # one launch instead of three
            if cast is None and self.x16 is not None:
                self.x16.copy_(self.x)
        else:
            self.x.copy_(x, non_blocking=True)
            if self.x16 is not None:
                self.x16.copy_(x, non_blocking=True)
            self.y.copy_(y, non_blocking=True)
            self.beta.fill_(float(beta))
        self.opt.sync_lr()
        self._sync_counter()
        self.graph.replay()
        if self.dp:
            w_hi = self._allreduce_upper()              # beside ...
            self.graph_bwd0.replay()                    # ... the backward of layer 0
            w_lo = self._allreduce_layer0()
            w_hi.wait()
            w_lo.wait()
            self.graph_update.replay()
        take_samples(self.samples * self.world)
        self._shared["mirror"] = (self._shared["mirror"] + self.samples * self.world) & 0xFFFFFFFF
        return self.out


def broadcast_parameters(net, src: int = 0):
    """Make every rank's replica identical to rank `src`'s (call once before data-parallel training)."""
    import torch.distributed as dist
    with torch.no_grad():
        for p in net.parameters():
            dist.broadcast(p, src=src)
