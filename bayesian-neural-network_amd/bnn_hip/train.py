"""One whole training step of the reference's loop (classification/class_task.py:66-79:
zero_grad -> sample_elbo -> loss.backward() -> optimiser.step()) captured as ONE hipGraph.

Eagerly that loop is launch-bound on MI355X (about 1 ms of host work around ~0.2 ms of kernels
at the MNIST configuration); replaying a graph removes the host from the step.  Everything
that varies between steps lives in device memory the graph reads: the minibatch (static
buffers), beta (a 0-dim tensor: class_task.py:70 computes it on the host per minibatch), the
Philox sample counter (so every replay draws fresh eps and the backward kernels regenerate
exactly the eps of that step's forward), Adam's step number and learning rate (FusedAdam
capturable mode; StepLR keeps working through FusedAdam.sync_lr()).
"""
from __future__ import annotations

import torch

from . import ops
from .optim import FusedAdam
from .runtime import state, take_samples


class GraphedTrainStep:
    def __init__(self, net, optimizer: FusedAdam, x: torch.Tensor, y: torch.Tensor, samples: int, sigma: float = 1.0,
                 warmup: int = 2):
        """`x`, `y`: an example minibatch (shape/dtype of every later one).  The warm-up steps run with
        the optimiser's learning rate forced to zero and its state restored afterwards, so building
        the graph leaves the model and the optimiser as they were."""
        if not all(g.get("capturable") for g in optimizer.param_groups):
            raise ops.BnnHipError("GraphedTrainStep needs FusedAdam(capturable=True)")
        if state.host_eps or any(sp.m._eps_stubbed() for sp in net._specs()):
            raise ops.BnnHipError("GraphedTrainStep draws eps on chip; host/injected eps cannot be captured")
        if state.shard_samples:
            raise ops.BnnHipError("GraphedTrainStep: shard MC samples outside the captured step")
        self.net, self.opt, self.samples, self.sigma = net, optimizer, int(samples), float(sigma)
        dev = x.device
        self.x, self.y = x.clone(), y.clone()
        self.beta = torch.zeros((), dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self._elbo = net.sample_elbo_lr if net.local_reparam else net.sample_elbo
        self.first = take_samples(0)

        # ---- warm-up on a side stream (allocator pools, lazy inits), then undo its effects
        params = [p for g in optimizer.param_groups for p in g["params"]]
        saved_p = [p.detach().clone() for p in params]
        saved_state = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in optimizer.state[p].items()}
                       for p in params if p in optimizer.state and len(optimizer.state[p])}
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
        for gi in optimizer._dev:
            steps = [int(saved_state[p]["step"]) for p in optimizer.param_groups[gi]["params"] if p in saved_state]
            optimizer._dev[gi][0].fill_(max(steps) if steps else 0)
        self.counter.zero_()
        torch.cuda.synchronize()

        # ---- capture
        self.graph = torch.cuda.CUDAGraph()
        state.device_counter = self.counter
        try:
            take_before = state.counter
            with torch.cuda.graph(self.graph):
                self.out = self._one_step()
            state.counter = take_before              # capture ran nothing: the indices are still unused
        finally:
            state.device_counter = None

    def _one_step(self):
        state.device_counter = self.counter
        try:
            self.opt.zero_grad(set_to_none=True)
            out = self._elbo(self.x, self.y, self.beta, self.samples, self.sigma)
            out[0].backward()
            self.opt.step()
            self.counter.add_(self.samples)          # the next step's MC samples: fresh Philox subsequences
        finally:
            state.device_counter = None
        return tuple(o.detach() for o in out)

    def step(self, x: torch.Tensor, y: torch.Tensor, beta: float):
        """One optimiser step on minibatch (x, y) with KL weight beta.  Returns the tuple
        sample_elbo* returns (static tensors: read them before the next call)."""
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        self.beta.fill_(float(beta))
        self.opt.sync_lr()
        self.graph.replay()
        take_samples(self.samples)
        return self.out
