"""F2: fused Adam for the (mu, rho) parameters — a drop-in for the `torch.optim.Adam` the
reference's trainers build (classification/class_task.py:60, regression/reg_task.py:53,
reinforcement_learning/bandits.py:36): same constructor keys, same update rule (torch's
`_single_tensor_adam`), `param_groups[i]['lr']` stays the knob `StepLR` turns
(class_task.py:61).  One HIP launch updates every parameter tensor (bnn_adam_step).

`capturable=True` keeps the step number and the learning rate in device words so that a
captured hipGraph of the whole training step (train.GraphedTrainStep) advances by itself.
"""
from __future__ import annotations

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=capturable))
        self._dev = {}            # group index -> (step int32[1], lr float32[1], last lr written, ticket int32[1])
        self._bump = None         # (uint32 device word, increment): advanced inside the step's launch

    def bump_after_step(self, counter: torch.Tensor, inc: int):
        """Capturable optimisers only: the (first group's) update launch also adds `inc` to the device word
        `counter` -- train.GraphedTrainStep's MC-sample counter, advanced once the backward has read it."""
        self._bump = (counter, int(inc))

    def _group_dev(self, gi, group, device):
        if gi not in self._dev:
            steps = [int(self.state[p]["step"]) for p in group["params"] if p in self.state and "step" in self.state[p]]
            self._dev[gi] = [torch.tensor([max(steps) if steps else 0], dtype=torch.int32, device=device),
                             torch.tensor([group["lr"]], dtype=torch.float32, device=device), group["lr"],
                             torch.zeros(16, dtype=torch.int32, device=device)]
        return self._dev[gi]

    def sync_lr(self):
        """Push param_groups[i]['lr'] (what an lr scheduler changed) into the device words a captured
        graph reads.  Call outside capture, before replaying."""
        for gi, group in enumerate(self.param_groups):
            if gi in self._dev and self._dev[gi][2] != group["lr"]:
                self._dev[gi][1].fill_(group["lr"])
                self._dev[gi][2] = group["lr"]

    def device_step(self, gi: int = 0) -> int:
        return int(self._dev[gi][0].item()) if gi in self._dev else 0

    def state_dict(self):
        """torch's format.  A capturable group counts its steps in a device word: mirror it into state[p]['step'] first,
        so that a checkpoint resumes the bias correction where training stopped (one device read per group)."""
        for gi, group in enumerate(self.param_groups):
            if gi in self._dev:
                step = self.device_step(gi)
                for p in group["params"]:
                    if p in self.state and len(self.state[p]):
                        self.state[p]["step"] = step
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """torch's format.  Everything a captured training step (train.GraphedTrainStep) holds the ADDRESS of survives the
        load: the device words of a capturable group are updated in place, and so are its moment tensors --
        torch.optim.Optimizer.load_state_dict replaces state[p]['exp_avg'] / ['exp_avg_sq'] with new tensors, which would
        leave the graph's replays updating the old (freed) buffers and ignoring the loaded moments.  The loaded values are
        copied into the tensors that existed before the load, and state[p] points at those again.  Groups that have no
        words yet get them at their next step."""
        keep = {}
        for gi, group in enumerate(self.param_groups):
            if gi in self._dev or group.get("capturable"):
                for p in group["params"]:
                    st = self.state.get(p)
                    if st and "exp_avg" in st:
                        keep[p] = (st["exp_avg"], st["exp_avg_sq"])
        super().load_state_dict(state_dict)
        for p, (m, v) in keep.items():
            st = self.state.get(p)
            if not st or "exp_avg" not in st:
                continue
            if st["exp_avg"] is not m:
                m.copy_(st["exp_avg"])
                st["exp_avg"] = m
            if st["exp_avg_sq"] is not v:
                v.copy_(st["exp_avg_sq"])
                st["exp_avg_sq"] = v
        for gi, group in enumerate(self.param_groups):
            if gi not in self._dev:
                continue
            steps = [int(self.state[p]["step"]) for p in group["params"] if p in self.state and "step" in self.state[p]]
            self._dev[gi][0].fill_(max(steps) if steps else 0)
            self._dev[gi][1].fill_(group["lr"])
            self._dev[gi][2] = group["lr"]
            self._dev[gi][3].zero_()

    @torch.no_grad()
    def step(self, closure=None, grads=None):
        """`grads` (optional dict parameter -> tensor): gradients to use instead of p.grad -- a data-parallel step's bf16
        views of its all-reduced 2-byte bucket (train.GraphedTrainStep(grad_dtype=torch.bfloat16))."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps, gs, ms, vs = [], [], [], []
            for p in group["params"]:
                pg = grads.get(p) if grads is not None else p.grad
                if pg is None:
                    continue
                if pg.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ps.append(p)
                gs.append(pg if pg.is_contiguous() else pg.contiguous())
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
            if not ps:
                continue
            kw = dict(lr=group["lr"], betas=group["betas"], eps=group["eps"], weight_decay=group["weight_decay"])
            if group["capturable"]:
                step_dev, lr_dev, _, ticket = self._group_dev(gi, group, ps[0].device)
                if not torch.cuda.is_current_stream_capturing():
                    self.sync_lr()
                bump = self._bump if gi == 0 and self._bump is not None else (None, 0)
                ops.adam_step(ps, gs, ms, vs, lr_device=lr_dev, step_device=step_dev, ticket=ticket,
                              bump_counter=bump[0], bump_by=bump[1], **kw)
            else:
                step = int(self.state[ps[0]]["step"]) + 1
                for p in ps:
                    self.state[p]["step"] = step
                ops.adam_step(ps, gs, ms, vs, step=step, **kw)
        return loss
