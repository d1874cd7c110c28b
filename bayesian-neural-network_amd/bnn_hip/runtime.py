"""Process-wide knobs of the HIP path.  The reference's constructors take no such
arguments (networks.py:50, :92, :142), so build-side settings live here and in env vars:

  BNN_HIP_MATH   bf16 (default; bf16 MFMA operands, fp32 accumulate, fp32 statistics)
                 f32  (exact fp32 MFMA; the parity mode)
                 bf16x3 (split-bf16 operands, three bf16 MFMAs per product: the reference's fp32 F.linear to ~1e-5 of
                         the output scale at 3/16 of the exact mode's matrix-core time; the BBB forward paths and the
                         stacked-minibatch path of a local-reparameterisation network (its mean product; the variance
                         product stays on bf16 operands) -- every other launch of a job in this mode, the backward kernels
                         included, runs exact fp32)
  BNN_HIP_SEED   64-bit Philox key (default 2026)
  BNN_HIP_EPS    device (default; on-chip Philox)  |  host (draw eps with torch's CPU
                 generator in the reference's order, networks.py:42, then copy H2D)
"""
from __future__ import annotations

import os
import threading

from . import _lib as L

_lock = threading.Lock()


class _State:
    def __init__(self):
        m = os.environ.get("BNN_HIP_MATH", "bf16").lower()
        self.math = L.MATH_F32 if m in ("f32", "fp32", "float32") else L.MATH_BF16X3 if m in ("bf16x3", "x3") else L.MATH_BF16
        self.seed = int(os.environ.get("BNN_HIP_SEED", "2026"))
        self.host_eps = os.environ.get("BNN_HIP_EPS", "device").lower() == "host"
        self.counter = 0            # next unused GLOBAL MC sample index
        self.shard_samples = False  # split MC samples over torch.distributed ranks
        self.form = L.FORM_AUTO     # kernel-form preference handed to every layer launch (bnn_form; the tests compare
                                    # the forms with each other through it)
        self.device_counter = None  # device int32[1] added to every layer's sample index at run time
                                    # (set while a training step is captured as a hipGraph: train.py)


state = _State()


def set_math(mode: str):
    """'bf16', 'f32' or 'bf16x3'."""
    m = mode.lower()
    if m in ("bf16", "bfloat16"):
        state.math = L.MATH_BF16
    elif m in ("f32", "fp32", "float32"):
        state.math = L.MATH_F32
    elif m in ("bf16x3", "x3"):
        state.math = L.MATH_BF16X3
    else:
        raise ValueError(f"unknown math mode {mode!r}")


def get_math() -> str:
    return {L.MATH_BF16: "bf16", L.MATH_F32: "f32", L.MATH_BF16X3: "bf16x3"}[state.math]


def manual_seed(seed: int, counter: int = 0):
    """Re-key the on-chip epsilon stream and rewind the global MC sample counter."""
    with _lock:
        state.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        state.counter = int(counter)


def take_samples(n: int) -> int:
    """Reserve n consecutive global MC sample indices; returns the first.  Every rank of a
    sharded job calls this with the same GLOBAL n so the counters stay in step."""
    with _lock:
        first = state.counter
        state.counter = (state.counter + int(n)) & 0xFFFFFFFF
        return first


def set_host_eps(flag: bool):
    state.host_eps = bool(flag)


def shard_samples(flag: bool = True):
    """Split the MC samples of sample_elbo*/predict over the ranks of the default
    torch.distributed group (RCCL on GPUs); the only collective is a sum all-reduce of the
    ELBO scalars."""
    state.shard_samples = bool(flag)
