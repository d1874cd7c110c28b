"""F3 / F4 host side: the reference's post-hoc analyses on device tensors.

`ECELoss` mirrors compute_ece.py:14-57 (same constructor, same return triple) and `compute_snr` /
`prune_weights` mirror weight_pruning.py:85-115 (same names and argument meaning), so the reference's scripts can
call them with the tensors they already hold; the per-element work runs in bnn_ece / bnn_snr_db / bnn_snr_prune.
Neither reference module can be imported in the build container (both need seaborn at import), so these two rows
have no reference-recorded vectors: the oracle restates them from the source text (parity UNPINNED).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class ECELoss(torch.nn.Module):
    """compute expected calibration error (compute_ece.py:14-57).  forward(probs [n, classes], labels [n]) with
    device tensors returns (ece float, bin_centers[have_data] ndarray, bin_acc ndarray) like the reference."""

    def __init__(self, bin_step=0.1, num_classes=10):
        super().__init__()
        self.bin_step = bin_step
        self.num_classes = num_classes

    def forward(self, probs, labels):
        bins = np.arange(0, 1.1, self.bin_step)                      # compute_ece.py:32, verbatim: the same float64 edges
        if not torch.is_tensor(probs):
            probs = torch.as_tensor(np.asarray(probs, np.float32))
            labels = torch.as_tensor(np.asarray(labels, np.int64))
            dev = torch.device("cuda", torch.cuda.current_device())
            probs, labels = probs.to(dev), labels.to(dev)
        out = ops.ece_bins(probs, labels, bins).double().cpu().numpy()
        stats = out[1:].reshape(-1, 3)
        counts, corrects = stats[:, 0], stats[:, 1]
        bin_centers = bins[1:] - self.bin_step / 2                   # :36
        have_data = counts > 0                                       # :50
        bin_acc = corrects[have_data] / counts[have_data]            # :51
        return float(out[0]), bin_centers[have_data], bin_acc

    def bins(self, probs, labels):
        """(counts, corrects, mean confidence) per bin -- what the reliability diagram plots."""
        bins = np.arange(0, 1.1, self.bin_step)
        st = ops.ece_bins(probs, labels, bins).double().cpu().numpy()[1:].reshape(-1, 3)
        with np.errstate(invalid="ignore", divide="ignore"):
            return st[:, 0], st[:, 1], st[:, 2] / st[:, 0]


def _bayesian_layers(model):
    return [l for l in model.children() if hasattr(l, "weight_mu") and hasattr(l, "weight_rho")]


def compute_snr(model_or_mu, sigma=None):
    """weight_pruning.py:85-87.  compute_snr(mu, sigma) with numpy / python inputs keeps the reference's numpy
    arithmetic; compute_snr(model) returns the SNR (dB, fp32 device tensor) of every stochastic parameter of the
    model in named_parameters() order of the (mu, rho) pairs -- the vector the reference builds through
    collect_weights (weight_pruning.py:15-40) -- in one device pass per tensor."""
    if sigma is not None:
        return 10 * np.log10(abs(model_or_mu) / sigma)
    parts = []
    for l in _bayesian_layers(model_or_mu):
        parts.append(ops.snr_db(l.weight_mu.detach(), l.weight_rho.detach()).flatten())
        parts.append(ops.snr_db(l.bias_mu.detach(), l.bias_rho.detach()).flatten())
    return torch.cat(parts)


def snr_threshold(snrs, drop_percentage: float) -> float:
    """np.percentile(snrs, 100 * drop_percentage) (weight_pruning.py:92: linear interpolation between order
    statistics), from a device sort when `snrs` is a device tensor."""
    if not torch.is_tensor(snrs):
        return float(np.percentile(snrs, 100 * drop_percentage))
    v, _ = torch.sort(snrs.flatten())
    pos = (v.numel() - 1) * float(drop_percentage)
    lo = int(np.floor(pos))
    hi = min(lo + 1, v.numel() - 1)
    a, b = float(v[lo]), float(v[hi])
    if a == b:                                                      # also covers (-inf, -inf): no inf - inf
        return a
    return a + (b - a) * (pos - lo)


def prune_weights(model, snrs=None, drop_percentage=0.5):
    """Remove weights with the lowest SNR (weight_pruning.py:89-115): in place, mu *= mask and rho *= mask with
    mask = snr > threshold, for the weights and biases of every Bayesian layer.  `snrs` None: computed here."""
    if snrs is None:
        snrs = compute_snr(model)
    thr = snr_threshold(snrs, drop_percentage)
    with torch.no_grad():
        for l in _bayesian_layers(model):
            for mu, rho in ((l.weight_mu, l.weight_rho), (l.bias_mu, l.bias_rho)):
                if not (mu.data.is_contiguous() and rho.data.is_contiguous()):
                    mu.data, rho.data = mu.data.contiguous(), rho.data.contiguous()
                ops.snr_prune_(mu.data, rho.data, thr)
    return thr
