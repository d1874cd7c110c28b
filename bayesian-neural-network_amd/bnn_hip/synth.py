"""Synthetic inputs with the reference's distributions.

Pure numpy on the frozen legacy ``RandomState`` stream so that the golden-vector script,
the tests and ``bench.py`` regenerate bit-identical parameters / batches / epsilon from
seeds instead of shipping large fixtures.  Distributions: mu ~ U(mu_init), rho ~ U(rho_init)
(config.py:22-23, 52-53 of the reference), x ~ U(0,1) images (ToTensor range), labels
~ U{0..C-1}; regression x ~ U(0, 0.6), y ~ U(-1, 1).
"""
from __future__ import annotations

import numpy as np

SEED_PARAMS = 1234
SEED_DATA = 5678
SEED_EPS = 9999

LAYER_NAMES = ("l1", "l2", "l3")
PARAM_NAMES = ("weight_mu", "weight_rho", "bias_mu", "bias_rho")


def layer_dims(input_shape: int, hidden_units: int, classes: int):
    return [(input_shape, hidden_units), (hidden_units, hidden_units), (hidden_units, classes)]


def synth_state_dict(input_shape, hidden_units, classes, local_reparam, seed=SEED_PARAMS,
                     mu_init=(-0.2, 0.2), rho_init=(-5.0, -4.0)):
    """12-key fp32 state_dict (numpy).  BBB weights [out,in]; LR weights [in,out]."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, (fin, fout) in zip(LAYER_NAMES, layer_dims(input_shape, hidden_units, classes)):
        wshape = (fin, fout) if local_reparam else (fout, fin)
        sd[f"{name}.weight_mu"] = rs.uniform(mu_init[0], mu_init[1], wshape).astype(np.float32)
        sd[f"{name}.weight_rho"] = rs.uniform(rho_init[0], rho_init[1], wshape).astype(np.float32)
        sd[f"{name}.bias_mu"] = rs.uniform(mu_init[0], mu_init[1], (fout,)).astype(np.float32)
        sd[f"{name}.bias_rho"] = rs.uniform(rho_init[0], rho_init[1], (fout,)).astype(np.float32)
    return sd


def synth_batch(mode, batch, input_shape, classes, seed=SEED_DATA):
    rs = np.random.RandomState(seed)
    if mode == "classification":
        side = int(round(input_shape ** 0.5))
        if side * side == input_shape:
            x = rs.uniform(0.0, 1.0, (batch, 1, side, side)).astype(np.float32)
        else:
            x = rs.uniform(0.0, 1.0, (batch, input_shape)).astype(np.float32)
        y = rs.randint(0, classes, (batch,)).astype(np.int64)
    else:
        x = rs.uniform(0.0, 0.6, (batch, input_shape)).astype(np.float32)
        y = rs.uniform(-1.0, 1.0, (batch, classes)).astype(np.float32)
    return x, y


def synth_eps(shapes, sample, seed=SEED_EPS):
    """Parity-mode epsilon for MC sample ``sample``: one RandomState per sample so any
    subset of samples can be regenerated independently (multi-GPU shards)."""
    rs = np.random.RandomState(seed + 7919 * int(sample))
    return [rs.standard_normal(s).astype(np.float32) for s in shapes]
