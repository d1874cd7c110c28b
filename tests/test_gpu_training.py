"""Drop-in behaviour under the reference's own call pattern (regression/reg_task.py:60-83,
classification/class_task.py:67-103): train_step loop with the beta schedule, Adam, backward
through the HIP kernels, then MC-averaged prediction."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bnn_hip
from bnn_hip import synth


def _beta(M, idx):
    return 2 ** (M - (idx + 1)) / (2 ** M - 1)


@pytest.mark.parametrize("local_reparam", [False, True])
def test_regression_training_reduces_loss_and_fits(local_reparam):
    import networks
    from config import DEVICE
    assert DEVICE.type == "cuda"
    bnn_hip.set_math("bf16")
    bnn_hip.manual_seed(123)
    torch.manual_seed(0)
    mp = dict(input_shape=1, classes=1, batch_size=128, hidden_units=50, mode="regression", mu_init=[-0.2, 0.2],
              rho_init=[-5, -4], prior_init=[1], mixture_prior=False, local_reparam=local_reparam)
    net = networks.BayesianNetwork(mp).to(DEVICE)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    rs = np.random.RandomState(0)                                        # data_utils.py:64-70 style toy curve
    X = rs.uniform(0.0, 0.5, (1024, 1)).astype(np.float32)
    Y = (X + 0.3 * np.sin(2 * np.pi * X) + 0.3 * np.sin(4 * np.pi * X)).astype(np.float32)
    Xd, Yd = torch.from_numpy(X).to(DEVICE), torch.from_numpy(Y).to(DEVICE)
    M = 8
    first = last = None
    for epoch in range(60):
        net.train()
        for idx in range(M):
            xb, yb = Xd[idx * 128:(idx + 1) * 128], Yd[idx * 128:(idx + 1) * 128]
            net.zero_grad()
            info = (net.sample_elbo_lr if local_reparam else net.sample_elbo)(xb, yb, _beta(M, idx), 5, 0.1)
            assert len(info) == (3 if local_reparam else 4) and info[0].shape == (1,)
            info[0].backward()
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
            opt.step()
        nll = float(info[-1].detach())
        first = nll if first is None else first
        last = nll
    assert last < 0.2 * first, (first, last)                              # the data term collapsed
    net.eval()
    with torch.no_grad():                                                 # reg_task.py:76-83: MC-averaged prediction
        preds = torch.stack([net(Xd[:128], sample=True) for _ in range(10)]).mean(0)
    rmse = float(((preds - Yd[:128]) ** 2).mean().sqrt())
    assert rmse < 0.08, rmse
    sd = net.state_dict()                                                 # main.py:55-57 / :63 round trip
    net2 = networks.BayesianNetwork(mp).to(DEVICE)
    net2.load_state_dict(sd)
    net2.eval()
    with torch.no_grad():
        a = net(Xd[:8])
        b = net2(Xd[:8])
    assert torch.equal(a, b)                                              # eval & sample=False: deterministic mean weights
