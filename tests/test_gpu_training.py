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


def test_fused_adam_matches_torch_adam():
    """F2: bnn_adam_step against torch.optim.Adam (class_task.py:60) over several steps, odd
    tensor sizes, weight decay, a learning-rate change (StepLR, class_task.py:61) and more
    than 16 tensors."""
    from bnn_hip.optim import FusedAdam
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(3)
    shapes = [(1200, 784), (1200,), (7,), (33, 5), (1,), (4096,), (1025,)] + [(3, 3)] * 12
    for wd, capturable in ((0.0, False), (0.01, False), (0.0, True)):
        base = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)).to(dev) for s in shapes]
        pa = [torch.nn.Parameter(b.clone()) for b in base]
        pb = [torch.nn.Parameter(b.clone()) for b in base]
        oa = torch.optim.Adam(pa, lr=1e-3, weight_decay=wd)
        ob = FusedAdam(pb, lr=1e-3, weight_decay=wd, capturable=capturable)
        sa = torch.optim.lr_scheduler.StepLR(oa, step_size=3, gamma=0.5)
        sb = torch.optim.lr_scheduler.StepLR(ob, step_size=3, gamma=0.5)
        for it in range(7):
            for a, b in zip(pa, pb):
                g = torch.from_numpy(rs.standard_normal(tuple(a.shape)).astype(np.float32)).to(dev) * (10.0 ** (it % 3 - 1))
                a.grad, b.grad = g.clone(), g.clone()
            oa.step(); ob.step(); sa.step(); sb.step()
        if capturable:
            assert ob.device_step() == 7
        for a, b in zip(pa, pb):
            err = float((a.detach() - b.detach()).abs().max())
            assert err <= 2e-6 * (float(a.detach().abs().max()) + 1.0), err
        for a, b in zip(pa, pb):
            close_m = float((oa.state[a]["exp_avg"] - ob.state[b]["exp_avg"]).abs().max())
            close_v = float((oa.state[a]["exp_avg_sq"] - ob.state[b]["exp_avg_sq"]).abs().max())
            assert close_m <= 1e-6 * (float(oa.state[a]["exp_avg"].abs().max()) + 1e-6) + 1e-9
            assert close_v <= 1e-6 * (float(oa.state[a]["exp_avg_sq"].abs().max()) + 1e-6) + 1e-12


def test_fused_adam_reads_bf16_gradients():
    """bnn_adam_args.grad_dtype = bf16 (the all-reduced 2-byte bucket of a data-parallel step): FusedAdam.step(grads=...)
    over bf16 gradient tensors against torch.optim.Adam fed the same values widened to fp32; odd sizes (scalar tail
    path), 20 tensors (two launches)."""
    from bnn_hip.optim import FusedAdam
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(4)
    shapes = [(1200, 784), (1200,), (7,), (33, 5), (1,), (4099,)] + [(3, 3)] * 14
    base = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)).to(dev) for s in shapes]
    pa = [torch.nn.Parameter(b.clone()) for b in base]
    pb = [torch.nn.Parameter(b.clone()) for b in base]
    oa = torch.optim.Adam(pa, lr=1e-3)
    ob = FusedAdam(pb, lr=1e-3, capturable=True)
    for it in range(4):
        g16 = {}
        for a, b in zip(pa, pb):
            g = (torch.from_numpy(rs.standard_normal(tuple(a.shape)).astype(np.float32)).to(dev) * (10.0 ** (it % 3 - 1))).to(torch.bfloat16)
            a.grad, g16[b] = g.float(), g
        oa.step()
        ob.step(grads=g16)
    assert ob.device_step() == 4
    for a, b in zip(pa, pb):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * (float(a.detach().abs().max()) + 1.0)


@pytest.mark.parametrize("mode", ["classification", "regression"])
def test_nll_backward_kernel_matches_autograd(mode):
    from bnn_hip import ops
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(4)
    S, B, C = 3, 37, (10 if mode == "classification" else 4)
    logits = torch.from_numpy(rs.standard_normal((S, B, C)).astype(np.float32) * 3).to(dev).requires_grad_(True)
    g = torch.from_numpy(rs.uniform(0.2, 1.0, S).astype(np.float32)).to(dev)
    if mode == "classification":
        tgt = torch.from_numpy(rs.randint(0, C, B)).to(dev)
        nll = torch.stack([torch.nn.functional.cross_entropy(logits[s], tgt, reduction="sum") for s in range(S)])
    else:
        tgt = torch.from_numpy(rs.standard_normal((B, C)).astype(np.float32)).to(dev)
        sig = 0.7
        nll = torch.stack([-torch.distributions.Normal(logits[s], sig).log_prob(tgt).sum() for s in range(S)])
    (nll * g).sum().backward()
    got = ops.nll_bwd(logits.detach(), tgt, g, mode, 0.7)
    assert float((got - logits.grad).abs().max()) <= 2e-6 * float(logits.grad.abs().max())


@pytest.mark.parametrize("mode", ["classification", "regression"])
@pytest.mark.parametrize("local_reparam", [False, True])
def test_fused_loss_and_nll_backward_equal_the_two_launches(mode, local_reparam):
    """bnn_elbo_loss_nll_bwd = bnn_elbo_loss followed by bnn_nll_bwd, bit for bit (same arithmetic, one launch)."""
    from bnn_hip import ops
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(8)
    for S, B in ((1, 128), (3, 37), (8, 300)):
        C = 10 if mode == "classification" else 4
        logits = torch.from_numpy(rs.standard_normal((S, B, C)).astype(np.float32) * 3).to(dev)
        tgt = (torch.from_numpy(rs.randint(0, C, B)) if mode == "classification"
               else torch.from_numpy(rs.standard_normal((B, C)).astype(np.float32))).to(dev)
        a, b, nll = (torch.from_numpy(rs.standard_normal(S).astype(np.float32) * 100).to(dev) for _ in range(3))
        beta = torch.tensor(0.37, device=dev)
        out4, g_a, g_b, g_nll, g_kl3 = ops.elbo_loss(a, None if local_reparam else b, nll, beta, 2 * S, local_reparam,
                                                     grad_scale=0.5)
        g_ref = ops.nll_bwd(logits, tgt, g_nll, mode, 0.7)
        o4, ga, gb, gk, g = ops.elbo_loss_nll_bwd(a, None if local_reparam else b, nll, beta, 2 * S, local_reparam, logits,
                                                  tgt, mode, 0.7, grad_scale=0.5)
        assert torch.equal(o4[:2], out4[:2]) and torch.equal(o4[3], out4[3]) and torch.equal(gk, g_kl3)
        assert torch.equal(ga, g_a) and torch.equal(gb, g_b) and torch.equal(g, g_ref)


def test_stage_inputs_copies_and_sets_the_word():
    from bnn_hip import ops
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(9)
    for nx, ny in ((128 * 784, 128), (7, 3), (1, 0)):
        x = torch.from_numpy(rs.standard_normal(nx).astype(np.float32)).to(dev)
        y = torch.from_numpy(rs.randint(0, 10, ny)).to(dev) if ny else None
        dx, dy, w = torch.zeros_like(x), (torch.zeros_like(y) if ny else None), torch.zeros((), device=dev)
        ops.stage_inputs(x, dx, y, dy, w, 0.625)
        assert torch.equal(dx, x) and (dy is None or torch.equal(dy, y)) and float(w) == 0.625
    xo = torch.arange(41, dtype=torch.float32, device=dev)[1:]           # 4-byte aligned only: byte path
    d = torch.zeros(40, device=dev)
    ops.stage_inputs(xo, d)
    assert torch.equal(d, xo)
    with pytest.raises(ops.BnnHipError):
        ops.stage_inputs(x, torch.zeros(3, device=dev))


def test_adam_launch_advances_step_and_sample_counter():
    """Capturable FusedAdam: the device step and the attached counter advance inside the update launch (last
    block's hand-off), once per step, also when the tensors need more than one launch."""
    from bnn_hip.optim import FusedAdam
    dev = torch.device("cuda:0")
    ps = [torch.nn.Parameter(torch.randn(5000 if i == 0 else 9, device=dev)) for i in range(20)]
    opt = FusedAdam(ps, lr=1e-3, capturable=True)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    opt.bump_after_step(counter, 6)
    for it in range(1, 5):
        for p_ in ps:
            p_.grad = torch.randn_like(p_)
        opt.step()
        assert opt.device_step() == it and int(counter) == 6 * it
        assert int(opt._dev[0][3].abs().sum()) == 0                      # ticket words back at zero


@pytest.mark.parametrize("variant", ["bbb", "lr"])
def test_input_gradient_applies_the_relu_mask_of_the_layer_below(variant):
    """gx_relu_mask: g_x * (x > 0) from the input-gradient kernel equals the separate mask pass the layer below
    would run on it (its output IS this layer's x)."""
    from bnn_hip import ops, _lib as L
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(10)
    for S, B, K, N in ((2, 128, 1200, 1200), (3, 20, 72, 38), (1, 7, 33, 10)):
        mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
        x = torch.relu(mk(S, B, K, lo=-1, hi=1))
        gy = mk(S, B, N, lo=-1, hi=1)
        b_mu, b_rho = mk(N), mk(N, lo=-5, hi=-2)
        kw = dict(n_samples=S, relu=False, eps_mode=L.EPS_PHILOX, seed=3, layer_id=1, sample_offset=5)
        if variant == "bbb":
            w_mu, w_rho = mk(N, K), mk(N, K, lo=-5, hi=-2)
            run = lambda m: ops.bbb_linear_bwd(x, gy, None, w_mu, w_rho, b_mu, b_rho, prior=ops.PriorSpec(False, 1.0),
                                               math_mode=L.MATH_F32, gx_relu_mask=m, **kw)
        else:
            w_mu, w_rho = mk(K, N), mk(K, N, lo=-5, hi=-2)
            v = mk(S, B, N, lo=0.1, hi=1.0)
            run = lambda m: ops.lr_linear_bwd(x, gy, None, v, w_mu, w_rho, b_mu, b_rho, sigma_p=1.0, gx_relu_mask=m, **kw)
        plain, masked = run(False), run(True)
        for a, b in zip(plain[:4], masked[:4]):
            assert torch.equal(a, b)
        assert torch.equal(masked[4], plain[4] * (x > 0))
        assert float(masked[4].abs().max()) > 0


def test_input_gradient_over_transposed_sampled_weights():
    """bnn_bbb_fwd_args.w_sampled_t_out / bnn_bbb_bwd_args.w_sampled_t: the forward's matmul-only launch leaves the
    sampled weights transposed, the backward computes g_x as that same launch over them (bf16 gy copy in, bf16 g_x copy
    out) -- against the transposed-gather form over w_sampled: same bf16 products, fp32 sums in another order."""
    from bnn_hip import ops, _lib as L
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(23)
    for S, B, K, N in ((2, 128, 1200, 1200), (3, 20, 72, 40), (1, 7, 40, 24), (2, 100, 784, 400)):
        mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
        x = torch.relu(mk(S, B, K, lo=-1, hi=1))
        w_mu, w_rho, b_mu, b_rho = mk(N, K), mk(N, K, lo=-5, hi=-2), mk(N), mk(N, lo=-5, hi=-2)
        w_s = (w_mu + 0.01 * mk(S, N, K)).to(torch.bfloat16).contiguous()
        b_s = mk(S, N)
        wt = torch.zeros((S, K, N), dtype=torch.bfloat16, device=dev)
        y = ops.bbb_sampled_matmul(x.to(torch.bfloat16), w_s, b_s, n_samples=S, relu=True, y_dtype=torch.float32, wt_out=wt)
        assert torch.equal(wt, w_s.transpose(1, 2).contiguous()), (S, B, K, N)
        want = torch.relu(torch.einsum("sbk,snk->sbn", x.to(torch.bfloat16).float(), w_s.float()) + b_s[:, None, :])
        assert float((y - want).abs().max()) <= 1e-4 * float(want.abs().max())
        gy = mk(S, B, N, lo=-1, hi=1)
        kw = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16, relu=False, eps_mode=L.EPS_PHILOX, seed=11,
                  layer_id=1, sample_offset=40, gx_relu_mask=True)
        ref = ops.bbb_linear_bwd(x, gy, None, w_mu, w_rho, b_mu, b_rho, w_sampled=w_s, **kw)
        for gy16 in (None, gy.to(torch.bfloat16)):
            got = ops.bbb_linear_bwd(x, gy, None, w_mu, w_rho, b_mu, b_rho, w_sampled=w_s, w_sampled_t=wt, gy16=gy16,
                                     want_gx16=True, **kw)
            for a, b in zip(got[:4], ref[:4]):
                assert torch.equal(a, b)
            assert float((got[4] - ref[4]).double().norm()) <= 1e-5 * float(ref[4].double().norm()), (S, B, K, N)
            assert torch.equal(got[5], got[4].to(torch.bfloat16))
            assert torch.equal(got[4] == 0, ref[4] == 0)


def test_lr_backward_from_the_saved_factor_equals_the_prepared_form():
    """bnn_lr_fwd_args.hfac_out / bnn_lr_bwd_args.hfac: a hidden layer of a training step saves eps_act / (2 sqrt(v)) in
    its forward and its backward forms h = gz * hfac as it loads the operands -- no preparation launch; the reference is
    the same backward from the saved v (eps regenerated by the preparation launch)."""
    from bnn_hip import ops, _lib as L
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(14)
    for S, B, K, N in ((2, 128, 1200, 1200), (3, 20, 72, 40), (1, 7, 33, 24), (2, 9, 31, 21)):
        mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
        x = torch.relu(mk(S, B, K, lo=-1, hi=1))
        p = (mk(K, N), mk(K, N, lo=-5, hi=-2), mk(N), mk(N, lo=-5, hi=-2))
        fw = ops.lr_linear_fwd(x, *p, n_samples=S, sigma_p=1.0, math_mode=L.MATH_F32, relu=True, y_dtype=torch.float32,
                               eps_mode=L.EPS_PHILOX, seed=3, layer_id=1, sample_offset=5, want_kl=True, want_v=True, want_hfac=True)
        eps = ops.philox_normal(3, 4 * 1 + 2, 5, S, B, N, dev)
        want = eps / (2 * torch.sqrt(fw["v"]))
        assert float((fw["hfac"] - want).abs().max()) <= 1e-5 * float(want.abs().max())
        gy = mk(S, B, N, lo=-1, hi=1)
        kw = dict(n_samples=S, relu=False, eps_mode=L.EPS_PHILOX, seed=3, layer_id=1, sample_offset=5, sigma_p=1.0, gx_relu_mask=True,
                  g_kl=mk(3, lo=0.1, hi=0.5))
        for mm in (L.MATH_F32, L.MATH_BF16):
            ref = ops.lr_linear_bwd(x, gy, None, fw["v"], *p, math_mode=mm, **kw)
            got = ops.lr_linear_bwd(x, gy, None, None, *p, math_mode=mm, hfac=fw["hfac"], **kw)
            for i, (a, b) in enumerate(zip(got, ref)):
                tol = 2e-3 if (mm == L.MATH_BF16 and i == 4 and N % 8 == 0) else 2e-5     # (a bf16 ulp of h here and there)
                assert float((a - b).double().norm()) <= tol * float(b.double().norm()), (S, B, K, N, mm, i)


def test_lr_input_gradient_in_bf16_math():
    """bnn_lr_bwd_args.math = BNN_MATH_BF16: g_x = gz M^T + 2 x (h (sigma^2)^T) with the four operands rounded to bf16
    (fp32 accumulation) -- checked against that arithmetic in torch and against the exact-fp32 form; the weight
    gradients do not depend on the mode."""
    from bnn_hip import ops, _lib as L
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(12)
    for S, B, K, N in ((2, 128, 1200, 1200), (3, 20, 72, 40), (1, 7, 33, 24)):     # (<= 16 outputs: the fp32 one-launch form)
        mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
        x = torch.relu(mk(S, B, K, lo=-1, hi=1))
        gy, v = mk(S, B, N, lo=-1, hi=1), mk(S, B, N, lo=0.1, hi=1.0)
        w_mu, w_rho, b_mu, b_rho = mk(K, N), mk(K, N, lo=-5, hi=-2), mk(N), mk(N, lo=-5, hi=-2)
        kw = dict(n_samples=S, relu=False, eps_mode=L.EPS_PHILOX, seed=3, layer_id=1, sample_offset=5, sigma_p=1.0, gx_relu_mask=True)
        f32 = ops.lr_linear_bwd(x, gy, None, v, w_mu, w_rho, b_mu, b_rho, math_mode=L.MATH_F32, **kw)
        b16 = ops.lr_linear_bwd(x, gy, None, v, w_mu, w_rho, b_mu, b_rho, math_mode=L.MATH_BF16, **kw)
        for a, b in zip(f32[:4], b16[:4]):
            assert torch.equal(a, b)
        eps = ops.philox_normal(3, 4 * 1 + 2, 5, S, B, N, dev)
        h = gy * eps / (2 * torch.sqrt(v))
        r = lambda t: t.to(torch.bfloat16).float()
        sig2 = torch.log1p(torch.exp(w_rho)) ** 2
        want = (r(gy) @ r(w_mu).T + 2 * x * (r(h) @ r(sig2).T)) * (x > 0)
        nrm = lambda t: float(t.double().norm())
        assert nrm(b16[4] - want) <= 2e-3 * nrm(want), (S, B, K, N)      # (a bf16 ulp of sigma^2 here and there: softplus forms differ)
        assert nrm(b16[4] - f32[4]) <= 2e-2 * nrm(f32[4]), (S, B, K, N)
        assert torch.equal(b16[4] == 0, f32[4] == 0) or nrm(((b16[4] == 0) != (f32[4] == 0)).float()) ** 2 < 1e-3 * x.numel()


@pytest.mark.parametrize("mode", ["classification", "regression"])
def test_lr_final_launch_with_loss_tail_and_saved_variance(mode):
    """bnn_lr_final_fwd as the training step calls it: the row-split launch (K3r) also saves the variance for the backward
    and carries the loss tail (bnn_finalize_args.loss); reference = bnn_lr_linear_fwd (K3a, want_v) + bnn_elbo_finalize +
    bnn_elbo_loss_nll_bwd on the same Philox elements.  The two forms sum the k range in different orders."""
    from bnn_hip import ops, _lib as L
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(41)
    shapes = ((1, 128, 1200, 10), (2, 128, 1200, 10), (3, 37, 64, 3), (2, 16, 64, 24)) if mode == "classification" else \
        ((1, 128, 56, 1), (3, 50, 400, 1))
    for S, B, K, N in shapes:
        mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
        p = (mk(K, N), mk(K, N, lo=-5, hi=-4), mk(N), mk(N, lo=-5, hi=-4))
        x = torch.relu(mk(S, B, K, lo=-1, hi=1)).to(torch.bfloat16).contiguous()
        y = (torch.from_numpy(rs.randint(0, N, B)).to(dev) if mode == "classification" else mk(B, N, lo=-1, hi=1))
        beta = torch.full((), 0.37, dtype=torch.float32, device=dev)
        kw = dict(n_samples=S, sigma_p=1.0, math_mode=L.MATH_BF16, relu=False, y_dtype=torch.float32, eps_mode=L.EPS_PHILOX,
                  seed=7, layer_id=2, sample_offset=11, want_kl=True, want_v=True)
        fin_kw = dict(layer_in=[K], layer_out=[N], local_reparam=True, prior=ops.PriorSpec(False, 1.0), n_samples=S, target=y,
                      mode=mode, nll_sigma=0.3)
        ws = ops.lr_workspace(N, dev)
        one, fin = ops.lr_final_fwd((x,) + p, dict(workspace=ws, **kw),
                                    dict(workspaces=[ws], scratch=ops.final_scratch(S, dev),
                                         ticket=torch.zeros(1, dtype=torch.int32, device=dev) if S > 1 else None,
                                         loss=dict(beta=beta, total_samples=S, grad_scale=0.5), **fin_kw))
        ws2 = ops.lr_workspace(N, dev)
        ref = ops.lr_linear_fwd(x, *p, workspace=ws2, **kw)
        fin2 = ops.elbo_finalize(workspaces=[ws2], logits=ref["y"], **fin_kw)
        loss2 = ops.elbo_loss_nll_bwd(fin2["kl"], None, fin2["nll"], beta, S, True, ref["y"], y, mode, 0.3, grad_scale=0.5)
        torch.cuda.synchronize()
        close = lambda a, b, tol: float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-12
        assert close(one["y"], ref["y"], 1e-4) and close(one["v"], ref["v"], 1e-4), (mode, S, B, K, N)
        assert close(fin["kl"], fin2["kl"], 1e-5) and close(fin["nll"], fin2["nll"], 1e-4), (mode, S, B, K, N)
        for i, (a, b) in enumerate(zip(fin["loss"], loss2)):
            assert torch.isfinite(b).all()
            assert close(a, b, 2e-4), (mode, S, B, K, N, i)
        if N <= 16:
            # the same launch over PREPARED operands (bnn_lr_prepare here; in an evaluation the rider of the previous layer's
            # launch): the row blocks park nothing, the statistics block takes the KL sums that came with the fragments --
            # the same bf16 operands in the same order, so the logits and the variance are the same bits
            wfrag, wsp = ops.lr_prepare(*p)
            pre, fin3 = ops.lr_final_fwd((x,) + p, dict(workspace=wsp, w_frag=wfrag, **kw),
                                         dict(workspaces=[wsp], scratch=ops.final_scratch(S, dev),
                                              ticket=torch.zeros(1, dtype=torch.int32, device=dev) if S > 1 else None,
                                              loss=dict(beta=beta, total_samples=S, grad_scale=0.5), **fin_kw))
            torch.cuda.synchronize()
            assert torch.equal(pre["y"], one["y"]) and torch.equal(pre["v"], one["v"]), (mode, S, B, K, N)
            assert close(fin3["kl"], fin["kl"], 1e-6) and torch.equal(fin3["nll"], fin["nll"]), (mode, S, B, K, N)
            for i, (a, b) in enumerate(zip(fin3["loss"], fin["loss"])):
                assert close(a, b, 1e-5), (mode, S, B, K, N, i)


@pytest.mark.parametrize("mode", ["classification", "regression"])
def test_loss_tail_on_the_final_launch_equals_the_separate_launch(mode):
    """bnn_finalize_args.loss: the row-split final launch (bnn_bbb_final_fwd over sampled weights) also differentiates
    the rows' NLL and assembles the loss and the seeds; bnn_elbo_loss_nll_bwd on the same per-sample scalars and logits
    is the reference.  fp32 activations (the training step's) and bf16 ones; a wide layer takes the follow-up launch."""
    from bnn_hip import ops, _lib as L
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(33)
    shapes = ((1, 128, 1200, 10), (2, 128, 1200, 10), (5, 37, 64, 3), (2, 16, 64, 24)) if mode == "classification" else \
        ((1, 128, 56, 1), (3, 50, 400, 1))
    for S, B, K, N in shapes:
        for xdt in (torch.float32, torch.bfloat16):
            mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
            prior = ops.PriorSpec(False, 1.0)
            w_mu, w_rho, b_mu, b_rho = mk(N, K), mk(N, K, lo=-5, hi=-4), mk(N), mk(N, lo=-5, hi=-4)
            wsamp = torch.empty((S, N, K), dtype=torch.bfloat16, device=dev)
            bsamp = torch.empty((S, N), dtype=torch.float32, device=dev)
            wstat = ops.sample_workspace(S, K, N, dev)
            ops.bbb_sample_weights([dict(w_mu=w_mu, w_rho=w_rho, b_mu=b_mu, b_rho=b_rho, prior=prior, layer_id=2, workspace=wstat,
                                         w_out=wsamp, b_out=bsamp)], n_samples=S, seed=5, sample_offset=9, sample_counter=None)
            x = torch.relu(mk(S, B, K, lo=-1, hi=1)).to(xdt).contiguous()
            y = (torch.from_numpy(rs.randint(0, N, B)).to(dev) if mode == "classification" else mk(B, N, lo=-1, hi=1))
            beta = torch.full((), 0.37, dtype=torch.float32, device=dev)
            fin_kw = dict(layer_in=[K], layer_out=[N], local_reparam=False, prior=prior, n_samples=S, target=y, mode=mode,
                          nll_sigma=0.3, ticket=torch.zeros(1, dtype=torch.int32, device=dev) if S > 1 else None)
            res, fin = ops.bbb_final_fwd((x, None, None, None, None),
                                         dict(n_samples=S, prior=prior, math_mode=L.MATH_BF16, relu=False, y_dtype=torch.float32,
                                              eps_mode=L.EPS_ZERO, want_stats=False, w_sampled=wsamp, b_sampled=bsamp),
                                         dict(workspaces=[wstat], scratch=ops.final_scratch(S, dev),
                                              loss=dict(beta=beta, total_samples=S, grad_scale=0.5), **fin_kw))
            ref = ops.elbo_loss_nll_bwd(fin["log_prior"], fin["log_q"], fin["nll"], beta, S, False, res["y"], y, mode, 0.3,
                                        grad_scale=0.5)
            got = fin["loss"]
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(got, ref)):
                assert torch.isfinite(b).all()
                assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-12, (mode, S, B, K, N, xdt, i)
            # the logits themselves: the matmul over the sampled weights, bf16-rounded activations
            want = torch.einsum("sbk,snk->sbn", x.to(torch.bfloat16).float(), wsamp.float()) + bsamp[:, None, :]
            assert float((res["y"] - want).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("relu", [False, True])
def test_narrow_output_layer_backward_equals_the_general_kernels(relu):
    """bnn_bbb_linear_bwd over the step's sampled weights: an output layer of <= 16 features takes ONE launch
    (bbb_out_layer_bwd_kernel, plain fp32 FMAs); the same call without w_sampled runs the general weight-gradient
    kernel and the transposed-generator input gradient.  Same Philox elements, same bf16 rounding of gz and w in the
    input gradient: they agree to summation order."""
    from bnn_hip import ops, _lib as L
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(21)
    for S, B, K, N, prior in ((2, 128, 1200, 10, ops.PriorSpec(False, 1.0)), (3, 20, 72, 1, ops.PriorSpec(True, 1.0, 0.5, 1.0, 0.0025)),
                              (1, 7, 40, 16, ops.PriorSpec(False, 0.7)), (5, 200, 56, 3, ops.PriorSpec(False, 1.0))):
        mk = lambda *sh, lo=-0.3, hi=0.3: torch.from_numpy(rs.uniform(lo, hi, sh).astype(np.float32)).to(dev)
        x = torch.relu(mk(S, B, K, lo=-1, hi=1))
        gy = mk(S, B, N, lo=-1, hi=1)
        y = mk(S, B, N, lo=-1, hi=1) if relu else None
        w_mu, w_rho, b_mu, b_rho = mk(N, K), mk(N, K, lo=-5, hi=-2), mk(N), mk(N, lo=-5, hi=-2)
        glp, glq = mk(S, lo=-0.5, hi=-0.1), mk(S, lo=0.1, hi=0.5)
        seed, layer_id, first = 11, 2, 40
        eps = ops.philox_normal(seed, 4 * layer_id, first, S, N, K, dev)
        w_s = (w_mu + torch.log1p(torch.exp(w_rho)) * eps).to(torch.bfloat16).contiguous()
        kw = dict(n_samples=S, prior=prior, math_mode=L.MATH_BF16, relu=relu, eps_mode=L.EPS_PHILOX, seed=seed, layer_id=layer_id,
                  sample_offset=first, g_log_prior=glp, g_log_q=glq, gx_relu_mask=True)
        one = ops.bbb_linear_bwd(x, gy, y, w_mu, w_rho, b_mu, b_rho, w_sampled=w_s, **kw)
        ref = ops.bbb_linear_bwd(x, gy, y, w_mu, w_rho, b_mu, b_rho, **kw)
        for a, b in zip(one[:4], ref[:4]):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (S, B, K, N)
        # the generator's bf16(w) and the bf16 of the fp32 w formed here may differ by one ulp in a few elements
        assert float((one[4] - ref[4]).norm()) <= 2e-3 * float(ref[4].norm()), (S, B, K, N)
        assert float(ref[4].abs().max()) > 0
        assert torch.equal(one[4] == 0, ref[4] == 0) or float(((one[4] == 0) != (ref[4] == 0)).float().mean()) < 1e-3


@pytest.mark.parametrize("math_mode", ["bf16", "f32"])
@pytest.mark.parametrize("local_reparam", [False, True])
def test_graphed_train_step_on_the_regression_config(local_reparam, math_mode):
    """The captured step on the regression network (1-50-1, reg_task.py:60-73: 5 MC samples, sigma 0.1): none of the
    MNIST shortcuts apply as they stand (in_features 1 and 50: no 16-byte rows, no pre-sampled weights; one output) and
    every one of the step's forms must still agree with the eager loop on the same Philox elements -- first step to
    fp32 / bf16 rounding, then the loss must go down."""
    import networks
    from bnn_hip.optim import FusedAdam
    from bnn_hip.train import GraphedTrainStep
    dev = torch.device("cuda:0")
    bnn_hip.set_math(math_mode)
    mp = dict(input_shape=1, classes=1, batch_size=128, hidden_units=50, mode="regression", mu_init=[-0.2, 0.2],
              rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=local_reparam)
    torch.manual_seed(4)
    net_a = networks.BayesianNetwork(mp).to(dev).train()
    net_b = networks.BayesianNetwork(mp).to(dev).train()
    net_b.load_state_dict(net_a.state_dict())
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.uniform(0, 0.6, (128, 1)).astype(np.float32)).to(dev)
    y = (x + 0.3 * torch.sin(2 * np.pi * x)).contiguous()
    S = 5
    oa = FusedAdam(net_a.parameters(), lr=1e-3)
    ob = FusedAdam(net_b.parameters(), lr=1e-3, capturable=True)
    bnn_hip.manual_seed(31, counter=200)
    g = GraphedTrainStep(net_b, ob, x, y, S, sigma=0.1)
    first = [o.clone() for o in g.step(x, y, 0.01)]
    grads_b = [p.grad.clone() for p in g.params]
    bnn_hip.manual_seed(31, counter=200)
    oa.zero_grad()
    out = (net_a.sample_elbo_lr if local_reparam else net_a.sample_elbo)(x, y, 0.01, S, 0.1)
    out[0].backward()
    tol = 1e-5 if math_mode == "f32" else 2e-3
    for got, want in zip(first, out):
        assert float((got - want.detach()).abs().max()) <= tol * (float(want.detach().abs().max()) + 1e-6)
    specs = net_a._specs()
    grads_a = [p.grad for sp in specs for p in (sp.m.weight_mu, sp.m.weight_rho, sp.m.bias_mu, sp.m.bias_rho)]
    for ga, gb in zip(grads_a, grads_b):
        assert float((ga - gb).norm()) <= (1e-4 if math_mode == "f32" else 2e-2) * (float(ga.norm()) + 1e-9)
    losses = [float(first[0])] + [float(g.step(x, y, 0.01)[0]) for _ in range(150)]
    assert np.isfinite(losses).all() and np.mean(losses[-10:]) < 0.7 * np.mean(losses[:10])
    bnn_hip.set_math("f32")


@pytest.mark.parametrize("autograd", [False, True])
@pytest.mark.parametrize("local_reparam", [False, True])
def test_graphed_train_step_equals_eager_steps(local_reparam, autograd):
    """train.GraphedTrainStep (zero_grad -> sample_elbo -> backward -> Adam in one hipGraph, device
    sample counter / Adam step / beta) reproduces the eager loop of class_task.py:66-79 step by
    step: same parameters after every step (same Philox sample indices, fp32 math)."""
    import networks
    from bnn_hip.optim import FusedAdam
    from bnn_hip.train import GraphedTrainStep
    dev = torch.device("cuda:0")
    bnn_hip.set_math("f32")
    mp = dict(input_shape=784, classes=10, batch_size=64, hidden_units=96, mode="classification", mu_init=[-0.2, 0.2],
              rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=local_reparam)
    torch.manual_seed(1)
    net_a = networks.BayesianNetwork(mp).to(dev).train()
    net_b = networks.BayesianNetwork(mp).to(dev).train()
    net_b.load_state_dict(net_a.state_dict())
    rs = np.random.RandomState(2)
    xs = [torch.from_numpy(rs.uniform(0, 1, (64, 1, 28, 28)).astype(np.float32)).to(dev) for _ in range(4)]
    ys = [torch.from_numpy(rs.randint(0, 10, 64)).to(dev) for _ in range(4)]
    S, M = 2, 4
    oa = FusedAdam(net_a.parameters(), lr=1e-3)
    ob = FusedAdam(net_b.parameters(), lr=1e-3, capturable=True)
    sched = torch.optim.lr_scheduler.StepLR(ob, step_size=2, gamma=0.5)
    sched_a = torch.optim.lr_scheduler.StepLR(oa, step_size=2, gamma=0.5)
    bnn_hip.manual_seed(99, counter=500)
    before = {k: v.clone() for k, v in net_b.state_dict().items()}
    graphed = GraphedTrainStep(net_b, ob, xs[0], ys[0], S, autograd=autograd)
    for k, v in net_b.state_dict().items():                    # building the graph left the model untouched
        assert torch.equal(v, before[k]), k
    assert ob.device_step() == 0
    outs_b = []
    for idx in range(M):
        outs_b.append([o.clone() for o in graphed.step(xs[idx], ys[idx], _beta(M, idx))])
        sched.step()
    assert ob.device_step() == M and int(graphed.counter.item()) == M * S
    bnn_hip.manual_seed(99, counter=500)
    elbo = net_a.sample_elbo_lr if local_reparam else net_a.sample_elbo
    for idx in range(M):
        oa.zero_grad()
        out = elbo(xs[idx], ys[idx], _beta(M, idx), S)
        out[0].backward()
        oa.step()
        sched_a.step()
        for got, want in zip(outs_b[idx], out):
            assert float((got - want.detach()).abs().max()) <= 1e-5 * (float(want.detach().abs().max()) + 1e-6), idx
    for (k, a), (_, b) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), k
    if not autograd:
        # a SECOND step object on the trained optimiser (another minibatch shape, e.g. the 96-row last MNIST batch):
        # its warm-up must leave the device step, the moments and the parameters where training left them, and a
        # checkpoint of the optimiser must carry the real step count
        before = {k: v.clone() for k, v in net_b.state_dict().items()}
        m_before = [ob.state[p]["exp_avg"].clone() for p in net_b.parameters()]
        graphed2 = GraphedTrainStep(net_b, ob, xs[0][:32], ys[0][:32], S)
        assert ob.device_step() == M
        for k, v in net_b.state_dict().items():
            assert torch.equal(v, before[k]), k
        for p, m in zip(net_b.parameters(), m_before):
            assert torch.equal(ob.state[p]["exp_avg"], m)
        sd = ob.state_dict()
        assert all(int(st["step"]) == M for st in sd["state"].values())
        # the two objects ALTERNATE (full batches and a short last batch, epoch after epoch): they share the optimiser's
        # one device sample counter, so every step draws the next unused global sample indices -- the eager loop on the
        # other replica, walking the host counter, sees the same epsilon, step by step
        plan = [(graphed2, xs[1][:32], ys[1][:32], 0.1), (graphed, xs[2], ys[2], 0.05), (graphed2, xs[3][:32], ys[3][:32], 0.02),
                (graphed, xs[0], ys[0], 0.01)]
        for j, (g, xb, yb, beta) in enumerate(plan):
            c = bnn_hip.runtime.state.counter
            assert c == 500 + (M + j) * S
            got = [o.clone() for o in g.step(xb, yb, beta)]
            assert ob.device_step() == M + 1 + j and int(graphed.counter.item()) == (M + 1 + j) * S
            assert graphed.counter is graphed2.counter and bnn_hip.runtime.state.counter == c + S
            oa.zero_grad()
            bnn_hip.manual_seed(99, counter=c)                # the replica draws the indices that replay drew
            out = elbo(xb, yb, beta, S)
            out[0].backward()
            oa.step()
            for a, b in zip(got, out):
                assert float((a - b.detach()).abs().max()) <= 1e-5 * (float(b.detach().abs().max()) + 1e-6), j
        for (k, a), (_, b) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), k
        # a checkpoint loaded AFTER the step objects exist: the device words the captured graphs read are updated in place
        # (step count -> bias correction, learning rate), and the next replay is torch.optim.Adam's update from that state
        sd = {"state": {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in st.items()} for k, st in ob.state_dict()["state"].items()},
              "param_groups": [dict(g) for g in ob.state_dict()["param_groups"]]}
        # ... with moments that DIFFER from the live ones (2 m, 3 v): torch's load_state_dict swaps in new tensors, the
        # captured graphs hold the old addresses -- the loaded values must land in the tensors the graphs update
        for st in sd["state"].values():
            st["step"] = 3
            st["exp_avg"] = st["exp_avg"] * 2.0
            st["exp_avg_sq"] = st["exp_avg_sq"] * 3.0
        sd["param_groups"][0]["lr"] = 7e-4
        words = [id(w) for w in ob._dev[0][:2]]
        moments = [(ob.state[p]["exp_avg"].data_ptr(), ob.state[p]["exp_avg_sq"].data_ptr()) for p in net_b.parameters()]
        ob.load_state_dict(sd)
        assert [id(w) for w in ob._dev[0][:2]] == words and ob.device_step() == 3
        assert moments == [(ob.state[p]["exp_avg"].data_ptr(), ob.state[p]["exp_avg_sq"].data_ptr()) for p in net_b.parameters()]
        for p, st in zip(net_b.parameters(), sd["state"].values()):
            assert torch.equal(ob.state[p]["exp_avg"], st["exp_avg"]) and torch.equal(ob.state[p]["exp_avg_sq"], st["exp_avg_sq"])
        ref = torch.optim.Adam(net_a.parameters(), lr=7e-4)
        net_a.load_state_dict(net_b.state_dict())
        ref.load_state_dict({"state": {k: {"step": torch.tensor(3.0), "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()}
                                       for k, st in sd["state"].items()}, "param_groups": ref.state_dict()["param_groups"]})
        got = graphed.step(xs[1], ys[1], 0.03)
        assert ob.device_step() == 4
        ref.zero_grad()
        bnn_hip.manual_seed(99, counter=500 + (M + len(plan)) * S)     # the indices that replay drew
        out = elbo(xs[1], ys[1], 0.03, S)
        out[0].backward()
        ref.step()
        for (k, a), (_, b) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), k


def test_out_of_range_label_poisons_loss_and_gradient():
    """A label outside [0, classes) must not train silently on a wrong loss (nn.CrossEntropyLoss, reference
    networks.py:186, raises): the NLL and the row's logit gradient come out NaN."""
    from bnn_hip import ops
    dev = torch.device("cuda:0")
    S, B, Cc = 2, 8, 10
    logits = torch.randn(S, B, Cc, device=dev)
    good = torch.randint(0, Cc, (B,), device=dev)
    bad = good.clone()
    bad[3] = Cc
    fin = lambda tg: ops.elbo_finalize(workspaces=[], layer_in=[], layer_out=[], local_reparam=False, prior=ops.PriorSpec(),
                                       n_samples=S, logits=logits, target=tg, mode="classification")["nll"]
    assert torch.isfinite(fin(good)).all() and torch.isnan(fin(bad)).all()
    g = ops.nll_bwd(logits, bad, torch.ones(S, device=dev), "classification")
    assert torch.isnan(g[:, 3]).all() and torch.isfinite(g[:, [0, 1, 2, 4, 5, 6, 7]]).all()


def test_graphed_step_with_presampled_weights_equals_the_fused_step(monkeypatch):
    """bf16 math: the hand-chained step that samples every layer once per step (bnn_bbb_sample_weights), runs
    matmul-only forwards and reuses the sampled weights in the input-gradient launches, against the same step with
    sampling fused into every launch: same Philox elements, so parameters agree to fp32 summation order."""
    import networks
    from bnn_hip import train
    from bnn_hip.optim import FusedAdam
    dev = torch.device("cuda:0")
    bnn_hip.set_math("bf16")
    mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=400, mode="classification", mu_init=[-0.2, 0.2],
              rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=False)
    rs = np.random.RandomState(5)
    xs = [torch.from_numpy(rs.uniform(0, 1, (128, 784)).astype(np.float32)).to(dev) for _ in range(3)]
    ys = [torch.from_numpy(rs.randint(0, 10, 128)).to(dev) for _ in range(3)]
    res = []
    for pre in (True, False):
        monkeypatch.setattr(train, "TRAIN_PRESAMPLE", pre)
        torch.manual_seed(3)
        net = networks.BayesianNetwork(mp).to(dev).train()
        opt = FusedAdam(net.parameters(), lr=1e-3, capturable=True)
        bnn_hip.manual_seed(17, counter=900)
        g = train.GraphedTrainStep(net, opt, xs[0], ys[0], 2)
        assert g.presample == pre
        first = [o.clone() for o in g.step(xs[0], ys[0], 0.25)]
        grads1 = [p.grad.clone() for p in g.params]                      # gradients of the FIRST step: identical parameters
        outs = [first] + [[o.clone() for o in g.step(xs[i], ys[i], 0.25)] for i in range(1, 3)]
        res.append((outs, grads1))
    (oa, ga), (ob, gb) = res
    for u, v in zip(oa[0], ob[0]):
        assert float((u - v).abs().max()) <= 2e-5 * (float(v.abs().max()) + 1e-6)
    for u, v in zip(ga, gb):
        # in norm: the two forwards sum in different orders, so the ReLU mask of an activation within rounding of zero
        # may differ and move a handful of gradient elements
        assert float((u - v).norm()) <= 2e-4 * (float(v.norm()) + 1e-9)
    # later steps: Adam turns rounding-level gradient differences into lr-sized parameter differences (sign of a
    # near-zero gradient), so only the losses are compared, loosely
    for a, b in zip(oa[1:], ob[1:]):
        for u, v in zip(a, b):
            assert float((u - v).abs().max()) <= 5e-3 * (float(v.abs().max()) + 1e-6)


DP_WORKER = r'''
import os, sys
sys.path.insert(0, r"{repo}"); sys.path.insert(0, os.path.join(r"{repo}", "bayesian-neural-network_amd"))
import numpy as np, torch, torch.distributed as dist
import bnn_hip, networks
from bnn_hip.optim import FusedAdam
from bnn_hip.train import GraphedTrainStep, broadcast_parameters
rank, world, lr_flag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[4] == "lr"
g16 = len(sys.argv) > 5 and sys.argv[5] == "bf16"
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks on cuda:0 (one-GPU box): gloo, not RCCL
dev = torch.device("cuda:0")
bnn_hip.set_math("f32")
mp = dict(input_shape=784, classes=10, batch_size=32, hidden_units=64, mode="classification", mu_init=[-0.2, 0.2],
          rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=lr_flag)
torch.manual_seed(100 + rank)                                    # replicas start DIFFERENT ...
net = networks.BayesianNetwork(mp).to(dev).train()
broadcast_parameters(net)                                        # ... and are made identical here
init = {{k: v.clone() for k, v in net.state_dict().items()}}
rs = np.random.RandomState(7)
T, S, M = 3, 2, 4
xs = [[torch.from_numpy(rs.uniform(0, 1, (32, 1, 28, 28)).astype(np.float32)).to(dev) for _ in range(T)] for _ in range(world)]
ys = [[torch.from_numpy(rs.randint(0, 10, 32)).to(dev) for _ in range(T)] for _ in range(world)]
beta = lambda idx: 2 ** (M - (idx + 1)) / (2 ** M - 1)
bnn_hip.manual_seed(5, counter=1000)
opt = FusedAdam(net.parameters(), lr=1e-3, capturable=True)
step = GraphedTrainStep(net, opt, xs[rank][0], ys[rank][0], S, data_parallel=True,
                        grad_dtype=torch.bfloat16 if g16 else torch.float32)
for t in range(T):
    step.step(xs[rank][t], ys[rank][t], beta(t))
torch.cuda.synchronize()
# every replica ends identical
flat = torch.cat([p.detach().flatten() for p in net.parameters()])
other = flat.clone()
dist.all_reduce(other, op=dist.ReduceOp.SUM)
assert float((other - world * flat).abs().max()) <= 1e-6 * float(flat.abs().max()), "replicas diverged"
if rank == 0:
    # single-process reference: average the eager gradients of the ranks' minibatches at the very same
    # Philox sample indices, one Adam step per round
    ref = networks.BayesianNetwork(mp).to(dev).train()
    ref.load_state_dict(init)
    ropt = FusedAdam(ref.parameters(), lr=1e-3)
    elbo = ref.sample_elbo_lr if lr_flag else ref.sample_elbo
    for t in range(T):
        acc = [torch.zeros_like(p) for p in ref.parameters()]
        for r in range(world):
            bnn_hip.manual_seed(5, counter=1000 + t * S * world + r * S)
            ropt.zero_grad()
            elbo(xs[r][t], ys[r][t], beta(t), S)[0].backward()
            for a, p in zip(acc, ref.parameters()):
                a += p.grad / world
        for a, p in zip(acc, ref.parameters()):
            p.grad = a
        ropt.step()
    for (k, a), (_, b) in zip(ref.state_dict().items(), net.state_dict().items()):
        err = float((a - b).abs().max())
        if not g16:
            assert err <= 2e-5 * float(a.abs().max()), (k, err)
        else:
            # bf16 bucket: each rank's finished gradient and the collective's sum are rounded to 8 significant bits.  Adam's
            # update is the NORMALISED m / sqrt(v) (about +-lr per step early on), so an element whose per-rank gradients
            # nearly cancel -- the rounding of the terms then exceeds their sum -- can step the other way: the bound per
            # element is Adam's own (2 lr per step), what is small is the share of such elements, i.e. the L2 distance of
            # the parameter displacement from the fp32-bucket run's (stated tolerance: 10 %)
            d_ref, d_got = (a - init[k].to(a.device)).double(), (b - init[k].to(a.device)).double()
            rel = float((d_got - d_ref).norm() / (d_ref.norm() + 1e-30))
            print(f"bf16 bucket {{k}}: max |dp| {{err:.2e}}, relative L2 of the displacement {{rel:.3e}}", flush=True)
            assert err <= 2.0 * T * 1e-3 * 1.0001 and rel <= 0.10, (k, err, rel)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("variant", ["bbb", "lr", "bbb-bf16", "lr-bf16"])
def test_data_parallel_train_step_two_ranks(tmp_path, variant):
    """F2, second half: GraphedTrainStep(data_parallel=True) on 2 ranks (both on this one GPU, so
    gloo instead of RCCL): one flat gradient bucket, one sum all-reduce per step between the two
    captured graphs; replicas stay identical and match a single-process run that averages the
    ranks' gradients.  `-bf16`: the bucket is all-reduced in bf16 (half the bytes on the wire) and Adam reads the bf16
    sums: replicas still identical, parameters within the stated bound of the fp32-bucket reference."""
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER.format(repo=repo))
    port = str(31500 + (os.getpid() % 2000))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port] + variant.split("-"), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


@pytest.mark.parametrize("config", [("bbb", False), ("bbb", True), ("lr", False)])
def test_whole_network_autograd_node_equals_per_layer_nodes(config):
    """functional.ElboFn (sample_elbo* as ONE autograd node: layer kernels + finalize forward, hand-chained
    backward kernels) gives the loss tuple and the gradients of the per-layer autograd path on the same
    Philox sample indices — classification and regression, Gaussian and mixture prior, LR."""
    import networks
    from bnn_hip import engine
    variant, mixture = config
    dev = torch.device("cuda:0")
    bnn_hip.set_math("f32")
    for mode, dims, B in (("classification", (784, 96, 10), 64), ("regression", (1, 50, 1), 37)):
        mp = dict(input_shape=dims[0], classes=dims[2], batch_size=B, hidden_units=dims[1], mode=mode, mu_init=[-0.2, 0.2],
                  rho_init=[-5, -4], prior_init=[0.5, 0.0, -6.0] if mixture else [1.0], mixture_prior=mixture,
                  local_reparam=variant == "lr")
        torch.manual_seed(3)
        net = networks.BayesianNetwork(mp).to(dev).train()
        rs = np.random.RandomState(4)
        if mode == "classification":
            x = torch.from_numpy(rs.uniform(0, 1, (B, 1, 28, 28)).astype(np.float32)).to(dev)
            y = torch.from_numpy(rs.randint(0, 10, B)).to(dev)
        else:
            x = torch.from_numpy(rs.uniform(0, 0.5, (B, 1)).astype(np.float32)).to(dev)
            y = torch.from_numpy(rs.uniform(-1, 1, (B, 1)).astype(np.float32)).to(dev)
        elbo = net.sample_elbo_lr if variant == "lr" else net.sample_elbo
        res = {}
        try:
            for fused in (True, False):
                engine.FUSED_ELBO_NODE = fused
                bnn_hip.manual_seed(21, counter=77)
                net.zero_grad()
                out = elbo(x, y, 0.37, 3, 0.4)
                out[0].backward()
                res[fused] = ([o.detach().clone() for o in out], [p.grad.clone() for p in net.parameters()])
        finally:
            engine.FUSED_ELBO_NODE = True
        for a, b in zip(res[True][0], res[False][0]):
            assert tuple(a.shape) == tuple(b.shape)
            assert float((a - b).abs().max()) <= 1e-5 * (float(b.abs().max()) + 1e-6)
        for (name, _), a, b in zip(net.named_parameters(), res[True][1], res[False][1]):
            assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-9), (mode, name)
