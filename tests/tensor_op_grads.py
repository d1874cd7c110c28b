"""TEST INFRASTRUCTURE: the closed-form gradients of SURVEY A.5 evaluated with device tensor ops (torch matmuls), the
cross-check the GPU tests compare the HIP backward kernels with.  Lived in the product (bnn_hip/functional.py) as a
second backward until round 3; the product now raises where the kernels cannot run.  Golden G5 / G6 pin these formulas
and the kernels against the real reference's autograd."""
import torch

from bnn_hip import _lib as L
from bnn_hip import ops


def _softplus(rho):
    return torch.log1p(torch.exp(rho))


def bbb_grads(x, gy, glp, glq, y, w_mu, w_rho, b_mu, b_rho, eps_w, eps_b, call):
    """Gradients of sum(y * gy) + sum(log_prior * glp) + sum(log_q * glq) of one BayesianLinear launch w.r.t.
    (x, w_mu, w_rho, b_mu, b_rho); `call` = functional.LayerCall, `y` = the launch's output (the ReLU mask)."""
    S = call.n_samples
    N, K = w_mu.shape
    dev = w_mu.device
    if call.eps_mode == L.EPS_PHILOX:
        eps_w = ops.philox_normal(call.seed, call.layer_id * 4 + 0, call.sample_offset, S, N, K, dev)
        eps_b = ops.philox_normal(call.seed, call.layer_id * 4 + 1, call.sample_offset, S, 1, N, dev).view(S, N)
    elif call.eps_mode == L.EPS_ZERO:
        eps_w, eps_b = torch.zeros((S, N, K), device=dev), torch.zeros((S, N), device=dev)
    else:
        eps_w, eps_b = eps_w.view(S, N, K), eps_b.view(S, N)
    g = gy.float()
    if call.relu:
        g = g * (y > 0).to(g.dtype)
    sig_w, sig_b = _softplus(w_rho), _softplus(b_rho)
    W = w_mu + sig_w * eps_w                                   # [S,N,K]
    bvec = b_mu + sig_b * eps_b                                # [S,N]
    x3 = (x if x.dim() == 3 else x.unsqueeze(0).expand(S, -1, -1)).float()
    gW = torch.matmul(g.transpose(1, 2), x3)                   # [S,N,K]
    gb = g.sum(1)                                              # [S,N]
    if call.want_stats and glp is not None:
        pr = call.prior
        if pr.mixture:
            def dlogp(w):
                n1 = pr.pi * torch.exp(-w * w / (2 * pr.sigma1 ** 2)) / pr.sigma1
                n2 = (1 - pr.pi) * torch.exp(-w * w / (2 * pr.sigma2 ** 2)) / pr.sigma2
                return -w * (n1 / pr.sigma1 ** 2 + n2 / pr.sigma2 ** 2) / (n1 + n2)
        else:
            def dlogp(w):
                return -w / (pr.sigma_p ** 2)
        gW = gW + glp.view(S, 1, 1) * dlogp(W)
        gb = gb + glp.view(S, 1) * dlogp(bvec)
    g_wmu, g_wsig = gW.sum(0), (gW * eps_w).sum(0)
    g_bmu, g_bsig = gb.sum(0), (gb * eps_b).sum(0)
    if call.want_stats and glq is not None:
        c = glq.sum()
        g_wsig = g_wsig - c / sig_w                            # d(log q)/d(sigma) = -1/sigma
        g_bsig = g_bsig - c / sig_b
    gx = torch.matmul(g, W)                                    # [S,B,K]
    if x.dim() == 2:
        gx = gx.sum(0)
    return gx, g_wmu, g_wsig * torch.sigmoid(w_rho), g_bmu, g_bsig * torch.sigmoid(b_rho)


def lr_grads(x, gy, gkl3, y, w_mu, w_rho, b_mu, b_rho, eps_act, eps_b, call):
    """Gradients of sum(y * gy) + sum(kl3 * gkl3) of one BayesianLinearLR launch (weights [in,out])."""
    S = call.n_samples
    K, N = w_mu.shape
    dev = w_mu.device
    x3 = (x if x.dim() == 3 else x.unsqueeze(0).expand(S, -1, -1)).float()
    B = x3.shape[1]
    if call.eps_mode == L.EPS_PHILOX:
        eps_act = ops.philox_normal(call.seed, call.layer_id * 4 + 2, call.sample_offset, S, B, N, dev)
        eps_b = ops.philox_normal(call.seed, call.layer_id * 4 + 1, call.sample_offset, S, 1, N, dev).view(S, N)
    elif call.eps_mode == L.EPS_ZERO:
        eps_act, eps_b = torch.zeros((S, B, N), device=dev), torch.zeros((S, N), device=dev)
    else:
        eps_act, eps_b = eps_act.view(S, B, N), eps_b.view(S, N)
    g = gy.float()
    if call.relu:
        g = g * (y > 0).to(g.dtype)
    sig_w, sig_b = _softplus(w_rho), _softplus(b_rho)
    s2 = sig_w * sig_w
    xsq = x3 * x3
    sd = torch.sqrt(torch.matmul(xsq, s2))                    # [S,B,N]
    h = torch.where(sd > 0, g * eps_act / (2 * sd), torch.zeros_like(g))
    g_M = torch.matmul(x3.transpose(1, 2), g).sum(0)          # [K,N]
    g_s2 = torch.matmul(xsq.transpose(1, 2), h).sum(0)
    g_sig = 2 * sig_w * g_s2
    gsum = g.sum(1)                                           # [S,N]
    g_bmu, g_bsig = gsum.sum(0), (gsum * eps_b).sum(0)
    if call.want_stats and gkl3 is not None:
        sp2 = call.prior.sigma_p ** 2
        cw, cb = gkl3[0] + gkl3[1], gkl3[0] + gkl3[2]         # kl3 = (kl, weight_kl, bias_kl)
        g_M = g_M + cw * w_mu / sp2
        g_sig = g_sig + cw * (sig_w / sp2 - 1 / sig_w)
        g_bmu = g_bmu + cb * b_mu / sp2
        g_bsig = g_bsig + cb * (sig_b / sp2 - 1 / sig_b)
    gx = torch.matmul(g, w_mu.t()) + 2 * x3 * torch.matmul(h, s2.t())
    if x.dim() == 2:
        gx = gx.sum(0)
    return gx, g_M, g_sig * torch.sigmoid(w_rho), g_bmu, g_bsig * torch.sigmoid(b_rho)


def nll_grads(logits, target, gnll, mode, sigma):
    S = logits.shape[0]
    if mode == "classification":
        p = torch.softmax(logits.float(), dim=-1)
        onehot = torch.zeros_like(p[0]).scatter_(1, target.view(-1, 1).to(torch.int64), 1.0)
        return (p - onehot.unsqueeze(0)) * gnll.view(S, 1, 1)
    return (logits.float() - target.float().view(1, *logits.shape[1:])) / (sigma ** 2) * gnll.view(S, 1, 1)
