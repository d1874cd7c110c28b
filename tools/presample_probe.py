"""One minibatch, S MC samples (C4's per-GPU share is 8): the evaluation as ONE sampling launch for all three layers (K1s) + matmul-only
layer launches over the sampled weights + the row-split output layer, against the product path (K-sliced K1b with fused sampling)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine, ops, _lib as L
from bnn_hip.runtime import state
import bench

dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
net, x, y = bench.build_net(bench.DIMS["mnist"], False, 128, dev, "classification", n_minibatches=1)
specs = net._specs()
xf = net._flat(x[0]).contiguous()

def timed(fn, n=200):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / n

for S in (4, 8, 16, 32):
    ev = engine.GraphedElbo(net, x[0], y[0], S)
    for _ in range(10):
        ev.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ev.replay()
    torch.cuda.synchronize()
    us_prod = (time.perf_counter() - t0) * 1e6 / 200
    # the pre-sampled chain
    x16 = torch.empty(xf.shape, dtype=torch.bfloat16, device=dev)
    layers = []
    for sp in specs:
        k, n = sp.in_out
        layers.append(dict(w_mu=sp.m.weight_mu.detach(), w_rho=sp.m.weight_rho.detach(), b_mu=sp.m.bias_mu.detach(), b_rho=sp.m.bias_rho.detach(),
                           prior=sp.m._prior_spec, layer_id=sp.layer_id, workspace=ops.sample_workspace(S, k, n, dev),
                           w_out=torch.empty((S, n, k), dtype=torch.bfloat16, device=dev), b_out=torch.empty((S, n), dtype=torch.float32, device=dev)))
    bufs = [torch.empty((S, 128, sp.in_out[1]), dtype=torch.float32 if i == 2 else torch.bfloat16, device=dev) for i, sp in enumerate(specs)]
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    out = {k: torch.zeros(S, dtype=torch.float32, device=dev) for k in ("log_prior", "log_q", "nll")}
    sums = torch.zeros((1, 4), dtype=torch.float32, device=dev)
    ticket = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = ops.final_scratch(S, dev)
    def chain():
        ops.bbb_sample_weights(layers, n_samples=S, seed=state.seed, sample_offset=0, sample_counter=counter, cast=(xf, x16))
        h = x16
        for i, sp in enumerate(specs[:2]):
            h = ops.bbb_sampled_matmul(h, layers[i]["w_out"], layers[i]["b_out"], n_samples=S, relu=True, y_dtype=torch.bfloat16, out=bufs[i])
        fin_kw = dict(layer_in=[sp.in_out[0] for sp in specs], layer_out=[sp.in_out[1] for sp in specs], local_reparam=False,
                      prior=specs[0].m._prior_spec, n_samples=S, target=y[0], mode="classification", nll_sigma=1.0, sample_counter=counter,
                      sample_counter_inc=S, out=out, sums=sums, ticket=ticket, scratch=scratch, group_samples=0)
        ops.bbb_final_fwd((h, None, None, None, None),
                          dict(n_samples=S, prior=specs[2].m._prior_spec, math_mode=L.MATH_BF16, relu=False, y_dtype=torch.float32, eps_mode=L.EPS_ZERO,
                               want_stats=False, out=bufs[2], w_sampled=layers[2]["w_out"], b_sampled=layers[2]["b_out"]),
                          dict(workspaces=[l["workspace"] for l in layers], **fin_kw))
    us_pre = timed(chain)
    # parts
    us_s = timed(lambda: ops.bbb_sample_weights(layers, n_samples=S, seed=state.seed, sample_offset=0, sample_counter=counter, cast=(xf, x16)))
    us_m1 = timed(lambda: ops.bbb_sampled_matmul(x16, layers[0]["w_out"], layers[0]["b_out"], n_samples=S, relu=True, y_dtype=torch.bfloat16, out=bufs[0]))
    us_m2 = timed(lambda: ops.bbb_sampled_matmul(bufs[0], layers[1]["w_out"], layers[1]["b_out"], n_samples=S, relu=True, y_dtype=torch.bfloat16, out=bufs[1]))
    print(f"S={S:3d}: product path {us_prod:7.1f} us | pre-sampled chain {us_pre:7.1f} us (sampling launch {us_s:6.1f}, matmul 1 {us_m1:6.1f}, matmul 2 {us_m2:6.1f}; each incl. ~8 us replay gap)", flush=True)
