"""Diagnostic: which part of a one-sample evaluation saturates the chip when NS evaluations run side by side (one
graph + stream each, as bench.py does)?  Times per-evaluation cost of chains made of only some of the launches.
usage: chain_parts.py [streams=4] [per_graph=4]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import torch
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PG = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream() for _ in range(NS)]
for st in streams:                      # bind the hardware queues first (see bench.py)
    with torch.cuda.stream(st):
        torch.zeros(1, device=dev)
torch.cuda.synchronize()
from bnn_hip import ops, _lib as L
B = 128
dims = [(784, 1200), (1200, 1200)]
prior = ops.PriorSpec(False, 1.0)
torch.manual_seed(0)

def make_state():
    layers = []
    for i, (K, N) in enumerate(dims):
        layers.append(dict(w_mu=torch.empty(N, K, device=dev).uniform_(-0.2, 0.2), w_rho=torch.empty(N, K, device=dev).uniform_(-5, -4),
                           b_mu=torch.empty(N, device=dev).uniform_(-0.2, 0.2), b_rho=torch.empty(N, device=dev).uniform_(-5, -4),
                           prior=prior, layer_id=i))
    res = ops.bbb_sample_weights(layers, n_samples=1, seed=1)
    for ly, r in zip(layers, res):
        ly.update(workspace=r["workspace"], w_out=r["w"], b_out=r["b"])
    st = dict(layers=layers, res=res, x=torch.rand(B, 784, device=dev), x16=torch.rand(B, 784, device=dev).to(torch.bfloat16),
              y1=torch.empty(1, B, 1200, dtype=torch.bfloat16, device=dev), y2=torch.empty(1, B, 1200, dtype=torch.bfloat16, device=dev))
    return st

def k1s(s): ops.bbb_sample_weights(s["layers"], n_samples=1, seed=1, cast=(s["x"], s["x16"]))
def mm(s):
    ops.bbb_sampled_matmul(s["x16"], s["res"][0]["w"], s["res"][0]["b"], n_samples=1, relu=True, y_dtype=torch.bfloat16, out=s["y1"], concurrency=NS)
    ops.bbb_sampled_matmul(s["y1"], s["res"][1]["w"], s["res"][1]["b"], n_samples=1, relu=True, y_dtype=torch.bfloat16, out=s["y2"], concurrency=NS)
def fused(s):
    kw = dict(n_samples=1, prior=prior, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=1, want_stats=True, concurrency=NS)
    p0 = [s["layers"][0][k] for k in ("w_mu", "w_rho", "b_mu", "b_rho")]; p1 = [s["layers"][1][k] for k in ("w_mu", "w_rho", "b_mu", "b_rho")]
    ops.bbb_linear_fwd(s["x"], *p0, layer_id=0, workspace=s["layers"][0]["workspace"], out=s["y1"], **kw)
    ops.bbb_linear_fwd(s["y1"], *p1, layer_id=1, workspace=s["layers"][1]["workspace"], out=s["y2"], **kw)

variants = {"K1s only": [k1s], "two matmul-only layers": [mm], "K1s + matmuls (split hidden layers)": [k1s, mm], "two fused K1a layers": [fused]}
states = [make_state() for _ in range(NS)]
for name, parts in variants.items():
    graphs = []
    for st, s in zip(streams, states):
        with torch.cuda.stream(st):
            for f in parts: f(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(PG):
                    for f in parts: f(s)
        graphs.append(g)
    torch.cuda.synchronize()
    def run(n):
        for _ in range(n):
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
    run(30); torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300; run(n); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n * NS * PG) * 1e6
    print(f"{name:40s} {dt:6.2f} us per evaluation ({NS} streams, {PG} per graph)", flush=True)
