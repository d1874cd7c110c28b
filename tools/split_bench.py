"""Diagnostic: the split form of a one-sample evaluation's hidden layers (K1s sampling launch + matmul-only
launches) timed kernel by kernel against the fused K1a launches, MNIST shapes.  usage: split_bench.py [conc]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import torch
from bnn_hip import ops, _lib as L

dev = torch.device("cuda:0")
conc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
B = 128
dims = [(784, 1200), (1200, 1200)]
prior = ops.PriorSpec(False, 1.0)
layers = []
for i, (K, N) in enumerate(dims):
    layers.append(dict(w_mu=torch.empty(N, K, device=dev).uniform_(-0.2, 0.2), w_rho=torch.empty(N, K, device=dev).uniform_(-5, -4),
                       b_mu=torch.empty(N, device=dev).uniform_(-0.2, 0.2), b_rho=torch.empty(N, device=dev).uniform_(-5, -4),
                       prior=prior, layer_id=i))
x = torch.rand(B, 784, device=dev)
x16 = x.to(torch.bfloat16)
h1 = torch.rand(1, B, 1200, device=dev).to(torch.bfloat16)
res = ops.bbb_sample_weights(layers, n_samples=1, seed=1)
for ly, r in zip(layers, res):
    ly.update(workspace=r["workspace"], w_out=r["w"], b_out=r["b"])
y1 = torch.empty(1, B, 1200, dtype=torch.bfloat16, device=dev); y2 = torch.empty_like(y1)

def timeit(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

g = torch.cuda.CUDAGraph()
def graphed(fn, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    torch.cuda.current_stream().wait_stream(s)
    return lambda: g.replay(), reps

def report(name, fn):
    rp, reps = graphed(fn)
    print(f"{name:58s} {timeit(rp, 200) / reps:7.2f} us", flush=True)

report("K1s: sample both hidden layers (2.38 M weights)", lambda: ops.bbb_sample_weights(layers, n_samples=1, seed=1))
report("K1s: layer 1 only", lambda: ops.bbb_sample_weights(layers[:1], n_samples=1, seed=1))
report("K1s: layer 2 only", lambda: ops.bbb_sample_weights(layers[1:], n_samples=1, seed=1))
report("matmul-only layer 1 (bf16 x)", lambda: ops.bbb_sampled_matmul(x16, res[0]["w"], res[0]["b"], n_samples=1, relu=True, y_dtype=torch.bfloat16, out=y1, concurrency=conc))
report("matmul-only layer 2 (bf16 x)", lambda: ops.bbb_sampled_matmul(h1, res[1]["w"], res[1]["b"], n_samples=1, relu=True, y_dtype=torch.bfloat16, out=y2, concurrency=conc))
kw = dict(n_samples=1, prior=prior, math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, eps_mode=L.EPS_PHILOX, seed=1, want_stats=True, concurrency=conc)
p0 = [layers[0][k] for k in ("w_mu", "w_rho", "b_mu", "b_rho")]; p1 = [layers[1][k] for k in ("w_mu", "w_rho", "b_mu", "b_rho")]
report("fused K1a layer 1", lambda: ops.bbb_linear_fwd(x, *p0, layer_id=0, workspace=layers[0]["workspace"], out=y1, **kw))
report("fused K1a layer 2", lambda: ops.bbb_linear_fwd(h1, *p1, layer_id=1, workspace=layers[1]["workspace"], out=y2, **kw))
def chain():
    ops.bbb_sample_weights(layers, n_samples=1, seed=1)
    ops.bbb_sampled_matmul(x16, res[0]["w"], res[0]["b"], n_samples=1, relu=True, y_dtype=torch.bfloat16, out=y1, concurrency=conc)
    ops.bbb_sampled_matmul(y1, res[1]["w"], res[1]["b"], n_samples=1, relu=True, y_dtype=torch.bfloat16, out=y2, concurrency=conc)
report("split chain: K1s + 2 matmuls", chain)
def chain_f():
    ops.bbb_linear_fwd(x, *p0, layer_id=0, workspace=layers[0]["workspace"], out=y1, **kw)
    ops.bbb_linear_fwd(y1, *p1, layer_id=1, workspace=layers[1]["workspace"], out=y2, **kw)
report("fused chain: 2 x K1a", chain_f)
