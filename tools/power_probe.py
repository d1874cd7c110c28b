"""Board power and shader clock (rocm-smi) while the headline kernel (K1b2, 1200 x 1200, 256 pairs) runs back to back for a few
seconds, against an idle reading and the eps = 0 launch.  Measurement tool: is the kernel running into the power limit?"""
import os, subprocess, sys, threading, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
from bnn_hip import ops, _lib as L

def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = d.get("card0", {})
        return {k: v for k, v in c.items() if any(t in k.lower() for t in ("power", "sclk", "mclk"))}
    except Exception as e:
        return {"error": repr(e)}

dev = torch.device("cuda:0")
S, B, K, N = 256, 128, 1200, 1200
g = torch.Generator().manual_seed(3)
w_mu = ((torch.rand(N, K, generator=g) - 0.5) * 0.4).to(dev); w_rho = (-5 + torch.rand(N, K, generator=g)).to(dev)
b_mu = ((torch.rand(N, generator=g) - 0.5) * 0.4).to(dev); b_rho = (-5 + torch.rand(N, generator=g)).to(dev)
x = torch.rand(S, B, K, generator=g).to(dev).to(torch.bfloat16)
sig = torch.log1p(torch.exp(w_rho)); out = torch.empty((S, B, N), dtype=torch.bfloat16, device=dev); ws = ops.bbb_workspace(S, N, dev)
print("idle:", smi(), flush=True)
for name, eps in (("generator on", L.EPS_PHILOX), ("eps = 0", L.EPS_ZERO)):
    kw = dict(n_samples=S, prior=ops.PriorSpec(False, 1.0), math_mode=L.MATH_BF16, relu=True, y_dtype=torch.bfloat16, seed=1, layer_id=1,
              workspace=ws, out=out, form=L.FORM_GEMM, eps_mode=eps, want_stats=True, w_sigma=sig)
    g_ = torch.cuda.CUDAGraph()
    ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw); torch.cuda.synchronize()
    with torch.cuda.graph(g_):
        for _ in range(50):
            ops.bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw)
    stop, samples = [False], []
    def poll():
        while not stop[0]:
            samples.append(smi()); time.sleep(0.2)
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        g_.replay(); n += 50
        torch.cuda.synchronize()
    e1.record(); e1.synchronize()
    stop[0] = True; th.join()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / n:.1f} us per launch over {n} launches", flush=True)
    for s_ in samples[2:8]:
        print("   ", s_, flush=True)
