"""K2 measurement: achieved HBM GB/s of the standalone closed-form KL kernel (bnn_gauss_kl: one
streaming pass over (mu, rho), 8 B per element), HIP events over back-to-back launches."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO)
import torch
from bnn_hip import ops
dev = torch.device("cuda:0")
rows = []
sizes = [int(a) for a in sys.argv[1:]] or [1200 * 1200, 4096 * 4096, 16 * 4096 * 4096]   # 268 M elements = 2.1 GB: above the Infinity Cache
for n in sizes:
    mu = torch.empty(n, device=dev).uniform_(-0.2, 0.2); rho = torch.empty(n, device=dev).uniform_(-5, -4)
    for _ in range(5): ops.gauss_kl(mu, rho, 1.0)
    torch.cuda.synchronize()
    reps = 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): ops.gauss_kl(mu, rho, 1.0)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    rows.append(dict(elements=n, us=us, GBps=8.0 * n / us / 1e3, frac_of_8TBps=8.0 * n / us / 1e3 / 8000.0))
    print(f"gauss_kl {n:>10d} elements: {us:8.1f} us  {rows[-1]['GBps']:8.0f} GB/s  ({rows[-1]['frac_of_8TBps']:.2f} of 8 TB/s)", flush=True)
print(json.dumps(rows))
