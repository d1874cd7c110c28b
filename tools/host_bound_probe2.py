"""Host cost of hipGraph replay vs node count, and of direct launches through the C ABI."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
import bnn_hip
from bnn_hip import ops

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
a = torch.zeros(64, device=dev)
b = torch.zeros(64, device=dev)
for nodes in (1, 2, 4, 8, 16, 32):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(nodes):
                ops.softplus(a, out=b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2000):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"graph of {nodes} serial kernels: host {1e6*(t1-t0)/2000:.2f} us/replay, total {1e6*(t2-t0)/2000:.2f}", flush=True)
# 4 parallel branches x 4 kernels in one graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
br = [torch.cuda.Stream() for _ in range(4)]
bufs = [torch.zeros(64, device=dev) for _ in range(4)]
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for st, bb in zip(br, bufs):
            st.wait_stream(s)
            with torch.cuda.stream(st):
                for _ in range(4):
                    ops.softplus(a, out=bb)
        for st in br:
            s.wait_stream(st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"graph of 4 branches x 4 kernels: host {1e6*(t1-t0)/2000:.2f} us/replay, total {1e6*(t2-t0)/2000:.2f}", flush=True)
# direct launches
t0 = time.perf_counter()
for i in range(8000):
    ops.softplus(a, out=b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"direct C-ABI launch via ops.softplus: host {1e6*(t1-t0)/8000:.2f} us/launch, total {1e6*(t2-t0)/8000:.2f}", flush=True)
t0 = time.perf_counter()
for i in range(8000):
    a.add_(1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"torch a.add_: host {1e6*(t1-t0)/8000:.2f} us/launch, total {1e6*(t2-t0)/8000:.2f}", flush=True)
