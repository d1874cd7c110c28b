"""Diagnostic: the headline launch group's layer launches timed IN the step (HIP events around each launch of an eager, uncaptured
evaluation) against the same launch repeated alone -- does a layer run slower behind its predecessor than back to back with itself?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch, bnn_hip
from bnn_hip import engine, ops
from bench import build_net, DIMS

dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
G = 256
net, x, y = build_net(DIMS["mnist"], False, 128, dev, "classification", n_minibatches=G)
ev = engine.GraphedElbo(net, x, y, 1, capture=False, stacked=True)
marks = []
orig = ops.bbb_linear_fwd
def timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(*a, **k); e1.record()
    marks.append((e0, e1, a, k))
    return r
ops.bbb_linear_fwd = timed
for _ in range(3): ev.replay()
torch.cuda.synchronize(); marks.clear()
for _ in range(10): ev.replay()
torch.cuda.synchronize()
per = {}
for i, (e0, e1, a, k) in enumerate(marks):
    per.setdefault(i % 2, []).append(e0.elapsed_time(e1) * 1e3)
for li, v in per.items():
    print(f"layer {li + 1} in the step: {sum(v) / len(v):8.1f} us (min {min(v):.1f}, max {max(v):.1f})")
ops.bbb_linear_fwd = orig
# the same launches alone, back to back
for li in (0, 1):
    _, _, a, k = marks[li]
    for _ in range(3): orig(*a, **k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): orig(*a, **k)
    e1.record(); e1.synchronize()
    print(f"layer {li + 1} alone, back to back: {e0.elapsed_time(e1) * 100:8.1f} us")
# alone, but behind 512 MB of unrelated writes (L2 and the 256 MB Infinity Cache hold nothing of the layer when it starts)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for li in (0, 1):
    _, _, a, k = marks[li]
    tot = 0.0
    for i in range(10):
        junk.fill_(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(*a, **k); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    print(f"layer {li + 1} alone, behind 512 MB of unrelated writes: {tot / 10:8.1f} us")
