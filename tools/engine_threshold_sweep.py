"""BBB / LR network (784-1200-1200-10, one minibatch of 128): us per evaluation (engine.GraphedElbo) by MC samples per evaluation
with engine thresholds overridden -- ENGINE_SETS="NAME=v,NAME=v;NAME=v" (';' separates the variants, the product's values always
run first).  Measurement tool."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bayesian-neural-network_amd"), REPO]
import torch
import bnn_hip
from bnn_hip import engine
from bench import build_net, DIMS

dev = torch.device("cuda:0")
bnn_hip.set_math("bf16")
lr = os.environ.get("VARIANT", "bbb") == "lr"
sets = [{}] + [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv) for v in os.environ.get("ENGINE_SETS", "").split(";") if v]
base = {k: getattr(engine, k) for st in sets for k in st}
net, x, y = build_net(DIMS["mnist"], lr, 128, dev, "classification", n_minibatches=1)
for S in [int(v) for v in os.environ.get("SWEEP_S", "2,3,4,5,6,8").split(",")]:
    row = [f"S={S:2d}"]
    for rnd in range(2):
        for st in sets:
            for k, v in base.items():
                setattr(engine, k, st.get(k, v))
            ev = engine.GraphedElbo(net, x[0], y[0], S)
            for _ in range(20):
                ev.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 300
            e0.record()
            for _ in range(n):
                ev.replay()
            e1.record()
            e1.synchronize()
            row.append(f"{','.join(f'{k}={v}' for k, v in st.items()) or 'product'}: {e0.elapsed_time(e1) * 1e3 / n:6.1f}")
            del ev
    print(" | ".join(row), flush=True)
