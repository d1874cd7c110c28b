"""F3 measurement: MC-averaged prediction of one minibatch (classification/class_task.py:81-87,
test_samples = 10 as in the reference's config.py) — the per-call Python loop over net(X, sample=True)
against predict_mc (samples batched per launch + bnn_mc_softmax_mean), eager and as a captured graph."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd")); sys.path.insert(0, REPO)
import torch, bnn_hip, networks
from bnn_hip import synth
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rows = []
for lr in (False, True):
    bnn_hip.set_math("bf16")
    mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification",
              mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=lr)
    net = networks.BayesianNetwork(mp).to(dev).eval()
    x, _ = synth.synth_batch("classification", 128, 784, 10)
    x = torch.from_numpy(x).to(dev)
    def loop():
        probs = torch.zeros(128, 10, device=dev)
        for _ in range(S):
            probs = probs + torch.softmax(net(x, sample=True), dim=1) / S
        return torch.argmax(probs, dim=1), probs
    def fused():
        return net.predict_mc(x, S)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad():
        fused(); torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                out = fused()
        torch.cuda.current_stream().wait_stream(side)
    with torch.no_grad():
        pred = net.predictor(x, S)                        # engine.GraphedPredict: captured, fresh eps per replay
    forms = (("python loop of net(X, sample=True)", loop, 30), ("predict_mc", fused, 100), ("predict_mc as a hipGraph*", g.replay, 300),
             ("net.predictor(x, S).replay()", pred.replay, 300),
             ("net.predictor(x, S, capture='calls').replay()", net.predictor(x, S, capture="calls").replay, 300))
    for name, fn, n in forms:
        with torch.no_grad():
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        rows.append(dict(variant="LR" if lr else "BBB", test_samples=S, form=name, us_per_minibatch=dt * 1e6))
        print(f"{'LR ' if lr else 'BBB'} S={S} {name:38s}: {dt*1e6:8.1f} us per 128-image minibatch", flush=True)
print("* the captured graph replays the SAME eps (host-side sample offset baked in): a timing of the kernels only")
print(json.dumps(rows))
