#!/usr/bin/env python3
"""bench.py — MC-forward-samples/sec (+ KL-elements/sec) of the Bayes-by-backprop hot path.

Workload (BASELINE.json configs[1], SURVEY §8(d) "C2"): the 784-1200-1200-10 BBB network,
batch 128, bf16 MFMA operands / fp32 statistics, synthetic inputs (mu~U(-0.2,0.2),
rho~U(-5,-4), x~U(0,1), labels~U{0..9}; numpy RandomState seeds 1234/5678), Gaussian prior
sigma_p=1, on-chip Philox epsilon.  One STEP = one forward-only ELBO evaluation of
`--samples` MC samples per GPU (default 1, as configs[1] names): for every sample the full
3-layer forward, the sampled log p(w) / log q(w) reductions over all 2 395 210 stochastic
parameters and the NLL (reference networks.py:199-203) — three kernel launches replayed as a
hipGraph.  `--streams` independent evaluations are kept in flight per GPU (one hipGraph +
HIP stream each; every evaluation still has `--samples` MC samples and its own Philox
sample indices).  With N>1 ranks every rank owns `--samples` samples of each evaluation
(weak scaling) and the only collective is one RCCL sum all-reduce of the 4 ELBO scalars per
step.

Prints ONE JSON line (rank 0).  value = total MC samples / s over all ranks; `roofline` is
the dominant kernel (layer 2, 1200x1200 weights) against the HBM roof with SURVEY §8(d)'s
algorithmic bytes; `cpu_baseline` is the parity-pinned CPU oracle timed on this host;
`extras` are the same network at larger MC batches per evaluation (throughput regime).
"""
import argparse
import contextlib
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
sys.path.insert(0, REPO)

import numpy as np
import torch

DIMS = {"mnist": (784, 1200, 10), "wide": (4096, 4096, 4096), "reg": (1, 50, 1)}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable


def algorithmic_bytes_layer(fin, fout, batch, x_bytes, y_bytes):
    """(mu, rho) fp32 read once per sample (8 B/param incl. bias, eps on chip,
    w never stored) + x read + y written."""
    return 8 * (fin * fout + fout) + batch * fin * x_bytes + batch * fout * y_bytes


def build_net(dims, lr, batch, device, mode="classification"):
    import networks
    from bnn_hip import synth
    mp = dict(input_shape=dims[0], classes=dims[2], batch_size=batch, hidden_units=dims[1], mode=mode,
              mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=lr)
    net = networks.BayesianNetwork(mp)
    sd = synth.synth_state_dict(dims[0], dims[1], dims[2], lr)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x, y = synth.synth_batch(mode, batch, dims[0], dims[2])
    return net.to(device).train(), torch.from_numpy(x).to(device), torch.from_numpy(y).to(device)


def n_stochastic(dims):
    return sum(i * o + o for i, o in ((dims[0], dims[1]), (dims[1], dims[1]), (dims[1], dims[2])))


def run_steps(evs, steps, warmup, dist, slab=None, ar_every=16, tail=None):
    """W untimed + K timed steps (a step = ONE ELBO evaluation) bracketed by barrier + synchronize;
    returns seconds (max over ranks).  An evaluator replay runs `evs[0].per_replay` consecutive
    evaluations (one hipGraph launch): steps and warmup must be multiples of it.
    `tail` = (evaluator with per_replay = r, its [r, 1, 4] slab or None): r more timed evaluations after the
    replays, so that ANY K is timed exactly whatever the evaluations per graph launch are.

    Multi-GPU: every evaluation's 4 ELBO scalars are sum-all-reduced over RCCL.  The evaluators'
    graphs deposit them in consecutive rows of `slab` [2*ar_every, n_evaluators, 4] (device-side
    ring cursor, bnn_finalize_args.sums_ring_pos), and each time a half of the ring is full it is
    all-reduced with ONE asynchronous call (the message is latency-bound either way) that overlaps
    the evaluations filling the other half: the per-step host work is the graph launch alone."""
    nstr = len(evs)
    E = evs[0].per_replay
    assert steps % E == 0 and warmup % E == 0 and (dist is None or ar_every % E == 0)
    have_gpu = torch.cuda.is_available()                   # (the CPU test of this bookkeeping runs it over gloo)
    main = torch.cuda.current_stream() if have_gpu else None
    per_flush = nstr * ar_every // E                    # replays between two all-reduces
    works = [None, None]
    flushed = []

    def on_streams(fn):
        for e in evs:
            if e.stream is not None:
                with torch.cuda.stream(e.stream):
                    fn()
            else:
                fn()

    def flush(half, partial=False):
        for e in evs:                                   # the rows were written on the evaluators' streams
            if e.stream is not None:
                main.wait_stream(e.stream)
        rows = slab[half * ar_every:(half + 1) * ar_every]
        if partial:                                     # barrier in mid-ring: reduce a copy, the rows get their
            rows = rows.clone()                         # own collective when the half completes
        works[half] = dist.all_reduce(rows, op=dist.ReduceOp.SUM, async_op=True)
        if not partial:
            flushed.append(half)
        other = works[1 - half]
        if other is not None:                           # the evaluators overwrite the other half next
            on_streams(other.wait)
            works[1 - half] = None

    def replay(i):
        evs[i % nstr].replay()
        if dist is not None and (i + 1) % per_flush == 0:
            flush(((i + 1) // per_flush - 1) % 2)

    def barrier(n_done):
        if dist is not None:
            if n_done % per_flush:
                flush((n_done // per_flush) % 2, partial=True)
            for w in works:
                if w is not None:
                    w.wait()
            dist.barrier()
        if have_gpu:
            torch.cuda.synchronize()

    nw, ns = warmup // E, steps // E
    for i in range(nw):
        replay(i)
    barrier(nw)
    t0 = time.perf_counter()
    for i in range(ns):
        replay(nw + i)
    if tail is not None:
        tail[0].replay()
        if dist is not None and tail[1] is not None:    # its rows get their own collective
            if tail[0].stream is not None:
                main.wait_stream(tail[0].stream)
            dist.all_reduce(tail[1], op=dist.ReduceOp.SUM)
    barrier(nw + ns)
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=slab.device)   # NCCL reduces device tensors only
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    run_steps.last_flushed_half = flushed[-1] if flushed else None
    return dt


def plan_steps(steps, warmup, evals_per_graph, nstr, ar_every=None):
    """(evaluations per graph launch E, steps run as whole launches, remaining steps, warm-up run).
    E is what was asked for (N>1: a divisor of the all-reduce period `ar_every`); any K is then timed exactly as K // E
    launches round-robin over the evaluators plus ONE more launch holding the K % E remaining evaluations, and the
    warm-up is rounded up to whole launches (the evaluator pipelines the evaluations of a launch: E = 1 would forgo
    that).  A short timed region takes one launch per evaluator (up to 8 evaluations each) instead of a few launches on
    some evaluators and none on others: K = 20 on four evaluators is 4 launches of 5, not 5 launches of 4."""
    per_replay = max(1, evals_per_graph)
    if steps < 4 * per_replay * nstr:
        per_replay = min(8, max(1, -(-steps // nstr)))
    if ar_every is not None:
        while per_replay > 1 and ar_every % per_replay:
            per_replay -= 1
    while per_replay > 1 and per_replay > steps:
        per_replay //= 2
    main_steps = steps // per_replay * per_replay
    return per_replay, main_steps, steps - main_steps, (warmup + per_replay - 1) // per_replay * per_replay


def make_evaluators(engine, net, x, y, S_global, nstr, graph=True, slab=None, per_replay=1, streams=None):
    if streams is None:
        streams = [torch.cuda.Stream() for _ in range(nstr)] if nstr > 1 else [None]
    ring = (lambda j: None) if slab is None else (lambda j: (slab.view(-1)[4 * j:], slab.shape[0], 4 * nstr))
    return [engine.GraphedElbo(net, x, y, S_global, capture=graph, counter_stride=nstr, stream=st, sums_ring=ring(j),
                               evals_per_replay=per_replay)
            for j, st in enumerate(streams)]


def kernel_alone_us(launch, stream, per_graph=20, reps=50):
    """Average duration of one launch of a kernel: `per_graph` back-to-back launches captured in
    a hipGraph, replayed `reps` times between two HIP events recorded on the launch stream."""
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(stream)
    with torch.cuda.stream(side):
        launch()
        with torch.cuda.graph(g, stream=side):
            for _ in range(per_graph):
                launch()
    stream.wait_stream(side)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        g.replay()
    e0.record(stream)
    for _ in range(reps):
        g.replay()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * per_graph)


def layer2_roofline(ev, net, dims, batch, S_local, lr, math_name):
    import bnn_hip
    from bnn_hip import ops, _lib as L
    hid_b = 4 if math_name == "f32" else 2
    if getattr(ev, "lr_pipe3", False):
        launch = ev.steady_state_stage()
        us = kernel_alone_us(launch, torch.cuda.current_stream())
        abytes = S_local * (algorithmic_bytes_layer(dims[0], dims[1], batch, hid_b, hid_b) +
                            algorithmic_bytes_layer(dims[1], dims[1], batch, hid_b, hid_b) +
                            algorithmic_bytes_layer(dims[1], dims[2], batch, hid_b, 4))
        achieved = abytes / (us * 1e-6) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "traffic_key": f"lr_S{S_local}_{math_name}_stage",
                "kernel": "K3e lr_stage_kernel: one pipeline stage = all three LR layers of one evaluation "
                          f"({dims[0]}x{dims[1]}, {dims[1]}x{dims[1]}, {dims[1]}x{dims[2]}; the finalize is a launch of its own)",
                "algorithmic_bytes_per_launch": abytes, "mc_samples_per_launch": S_local, "avg_launch_us": us,
                "note": "HIP events around back-to-back graph launches of this kernel, ALONE on its stream (incl. the "
                        "dependent-launch boundary), with the tile plan of the timed region"
                        + (f" (sized for 1/{ev.stride} of the chip because {ev.stride} evaluators run side by side there)"
                           if ev.stride > 1 else "")
                        + "; un-amortised 8 B/param/sample formula of SURVEY 8(d) summed over the three layers, bf16 activations"}
    if getattr(ev, "pipe3", False):
        # the timed region's launches are pipeline stages: one launch = one whole evaluation's work (first layer of
        # evaluation j+2, hidden layer of j+1, output layer + finalize of j)
        launch = ev.steady_state_stage()
        us = kernel_alone_us(launch, torch.cuda.current_stream())
        xb = ev.x.element_size()
        abytes = S_local * (algorithmic_bytes_layer(dims[0], dims[1], batch, xb, hid_b) +
                            algorithmic_bytes_layer(dims[1], dims[1], batch, hid_b, hid_b) +
                            algorithmic_bytes_layer(dims[1], dims[2], batch, hid_b, 4))
        achieved = abytes / (us * 1e-6) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "traffic_key": f"bbb_S{S_local}_{math_name}_stage",
                "kernel": "K1e bbb_fwd_final_next_kernel: one pipeline stage = all three layers of one "
                                           f"evaluation ({dims[0]}x{dims[1]}, {dims[1]}x{dims[1]}, {dims[1]}x{dims[2]} + finalize)",
                "algorithmic_bytes_per_launch": abytes, "mc_samples_per_launch": S_local, "avg_launch_us": us,
                "note": "HIP events around back-to-back graph launches of this kernel, ALONE on its stream (incl. the "
                        "dependent-launch boundary), with the tile plan of the timed region"
                        + (f" (sized for 1/{ev.stride} of the chip because {ev.stride} evaluators run side by side there: "
                           "alone it leaves most CUs idle)" if ev.stride > 1 else "")
                        + "; un-amortised 8 B/param/sample formula of SURVEY 8(d) summed over the three layers"}
    l2 = net.l2
    pd = tuple(t.detach() for t in (l2.weight_mu, l2.weight_rho, l2.bias_mu, l2.bias_rho))
    xin, ws, out = ev.bufs[0], ev.ws[1], ev.bufs[1]
    mm = bnn_hip.runtime.state.math

    def launch():
        if lr:
            ops.lr_linear_fwd(xin, *pd, n_samples=S_local, sigma_p=1.0, math_mode=mm, relu=True, y_dtype=out.dtype,
                              eps_mode=L.EPS_PHILOX, seed=1, layer_id=1, want_kl=True, workspace=ws, out=out,
                              x_sq=ev.bufs_sq[0], out_sq=ev.bufs_sq[1], w_frag=ev.wfrag[1], concurrency=ev.stride)
        else:
            ops.bbb_linear_fwd(xin, *pd, n_samples=S_local, prior=l2._prior_spec, math_mode=mm, relu=True,
                               y_dtype=out.dtype, eps_mode=L.EPS_PHILOX, seed=1, layer_id=1, want_stats=True,
                               workspace=ws, out=out, w_sigma=ev.wsigma[1], split_scratch=ev.split[1],
                               concurrency=ev.stride)
    us = kernel_alone_us(launch, torch.cuda.current_stream())
    abytes = S_local * algorithmic_bytes_layer(dims[1], dims[1], batch, hid_b, hid_b)
    note = "un-amortised 8 B/param/sample formula of SURVEY 8(d)"
    if lr and ev.wfrag[1] is not None:
        # prepared-operand form: the sample loop streams bf16 (M, sigma^2) = 4 B/param/sample plus bf16 x, x^2, y, y^2;
        # the fp32 (M, rho) pass and the KL sums are hoisted into bnn_lr_prepare, once per evaluation (not in this kernel)
        abytes = S_local * (dims[1] * dims[1] * 4 + batch * dims[1] * 4 + batch * dims[1] * 4)
        note = "prepared bf16 operands: 4 B/param/sample + bf16 x, x^2, y, y^2 (the 8 B/param fp32 pass runs once per evaluation in bnn_lr_prepare)"
    achieved = abytes / (us * 1e-6) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "kernel": ("lr_linear_fwd_kernel" if lr else "K1 bbb_linear_fwd") + f" layer 2 ({dims[1]}x{dims[1]})",
            "algorithmic_bytes_per_launch": abytes, "mc_samples_per_launch": S_local, "avg_launch_us": us,
            "note": "HIP events around back-to-back graph launches of this kernel, ALONE on its stream (incl. the "
                    "dependent-launch boundary), with the tile plan of the timed region"
                    + (f" (sized for 1/{ev.stride} of the chip because {ev.stride} evaluations run side by side there: "
                       "alone it leaves most CUs idle)" if ev.stride > 1 else "") + "; " + note}


def cpu_baseline(dims, lr, batch, budget_s=15.0):
    """The oracle (op-for-op CPU restatement of the reference path, parity-pinned by tests/golden)
    timed on this box's host cores: S=1 sample_elbo calls incl. the CPU eps draw, like the
    reference's own loop body."""
    from oracle import bnn_oracle as O
    from bnn_hip import synth
    ncpu = os.cpu_count() or 1
    sd = synth.synth_state_dict(dims[0], dims[1], dims[2], lr)
    p = O.NetParams.from_state_dict(sd, "classification", dims[0], lr, O.Prior.from_init([1.0], False))
    x, y = synth.synth_batch("classification", batch, dims[0], dims[2])
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    fn = O.sample_elbo_lr if lr else O.sample_elbo
    with torch.no_grad():
        # pick the intra-op thread count that is fastest for these op sizes on this host (all cores
        # is far from it on a many-core box: ~25 tiny elementwise ops per tensor), then time that.
        best_t, best = 1, float("inf")
        for nthr in [t for t in (1, 4, 8, 16, 32) if t <= ncpu]:
            torch.set_num_threads(nthr)
            fn(p, xt, yt, 0.5, 1)
            t0 = time.perf_counter()
            fn(p, xt, yt, 0.5, 1)
            fn(p, xt, yt, 0.5, 1)
            d = (time.perf_counter() - t0) / 2
            if d < best:
                best_t, best = nthr, d
        torch.set_num_threads(best_t)
        times = []
        t_end = time.perf_counter() + budget_s
        while time.perf_counter() < t_end and len(times) < 400:
            t0 = time.perf_counter()
            fn(p, xt, yt, 0.5, 1)
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": 1.0 / med, "unit": "MC-samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} sample_elbo(S=1) calls of the CPU oracle (fp32, no_grad, incl. eps draw) in "
                      f"{sum(times):.1f} s, median {med*1e3:.2f} ms, with the fastest of 1/4/8/16/32 intra-op threads "
                      f"(= `cores`); os.cpu_count()={ncpu}; {model}",
            "kl_elements_per_s": p.n_stochastic() / med}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--samples", type=int, default=1, help="MC samples per GPU per ELBO evaluation")
    ap.add_argument("--streams", type=int, default=4,
                    help="independent ELBO evaluations in flight per GPU (one hipGraph + HIP stream each)")
    ap.add_argument("--allreduce-every", type=int, default=64,
                    help="N>1: ELBO scalars of this many consecutive evaluations per evaluator share one all-reduce call "
                         "(the collective's stream shares a hardware queue with an evaluator: each call is a bubble on "
                         "that evaluator, 96.5k / 104.1k / 107.4k / 109.9k samples/s at 16 / 32 / 64 / 128 against "
                         "110.0k without the collective, one rank through the RCCL path)")
    ap.add_argument("--evals-per-graph", type=int, default=4,
                    help="consecutive ELBO evaluations captured in one hipGraph (amortises the ~10 us host cost of a "
                         "graph launch); reduced to a common divisor of --steps, --warmup and --allreduce-every")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--net", default="mnist", choices=list(DIMS))
    ap.add_argument("--variant", default="bbb", choices=["bbb", "lr"])
    ap.add_argument("--math", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the isolated launches of the dominant (layer-2) kernel: run under `rocprofv3 --kernel-trace "
                         "--stats` to get a per-kernel average that is directly comparable with roofline.avg_launch_us")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N>1 code path on a box with fewer GPUs than ranks (never a measurement):
    # BNN_BENCH_REHEARSAL=1 puts every rank on cuda:0 and exchanges over gloo instead of RCCL
    rehearsal = os.environ.get("BNN_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import bnn_hip
    from bnn_hip import engine
    bnn_hip.set_math(args.math)
    # HIP multiplexes a process's streams onto 4 hardware queues, bound at first use.  The evaluator
    # streams are created and used FIRST, before RCCL or any helper stream exists: then the four of
    # them (and the idle null stream) map onto distinct queues in the single-GPU and the multi-rank path
    # alike (16.3 / 16.9 us per one-sample evaluation); bound later, two evaluators end up sharing a
    # queue (24.7 us), and a fifth busy queue is slower again.
    pre_streams = [torch.cuda.Stream() for _ in range(max(1, args.streams))]
    for st in pre_streams:
        with torch.cuda.stream(st):
            torch.zeros(16, device=dev).add_(1.0)
    torch.cuda.synchronize()
    dist = None
    # BNN_BENCH_FORCE_DIST=1: take the N>1 code path (process group, slab all-reduces) with one rank
    if world > 1 or os.environ.get("BNN_BENCH_FORCE_DIST", "0") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        bnn_hip.shard_samples(True)

    dims, lr = DIMS[args.net], args.variant == "lr"
    # the wide stack has no task attached in BASELINE: Gaussian NLL over its 4096 outputs
    net, x, y = build_net(dims, lr, args.batch, dev, "regression" if args.net == "wide" else "classification")
    S_local, S_global = args.samples, args.samples * world
    nstr = max(1, args.streams)
    ar_every = max(1, args.allreduce_every)
    per_replay, main_steps, tail_steps, warmup_run = plan_steps(args.steps, args.warmup, args.evals_per_graph, nstr,
                                                                ar_every if dist is not None else None)
    slab = torch.zeros((2 * ar_every, nstr, 4), dtype=torch.float32, device=dev) if dist is not None else None
    evs = make_evaluators(engine, net, x, y, S_global, nstr, graph=not args.no_graph, slab=slab, per_replay=per_replay,
                          streams=pre_streams[:nstr] if nstr > 1 else None)
    assert evs[0].n_local == S_local
    tail = None
    if tail_steps:
        tslab = torch.zeros((tail_steps, 1, 4), dtype=torch.float32, device=dev) if dist is not None else None
        tev = make_evaluators(engine, net, x, y, S_global, 1, graph=not args.no_graph, slab=tslab, per_replay=tail_steps,
                              streams=[pre_streams[0]] if nstr > 1 else None)[0]
        tail = (tev, tslab)
    if args.roofline_only:
        torch.cuda.synchronize()
        roof = layer2_roofline(evs[0], net, dims, args.batch, S_local, lr, args.math)
        print(json.dumps({"roofline": roof, "mc_samples_per_launch": S_local, "variant": args.variant}), flush=True)
        return
    # clocks, caches and the allocator settle during the first few dozen replays after capture: always run
    # some untimed ones before the W warm-up steps the caller asked for (they matter when W is tiny)
    prewarm = 16 * per_replay * nstr
    # (N>1: a whole number of ring laps, so that the device-side ring cursors are back at row 0 when the measured
    # call starts counting its flushes from 0)
    lap = 2 * ar_every * nstr
    run_steps(evs, 0, prewarm if dist is None else (prewarm + lap - 1) // lap * lap, dist, slab, ar_every)
    dt = run_steps(evs, main_steps, warmup_run, dist, slab, ar_every, tail=tail)
    if dist is not None and run_steps.last_flushed_half is not None:
        # every all-reduced row carries the GLOBAL sample count in its 4th word
        h = run_steps.last_flushed_half
        got = slab[h * ar_every:(h + 1) * ar_every, :, 3]
        assert bool((got == float(S_global)).all()), f"all-reduced sample counts {got.flatten().tolist()} != {S_global}"
    value = S_global * args.steps / dt
    nst = n_stochastic(dims)
    layers = "-".join(map(str, (dims[0], dims[1], dims[1], dims[2])))

    out = {
        "metric": "MC-forward-samples/sec + KL-elements/sec, 784-1200-1200-10 BNN" if args.net == "mnist"
        else f"MC-forward-samples/sec + KL-elements/sec, {layers} BNN",
        "value": value, "unit": "MC-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "warmup_executed": warmup_run,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.math == "bf16" else "f32", "data": "synthetic",
        **({"rehearsal": "all ranks on cuda:0 over gloo; NOT a measurement"} if rehearsal else {}),
        "config": {"workload": f"{layers} {'LR' if lr else 'BBB'} forward-only ELBO evaluation (3-layer forward + "
                               f"log p/log q reductions over {nst} stochastic params + NLL per MC sample), batch "
                               f"{args.batch}, {S_local} MC sample(s) per GPU per evaluation, {nstr} evaluation(s) in "
                               f"flight per GPU, Gaussian prior, on-chip Philox eps",
                   "batch": args.batch, "mc_samples_per_gpu_per_step": S_local, "mc_samples_per_step": S_global,
                   "stochastic_params": nst, "hipgraph": not args.no_graph, "evaluations_in_flight": nstr,
                   "evaluations_per_graph_launch": per_replay,
                   "pipelined_evaluations": bool(getattr(evs[0], "pipelined", False)),   # output layer + finalize of one
                   # evaluation share a launch with the next ones' hidden and first layers (same per-evaluation results)
                   "parallelism": (f"mc-sample-shard x{world}; RCCL sum all-reduce of the 4 ELBO scalars of every evaluation, "
                                   f"{args.allreduce_every * nstr} evaluations per call, asynchronous") if world > 1 else "single GPU"},
        "kl_elements_per_s": value * nst,
    }

    if rank == 0:
        roof = layer2_roofline(evs[0], net, dims, args.batch, S_local, lr, args.math)
        tj = os.path.join(REPO, "profiles", "traffic.json")      # measured by tools/collect_traffic.py (rocprofv3 --pmc)
        if os.path.exists(tj):
            try:
                t = json.load(open(tj))
                key = roof.pop("traffic_key", f"{args.variant}_S{S_local}_{args.math}")
                if key in t:
                    roof["traffic"] = t[key]["hbm_bytes_per_launch"]
                    roof["traffic_source"] = t[key].get("source", "profiles/traffic.json")
            except Exception:
                pass
        if roof.get("mc_samples_per_launch"):
            # context, not the roofline figure: the same algorithmic bytes at the rate of the whole timed region (all
            # evaluators side by side), where the kernel above shares the chip with its peers
            evals_per_s = value / S_global
            roof["timed_region_aggregate_GBps"] = roof["algorithmic_bytes_per_launch"] * evals_per_s / 1e9 \
                if "one pipeline stage" in roof.get("kernel", "") else None
        out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dims, lr, args.batch)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if world == 1 and nstr > 1 and not args.no_extras:
            e1 = make_evaluators(engine, net, x, y, S_global, 1)
            d1 = run_steps(e1, 600, 60, None)
            out["single_evaluation_in_flight"] = {"samples_per_s": S_global * 600 / d1, "us_per_evaluation": d1 * 1e6 / 600,
                                                  "note": "same workload, one hipGraph replayed back to back on one stream "
                                                          "(latency of one ELBO evaluation)"}
            del e1
        if world == 1 and not args.no_extras and args.net == "mnist":
            extras = []
            for (S, ns, steps) in ((8, 3, 300), (64, 1, 100), (256, 1, 40)):
                e2 = make_evaluators(engine, net, x, y, S, ns, streams=pre_streams[:ns] if ns > 1 else None)
                d2 = run_steps(e2, steps, max(5, steps // 10), None)
                r2 = layer2_roofline(e2[0], net, dims, args.batch, S, lr, args.math)
                extras.append({"mc_samples_per_evaluation": S, "evaluations_in_flight": ns, "steps": steps,
                               "samples_per_s": S * steps / d2, "kl_elements_per_s": S * steps / d2 * nst,
                               "us_per_evaluation": d2 * 1e6 / steps, "layer2_us_per_launch": r2["avg_launch_us"],
                               "layer2_hbm_frac": r2["frac"]})
                del e2
            out["extras"] = extras
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
