#!/usr/bin/env python3
"""bench.py — MC-forward-samples/sec (+ KL-elements/sec) of the Bayes-by-backprop hot path.

Workload (BASELINE.json configs[1], SURVEY §8(d) "C2"): the 784-1200-1200-10 BBB network, minibatches of 128 rows,
1 MC sample per minibatch per GPU, bf16 MFMA operands / fp32 statistics, synthetic inputs (mu~U(-0.2,0.2),
rho~U(-5,-4), x~U(0,1), labels~U{0..9}; numpy RandomState seeds 1234/5678+m), Gaussian prior sigma_p=1, on-chip
Philox epsilon.

One STEP = one LAUNCH GROUP: `--group` (256) independent minibatches resident in HBM, each evaluated forward-only --
the full 3-layer forward with freshly sampled weights, the sampled log p(w) / log q(w) reductions over all 2 395 210
stochastic parameters and the NLL (reference networks.py:199-203), `--samples` MC samples per GPU -- through ONE launch
per layer together (`BayesianNetwork.elbo_many`, the product API; one hipGraph replay), every (minibatch, MC sample)
pair with its own Philox subsequence.  The minibatches are a stream of INDEPENDENT evaluations sharing the parameters
(what class_task.py:89-103 walks one at a time).  `--steps K --warmup W` = K timed + W untimed replays, so the driver's
`--steps 20` times 20 launch groups (5120 minibatch evaluations, milliseconds), not one.  `value` counts MC forward
samples: group x samples x GPUs per step.  `single_evaluation_in_flight` in the output is the other regime: one
minibatch at a time, each evaluation waiting for the previous one (the training loop's dependency,
class_task.py:73-79).

Cold start: a device coming out of idle runs its first ~30 ms of launch groups up to 12 % slower than it does from then on
(tools/warmup_probe.py: consecutive windows of 20 steps take 684, 634, 617, 614, 611 ... us per step), and the driver's `--warmup 5
--steps 20` last 17 ms.  The W + K steps are therefore run twice: straight after the evaluator is built (reported as `cold_start`),
and again -- W untimed, K timed -- after `--settle-ms` (250 ms) of the same replays: that second measurement is `value`.

Multi-GPU (`--gpus N`, one process per GPU; started by the driver through torch.distributed.run, or by this script
itself when WORLD_SIZE is unset): weak scaling.  Every minibatch is evaluated with N x `--samples` MC samples, rank r
owning samples [r * samples, (r + 1) * samples) of every minibatch (Philox subsequence = global sample index, so the
result does not depend on N), and the only collective is ONE RCCL sum all-reduce of the group's [group, 4] ELBO
scalars per launch group.  `c4` in the output is BASELINE configs[3] as SURVEY §8(e) words it: one minibatch, 64 (and
512) MC samples split over the ranks, one all-reduce of the 4-vector PER EVALUATION.

Prints ONE JSON line (rank 0).  value = total MC samples / s over all ranks; `roofline` is the dominant kernel (layer
2, 1200x1200 weights) against the roof that binds it -- vector issue for K1b2, whose on-chip generator is most of the
~160 VALU instructions per 8 sampled weights (instruction mix: profiles/isa_mix.json; counters: profiles/pmc.json), with the
ALGORITHMIC lane-op count of the epsilon map beside the compiled loop's (`valu.algorithmic`: derived from the map in
include/bnn_hip.h, not from the kernel), SURVEY §8(d)'s HBM figure in algorithmic bytes (`hbm_algorithmic`), the bf16 MFMA
fraction (`mfma_frac_of_bf16_peak`) and the fabric traffic counters measured (`traffic`); `cpu_baseline` is the
parity-pinned CPU oracle timed on this host (fastest thread count, all cores, one thread); `extras` are the other BASELINE
configs and the other math modes of the headline workload (`math_modes`: f32 = the exact-fp32 matrix core, bf16x3 = split
bf16 operands on the bf16 matrix core -- the two modes that meet ELBO rtol 1e-4 against the reference's arithmetic at
every beta).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "bayesian-neural-network_amd"))
sys.path.insert(0, REPO)

DIMS = {"mnist": (784, 1200, 10), "wide": (4096, 4096, 4096), "reg": (1, 50, 1)}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3     # v_mfma_f32_16x16x4_f32: the fp32 vector rate (MI355X_MICROARCH.md)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed steps; ONE STEP = one launch group = `--group` independent minibatches")
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps (launch groups)")
    ap.add_argument("--samples", type=int, default=1, help="MC samples per GPU per minibatch evaluation")
    ap.add_argument("--group", type=int, default=256,
                    help="independent minibatches resident in HBM that share one launch per layer (1 = one minibatch "
                         "at a time, each evaluation waiting for the previous one)")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--net", default="mnist", choices=list(DIMS))
    ap.add_argument("--variant", default="bbb", choices=["bbb", "lr"])
    ap.add_argument("--math", default="bf16", choices=["bf16", "f32", "bf16x3"])
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="untimed replays of the launch group ahead of the W warm-up steps, until the device has reached its steady "
                         "state under this load (the first ~40 launch groups after idle run up to 12 %% slower: tools/warmup_probe.py); "
                         "0 = time the K steps straight after the W warm-up steps of a cold device (reported as `cold_start` otherwise)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the isolated launches of the dominant (layer-2) kernel: run under `rocprofv3 --kernel-trace "
                         "--stats` to get a per-kernel average that is directly comparable with roofline.avg_launch_us")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher (no GPU touched)
def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent never initialises the
    GPU and never execs), relay rank 0's output and return the worst exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ------------------------------------------------------------------------------------------ bookkeeping (CPU-testable)
def plan_groups(steps: int, warmup: int, group: int):
    """(G, full replays, remainder, warm-up replays): K steps = `full` replays of a G-minibatch launch group plus one
    launch of `remainder` minibatches; the warm-up is rounded UP to whole replays of the G-group."""
    g = max(1, min(int(group), int(steps)))
    full, rem = divmod(int(steps), g)
    warm = (int(warmup) + g - 1) // g
    return g, full, rem, warm


def algorithmic_bytes_layer(fin, fout, batch, x_bytes, y_bytes):
    """(mu, rho) fp32 read once per sample (8 B/param incl. bias, eps on chip,
    w never stored) + x read + y written."""
    return 8 * (fin * fout + fout) + batch * fin * x_bytes + batch * fout * y_bytes


def sampling_bytes(fin, fout, n_samples):
    """K1s (bnn_bbb_sample_weights): a block serves a GROUP of four samples from one read of (mu, rho) -- 8 B per weight
    and group -- and writes each sample's bf16 weights -- 2 B per weight and sample (the biases likewise: fp32 out)."""
    groups = (n_samples + 3) // 4
    return (8 * groups + 2 * n_samples) * fin * fout + (8 * groups + 4 * n_samples) * fout


def n_stochastic(dims):
    return sum(i * o + o for i, o in ((dims[0], dims[1]), (dims[1], dims[1]), (dims[1], dims[2])))


# the source files a kernel family is built from: entries of profiles/traffic.json, pmc.json and isa_mix.json carry the
# hash of their family's files and are ignored (reported as null) once those have changed
# (../../include/bnn_hip.h: BNN_PHILOX_ROUNDS, the largest driver of the generator's instruction mix, lives there)
KERNEL_SOURCES = {
    "bbb": ("bbb_linear.hip", "bnn_device.h", "bbb_sample_body.h", "bnn_fin.h", "../../include/bnn_hip.h"),
    "lr": ("lr_linear.hip", "bnn_device.h", "bnn_fin.h", "../../include/bnn_hip.h"),
    "block_gemm": ("bbb_block_gemm.h", "bnn_device.h", "../../include/bnn_hip.h"),
}


def board_power_while(replay, seconds=1.5):
    """rocm-smi readings (package power, its limit, shader clock) of GPU 0 while `replay()` keeps the device busy: the headline
    kernel runs at the board's power limit (profiles/r04_power_probe.log), which bounds it before any pipe does.
    Outside the timed region; None when rocm-smi is absent or refuses."""
    import threading
    import torch

    def smi():
        try:
            o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showmaxpower", "--showclocks", "--json"], capture_output=True,
                               text=True, timeout=10).stdout
            c = json.loads(o).get("card0", {})
            f = lambda key: next((float(str(v).strip("()MmHhZz ")) for k, v in c.items() if key in k.lower()), None)
            return {"watts": f("current socket graphics package power") or f("average graphics package power"),
                    "limit_watts": f("max graphics package power"), "sclk_mhz": f("sclk clock speed")}
        except Exception:
            return None
    stop, got = [False], []

    def poll():
        time.sleep(0.4)                                   # let the load settle
        while not stop[0]:
            r = smi()
            if r and r.get("watts"):
                got.append(r)
            time.sleep(0.1)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    while time.time() - t0 < seconds or (not got and time.time() - t0 < 2 * seconds):
        replay()
        torch.cuda.synchronize()
    stop[0] = True
    th.join()
    if not got:
        return None
    w = [g["watts"] for g in got]
    lim = got[0].get("limit_watts")
    return {"watts_mean": sum(w) / len(w), "watts_max": max(w), "limit_watts": lim, "frac_of_limit": (sum(w) / len(w) / lim) if lim else None,
            "sclk_mhz": got[-1].get("sclk_mhz"), "readings": len(w),
            "source": "rocm-smi --showpower --showmaxpower --showclocks while the launch group replays back to back (outside the timed region)"}


def _profile_file(name):
    """profiles/<name> -- or the copy a profiling run is writing (tools/profile_round.sh exports BNN_PROFILES_DIR), so that the
    bench line of that run carries the counters collected just before it."""
    d = os.environ.get("BNN_PROFILES_DIR")
    if d and os.path.exists(os.path.join(d, name)):
        return os.path.join(d, name)
    return os.path.join(REPO, "profiles", name)


def source_hash(files=None) -> str:
    """Hash of kernel sources (all of csrc/ by default, or the named files of one kernel family)."""
    h = hashlib.sha256()
    d = os.path.join(REPO, "bayesian-neural-network_amd", "csrc")
    for f in sorted(files if files is not None else [f for f in os.listdir(d) if f.endswith((".hip", ".h"))]):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def run_groups(main_ev, full, warm, tail_ev, dist=None, collective_every_eval=False):
    """`warm` untimed + `full` timed replays of main_ev (+ one of tail_ev), bracketed by barrier + synchronize.
    Returns seconds (max over ranks).  N > 1: after every replay the launch group's [G, 4] sums are copied into one of
    two slab slots and sum-all-reduced asynchronously (ONE collective per launch group) while the next group runs;
    `collective_every_eval`: the all-reduce is awaited on the evaluation's stream before the next one starts (C4)."""
    import torch
    have_gpu = torch.cuda.is_available()
    slots, works = None, [None, None]
    if dist is not None:
        slots = [torch.zeros_like(main_ev.sums) for _ in range(2)]
    count = [0]

    def one(ev):
        out = ev.replay()
        if dist is None:
            return
        k = count[0] & 1
        count[0] += 1
        if collective_every_eval:
            dist.all_reduce(out, op=dist.ReduceOp.SUM)          # in place; the next evaluation waits for it
            return
        if out.shape != slots[k].shape:                         # the remainder group
            dist.all_reduce(out, op=dist.ReduceOp.SUM)
            return
        if works[k] is not None:
            works[k].wait()
        slots[k].copy_(out)
        works[k] = dist.all_reduce(slots[k], op=dist.ReduceOp.SUM, async_op=True)

    def barrier():
        for w in works:
            if w is not None:
                w.wait()
        if dist is not None:
            dist.barrier()
        if have_gpu:
            torch.cuda.synchronize()

    for _ in range(warm):
        one(main_ev)
    barrier()
    t0 = time.perf_counter()
    for _ in range(full):
        one(main_ev)
    if tail_ev is not None:
        one(tail_ev)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=main_ev.sums.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    run_groups.last_reduced = slots
    return dt


# ------------------------------------------------------------------------------------------ GPU side
def build_net(dims, lr, batch, device, mode="classification", n_minibatches=1):
    import torch
    import networks
    from bnn_hip import synth
    mp = dict(input_shape=dims[0], classes=dims[2], batch_size=batch, hidden_units=dims[1], mode=mode,
              mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=lr)
    net = networks.BayesianNetwork(mp)
    sd = synth.synth_state_dict(dims[0], dims[1], dims[2], lr)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    xs, ys = zip(*[synth.synth_batch(mode, batch, dims[0], dims[2], seed=5678 + m) for m in range(n_minibatches)])
    import numpy as np
    x = torch.from_numpy(np.stack(xs)).to(device)
    y = torch.from_numpy(np.stack(ys)).to(device)
    return net.to(device).train(), x, y


def make_evaluator(engine, net, x, y, S_global, G, graph=True, per_replay=1):
    """Evaluator of the first G resident minibatches (G = 1: one minibatch, unstacked)."""
    if G == 1:
        return engine.GraphedElbo(net, x[0], y[0], S_global, capture=graph, evals_per_replay=per_replay)
    return engine.GraphedElbo(net, x[:G], y[:G], S_global, capture=graph, stacked=True, evals_per_replay=per_replay)


def kernel_alone_us(launch, stream, per_graph=20, reps=30):
    """Average duration of one launch of a kernel: `per_graph` back-to-back launches captured in
    a hipGraph, replayed `reps` times between two HIP events recorded on the launch stream."""
    import torch
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


def layer2_roofline(ev, net, dims, batch, lr, math_name):
    """The dominant kernel of an evaluator's launch group -- layer 2 (dims[1] x dims[1] weights) over all of its
    (minibatch, MC sample) pairs -- timed alone with HIP events, against the HBM roof (and the bf16 MFMA roof)."""
    import torch
    import bnn_hip
    from bnn_hip import ops, _lib as L
    n = ev.n_local
    hid_b = 2 if math_name == "bf16" else 4           # (bf16x3: a bf16 activation is a pair of planes)
    l2 = net.l2
    pd = tuple(t.detach() for t in (l2.weight_mu, l2.weight_rho, l2.bias_mu, l2.bias_rho))
    xin, ws, out = ev.bufs[0], ev.ws[1], ev.bufs[1]
    mm = bnn_hip.runtime.state.math
    if getattr(ev, "lib", [False] * 3)[1]:
        # large batch: the layer is a sampling launch (HBM: 8 B read + 2 B written per weight and sample) followed by the
        # 256 x 256 block form of the matmul over the sampled weights (K1g, matrix cores): the matmul is the dominant kernel
        wl, bl = ev.lib_w[1], ev.lib_b[1]
        smp = lambda: ops.bbb_sample_weights([dict(w_mu=pd[0], w_rho=pd[1], b_mu=pd[2], b_rho=pd[3], prior=l2._prior_spec, layer_id=1,
                                                    workspace=ws, w_out=wl, b_out=bl)], n_samples=n, seed=1)
        gemm = lambda: ops.bbb_sampled_matmul(xin, wl, bl, n_samples=n, relu=True, y_dtype=out.dtype, out=out)
        us_s = kernel_alone_us(smp, torch.cuda.current_stream(), per_graph=10, reps=10)
        us_g = kernel_alone_us(gemm, torch.cuda.current_stream(), per_graph=4, reps=10)
        flops = 2.0 * n * batch * dims[1] * dims[1]
        sbytes = sampling_bytes(dims[1], dims[1], n)
        tf = flops / (us_g * 1e-6) / 1e12
        return {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS,
                "traffic": None, "traffic_key": f"block_gemm_{dims[1]}_n{n}_b{batch}_{math_name}",
                "kernel": f"bnn::bbb_block_gemm_kernel (K1g: 256x256 block tile, v_mfma_f32_16x16x32_bf16, LDS-DMA operands, "
                                           f"bias/ReLU/bf16 epilogue) over pre-sampled weights, layer 2 ({dims[1]}x{dims[1]}), batch {batch}",
                "algorithmic_flops_per_launch": flops, "mc_samples_per_launch": n, "avg_launch_us": us_g,
                "sampling": {"kernel": "K1s bbb_sample_kernel", "avg_launch_us": us_s, "algorithmic_bytes_per_launch": sbytes,
                             "hbm_GBps": sbytes / (us_s * 1e-6) / 1e9, "hbm_frac": sbytes / (us_s * 1e-6) / 1e9 / HBM_PEAK_GBS},
                "note": "HIP events around back-to-back graph launches of the one matmul launch (bias, ReLU and the bf16 conversion are its "
                        "epilogue); 2 * batch * weights flops per sample against the dense bf16 MFMA peak"}
    common = dict(n_samples=n, math_mode=mm, relu=True, y_dtype=out.dtype, eps_mode=L.EPS_PHILOX, seed=1, layer_id=1,
                  workspace=ws, out=out)
    if lr:
        kw = dict(sigma_p=1.0, want_kl=True, x_sq=ev.bufs_sq[0], out_sq=ev.bufs_sq[1], w_frag=ev.wfrag[1],
                  split_scratch=getattr(ev, "lr_split", [None] * 3)[1], **common)
        if getattr(ev, "lr_x3", False):                # split-bf16 math: the plane pair in, fp32 out (the layer below the output layer)
            kw.update(x_lo=ev.bufs_lo[0], out_lo=ev.bufs_lo[1])
        else:
            kw["math_mode"] = ev.math
        plan = ops.lr_plan(xin, *pd, **kw)
        launch = lambda: ops.lr_linear_fwd(xin, *pd, **kw)
    else:
        kw = dict(prior=l2._prior_spec, want_stats=True, w_sigma=ev.wsigma[1], split_scratch=ev.split[1], **common)
        if math_name == "bf16x3":
            kw.update(x_lo=ev.bufs_lo[0], out_lo=ev.bufs_lo[1])
        plan = ops.bbb_plan(xin, *pd, **kw)
        launch = lambda: ops.bbb_linear_fwd(xin, *pd, **kw)
    us = kernel_alone_us(launch, torch.cuda.current_stream())
    abytes = n * algorithmic_bytes_layer(dims[1], dims[1], batch, hid_b, hid_b)
    note = "un-amortised 8 B/param/sample formula of SURVEY 8(d): every (minibatch, sample) pair is charged a full read of (mu, rho)"
    if lr and ev.wfrag[1] is not None:
        # prepared-operand form: the sample loop streams bf16 (M, sigma^2) = 4 B/param/sample plus bf16 x, x^2, y, y^2;
        # the fp32 (M, rho) pass and the KL sums are hoisted into bnn_lr_prepare, once per evaluation (not in this kernel)
        abytes = n * (dims[1] * dims[1] * 4 + batch * dims[1] * 4 + batch * dims[1] * 4)
        note = "prepared bf16 operands: 4 B/param/sample + bf16 x, x^2, y, y^2 (the 8 B/param fp32 pass runs once per evaluation in bnn_lr_prepare)"
    achieved = abytes / (us * 1e-6) / 1e9
    flops = n * (4 if lr else 2) * batch * dims[1] * dims[1]
    form = {1: "tile", 2: "gemm", 3: "gemm_kslice"}[plan["form"]]
    kname = {("bbb", "tile"): "K1a bbb_fwd_kernel",
             ("bbb", "gemm"): ("K1b2 bbb_fwd_gemm2_kernel<X3> (split-bf16 operands: three bf16 MFMAs per product; parameters and the x plane pair through LDS, 2 pairs per block)"
                               if math_name == "bf16x3" else "K1b2 bbb_fwd_gemm2_kernel (parameters and x through LDS, 2 pairs per block)") if plan["waves"] == 8 else "K1b bbb_fwd_gemm_kernel",
             ("bbb", "gemm_kslice"): "K1b bbb_fwd_gemm_kernel, K-sliced with the fused last-arriver reduce",
             ("lr", "tile"): "K3a lr_fwd_kernel",
             ("lr", "gemm"): "K3b lr_fwd_gemm_kernel<X3> (mean product in split-bf16: three MFMAs; variance product one)" if math_name == "bf16x3" else "K3b lr_fwd_gemm_kernel",
             ("lr", "gemm_kslice"): "K3s lr_fwd_kslice_kernel (32-feature groups x K slices, whole parameter lines, slices meet through a scratch)"}[("lr" if lr else "bbb", form)]
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "traffic_key": f"{'lr' if lr else 'bbb'}_{dims[1]}_n{n}_b{batch}_{math_name}",
            "kernel": f"{kname}, layer 2 ({dims[1]}x{dims[1]})", "plan": plan,
            "algorithmic_bytes_per_launch": abytes, "mc_samples_per_launch": n, "avg_launch_us": us,
            "mfma_tflops": flops / (us * 1e-6) / 1e12, "mfma_frac_of_bf16_peak": flops / (us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "note": "HIP events around back-to-back graph launches of this kernel alone on its stream (incl. the "
                    "dependent-launch boundary); " + note}
    roof["math"] = math_name
    if math_name == "f32":
        # the exact-fp32 matrix core (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate): 2 * batch flops per sampled weight
        tf = flops / (us * 1e-6) / 1e12
        roof["hbm_algorithmic"] = {k: roof[k] for k in ("achieved", "peak", "unit", "frac")}
        roof.update({"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS})
    elif math_name == "bf16x3":
        issued = (2 if lr else 3) * flops                           # BBB: three bf16 MFMAs per product; LR: 3 (mean) + 1 (variance) of 2
        roof["mfma_flops_issued_per_launch"] = issued
        roof["mfma_issued_frac_of_bf16_peak"] = issued / (us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS
    if not lr and form in ("gemm", "gemm_kslice") and math_name in ("bf16", "bf16x3"):
        roof = valu_bound(roof, dims[1], dims[1], n, batch, us, ev.wsigma[1] is not None)
    elif lr and form == "gemm" and math_name in ("bf16", "bf16x3"):
        # K3b: two bf16 GEMMs (mean, variance) over parameters the launch's pairs share -- the matrix cores are its busiest
        # unit (profiles/pmc.json: MFMA 0.36, VALU 0.28 busy), not memory: price it against the bf16 MFMA figure, with the
        # SURVEY 8(d) HBM figure beside it
        hbm = {k: roof[k] for k in ("achieved", "peak", "unit", "frac")}
        hbm["note"] = "SURVEY 8(d) algorithmic bytes over the launch time: an accounting convention, not the binding resource"
        roof["hbm_algorithmic"] = hbm
        roof.update({"bound": "mfma", "achieved": roof["mfma_tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": roof["mfma_frac_of_bf16_peak"]})
    return roof


VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12      # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-ops/s
ISSUE_CYCLES = {"valu": 2, "trans": 8, "imul": 7}      # per-class issue cost per wave-instruction and SIMD (tools/ubench.hip)


def algorithmic_valu_per_normal(philox_rounds=7, sigma_hoisted=True, x3=False):
    """Vector work per SAMPLED WEIGHT that the epsilon map of include/bnn_hip.h and the layer's arithmetic REQUIRE,
    counted from the map -- not from the compiled loop -- as wave-instructions per lane-element by class:
      Philox4x32-R, 4 normals per call: per round 2 32x32->64 multiplies and 4 xors (the key schedule is wave-uniform:
        scalar unit); round 1's product of the constant counter word is wave-uniform too -> 2R - 1 multiplies, 4R xors;
      Box-Muller, 2 normals per pair of words: 2 int->float conversions + 2 fma (u in (0,1]), log2, one multiply
        (-2 ln 2), sqrt, sin, cos, 2 multiplies = 7 plain + 4 transcendental;
      the layer: w = fma(sigma, eps, mu), sum eps^2, sum w^2 (3 fma, two weights per v_pk_fma_f32 = 1.5 instructions),
        half a packed bf16 conversion; softplus when sigma
        is not hoisted (+ 2 transcendental + 5 plain); the split-bf16 mode's low part (+ 1 subtract, 1 unpack, half a
        conversion).
    Returns ({class: instructions per weight}, issue cycles per weight at ISSUE_CYCLES)."""
    per4 = {"imul": 2 * philox_rounds - 1, "valu": 4 * philox_rounds + 2 * 7, "trans": 2 * 4}
    c = {k: v / 4.0 for k, v in per4.items()}
    c["valu"] += 2.0
    if not sigma_hoisted:
        c["valu"] += 5
        c["trans"] += 2
    if x3:
        c["valu"] += 2.5
    return c, sum(ISSUE_CYCLES[k] * v for k, v in c.items())


def valu_bound(roof, fin, fout, n, batch, us, sig):
    """K1b is bound by vector issue, not by memory (profiles/pmc.json: VALU busy 0.79, MFMA 0.09): the on-chip generator
    costs ~80 VALU wave-instructions per 8 sampled weights.  Price the launch against the VECTOR-ISSUE roof from the
    kernel's own instruction mix (profiles/isa_mix.json, tools/make_isa_mix.py): issue cycles of one k-step of one wave
    x the wave-steps of the launch, against 1024 SIMDs x 2.4 GHz; the SURVEY 8(d) HBM figure stays beside it."""
    key = "bbb_fwd_gemm2_kernel<4,2,philox>" if roof["plan"]["waves"] == 8 else f"bbb_fwd_gemm_kernel<4,{'true' if sig else 'false'},philox>"
    if roof.get("math") == "bf16x3":
        key = "bbb_fwd_gemm2_kernel<4,2,philox,x3>"
    hbm = {k: roof[k] for k in ("achieved", "peak", "unit", "frac")}
    hbm["note"] = "SURVEY 8(d) algorithmic bytes (un-amortised: 8 B/param per (minibatch, sample) pair) over the launch time: an accounting " \
                  "convention, not the binding resource -- the pairs of a launch share (mu, sigma) through L2 (see traffic)"
    try:
        e = json.load(open(os.path.join(REPO, "profiles", "isa_mix.json")))[key]
        fresh = e["source_hash"] == source_hash(KERNEL_SOURCES["bbb"])
    except Exception:
        e, fresh = None, False
    roof["hbm_algorithmic"] = hbm
    if e is None or not fresh:
        roof["bound"] = "valu"
        roof.update({"achieved": None, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "frac": None,
                     "valu_note": "profiles/isa_mix.json has no entry for the current kernel sources (run tools/make_isa_mix.py)"})
        return roof
    wave_steps = ((fout + 15) // 16) * n * ((batch + 127) // 128) * ((fin + 31) // 32)
    cycles = wave_steps * e["valu_issue_cycles_per_iteration"]          # SIMD issue cycles of the launch's vector instructions
    lane_ops = cycles * 32.0
    achieved = lane_ops / (us * 1e-6) / 1e12
    roof.update({"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "frac": achieved / VALU_PEAK_TLANEOPS,
                 "valu": {"issue_cycles_per_kstep_per_wave": e["valu_issue_cycles_per_iteration"], "classes": e["classes"],
                          "cycle_costs": e["cycle_costs"], "wave_ksteps_per_launch": wave_steps,
                          "floor_us_at_2p4GHz": cycles / (1024 * 2.4e9) * 1e6,
                          "source": "profiles/isa_mix.json (static instruction mix of the k-step loop, tools/make_isa_mix.py; issue costs from "
                                    "tools/ubench.hip); frac = that issue time at 1024 SIMDs x 2.4 GHz over the measured launch time"}})
    # the same roof from the ALGORITHMIC count (the map's arithmetic per weight, independent of how the loop was compiled: a
    # fatter loop scores higher on the instruction-mix figure above, not on this one)
    try:
        import re as _re
        hdr = open(os.path.join(REPO, "include", "bnn_hip.h")).read()
        rounds = int(_re.search(r"#\s*define\s+BNN_PHILOX_ROUNDS\s+(\d+)", hdr).group(1))
    except Exception:
        rounds = 7
    cls_w, cyc_w = algorithmic_valu_per_normal(rounds, sig, roof.get("math") == "bf16x3")
    weights = float(fin) * fout * n * ((batch + 127) // 128)          # sampled once per (pair, batch block)
    alg_floor_us = weights / 64.0 * cyc_w / (1024 * 2.4e9) * 1e6
    roof["valu"]["algorithmic"] = {"wave_instructions_per_weight": cls_w, "issue_cycles_per_weight": cyc_w, "philox_rounds": rounds,
                                   "issue_cycles_per_kstep_per_wave": cyc_w * 8, "floor_us_at_2p4GHz": alg_floor_us, "frac": alg_floor_us / us,
                                   "source": "bench.algorithmic_valu_per_normal: the epsilon map's arithmetic per sampled weight "
                                             "(Philox rounds, Box-Muller, w / statistics FMAs, bf16 pack), not the compiled loop"}
    try:       # the measured floor: the same kernel with everything but its vector work compiled out (tools/make_valu_floor.py)
        vf = json.load(open(_profile_file("valu_floor.json")))
        if roof["plan"]["waves"] == 8 and vf.get("source_hash") == source_hash(KERNEL_SOURCES["bbb"]):
            roof["valu"]["in_situ"] = {k: vf[k] for k in ("full_launch_us", "vector_work_only_us", "without_dma_us", "frac", "source")}
    except Exception:
        pass
    try:
        pm = json.load(open(_profile_file("pmc.json")))
        for k, v in pm.items():
            if "bbb_fwd_gemm" in k and k.startswith("bbb_g256_x3:" if roof.get("math") == "bf16x3" else "bbb_g256:") and "valu_busy" in v and \
                    v.get("source_hash") == source_hash(KERNEL_SOURCES["bbb"]):
                roof["valu"]["pmc"] = {"valu_busy": v.get("valu_busy"), "mfma_util": v.get("mfma_util"), "waves_per_simd": v.get("waves_per_simd"),
                                       # SURVEY 8(d) "report wall and cycles": active cycles / duration of the counter pass's own
                                       # dispatches -- a crude clock figure (it differs between passes: DESIGN.md 4, the power probe
                                       # is the better evidence); the issue-cycle fractions above are priced at the 2.4 GHz peak
                                       "clock_ghz_counter_pass": v.get("clock_ghz"), "dispatch_us_counter_pass": v.get("dispatch_us"),
                                       "key": k, "source": "profiles/pmc.json (rocprofv3 --pmc, tools/collect_pmc.py)"}
    except Exception:
        pass
    return roof


def attach_traffic(roof):
    """PMC traffic measured by tools/collect_traffic.py (rocprofv3 --pmc, separate passes) for the same kernel and
    launch shape -- only if the kernel sources have not changed since."""
    key = roof.pop("traffic_key", None)
    tj = _profile_file("traffic.json")
    if key is None or not os.path.exists(tj):
        return roof
    try:
        t = json.load(open(tj))
        e = t.get(key)
        if e is not None:
            fam = "lr" if key.startswith("lr") else "block_gemm" if key.startswith("block") else "bbb"
            if e.get("source_hash") == source_hash(KERNEL_SOURCES[fam]):
                roof["traffic"] = e["hbm_bytes_per_launch"]
                roof["traffic_source"] = e.get("source", "profiles/traffic.json")
            else:
                roof["traffic_note"] = "profiles/traffic.json entry is stale (kernel sources changed since it was measured)"
    except Exception:
        pass
    return roof


def cpu_baseline(dims, lr, batch, budget_s=15.0):
    """The oracle (op-for-op CPU restatement of the reference path, parity-pinned by tests/golden)
    timed on this box's host cores: S=1 sample_elbo calls incl. the CPU eps draw, like the
    reference's own loop body."""
    import numpy as np
    import torch
    from oracle import bnn_oracle as O
    from bnn_hip import synth
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))       # the cores this process may run on
    except Exception:
        pass
    sd = synth.synth_state_dict(dims[0], dims[1], dims[2], lr)
    p = O.NetParams.from_state_dict(sd, "classification", dims[0], lr, O.Prior.from_init([1.0], False))
    x, y = synth.synth_batch("classification", batch, dims[0], dims[2])
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    fn = O.sample_elbo_lr if lr else O.sample_elbo
    def rate(nthr, seconds, max_calls):
        """median seconds per sample_elbo(S=1) call at `nthr` intra-op threads (after one warm-up call; a thread count
        whose warm-up call alone takes > 1.5 s -- hundreds of threads on ~25 small elementwise ops per tensor -- is
        reported from that single call instead of being given more of the budget)"""
        torch.set_num_threads(nthr)
        t0 = time.perf_counter()
        fn(p, xt, yt, 0.5, 1)
        first = time.perf_counter() - t0
        if first > 1.5:
            return first, 1, first
        ts, t_end = [], time.perf_counter() + seconds
        while (time.perf_counter() < t_end or len(ts) < 3) and len(ts) < max_calls:
            t0 = time.perf_counter()
            fn(p, xt, yt, 0.5, 1)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), len(ts), float(sum(ts))

    with torch.no_grad():
        # the all-cores figure AND the one-thread figure; plus the thread count that is fastest for these op
        # sizes on this host (all cores is far from it on a many-core box: ~25 small elementwise ops per tensor) -- every
        # candidate gets the same 1.2 s probe (median of >= 3 calls), the fastest then gets the rest of the budget
        cands = sorted({t for t in (1, 4, 8, 16, 32, ncpu) if 1 <= t <= ncpu})
        probe = {t: rate(t, 1.2, 60) for t in cands}
        best_t = min(probe, key=lambda t: probe[t][0])
        spent = sum(v[2] for v in probe.values())
        med, calls, secs = rate(best_t, max(3.0, budget_s - spent), 400)
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": 1.0 / med, "unit": "MC-samples/s", "cores": best_t, "threads": best_t, "usable_cores": ncpu, "kind": "port",
            "sample": f"{calls} sample_elbo(S=1) calls of the CPU oracle (fp32, no_grad, incl. eps draw) in {secs:.1f} s, median "
                      f"{med*1e3:.2f} ms, at the fastest intra-op thread count of {cands} (= `cores` = `threads`: the threads actually used, 1.2 s probe each); "
                      f"usable cores (affinity) = {ncpu}, os.cpu_count() = {os.cpu_count()}; {model}",
            "all_cores": {"threads": ncpu, "value": 1.0 / probe[ncpu][0], "median_ms": probe[ncpu][0] * 1e3, "calls": probe[ncpu][1]},
            "one_thread": {"threads": 1, "value": 1.0 / probe[1][0], "median_ms": probe[1][0] * 1e3, "calls": probe[1][1]},
            "by_threads": {str(t): 1.0 / v[0] for t, v in probe.items()},
            "kl_elements_per_s": p.n_stochastic() / med}


SETTLE_MS = 120.0


def settle(ev, dist=None, every_eval=False, ms=None):
    """Untimed replays of a freshly built evaluator until it has run for ~`ms` milliseconds: its first ~30 ms under load are up to
    12 % slower than its steady state (see the module docstring).  Every rank runs the same number of replays (the probe's time is
    the maximum over the ranks)."""
    ms = SETTLE_MS if ms is None else ms
    if ms <= 0:
        return 0
    dt = run_groups(ev, 4, 1, None, dist, every_eval)
    n = max(1, int(ms * 1e-3 / (dt / 4) + 0.999))
    run_groups(ev, n, 0, None, dist, every_eval)
    return n + 5


def timed_config(engine, net, x, y, S_global, G, steps, dist=None, every_eval=False, graph=True):
    """samples/s and us per minibatch evaluation of one evaluator configuration (own warm-up)."""
    ev = make_evaluator(engine, net, x, y, S_global, G, graph=graph)
    g, full, rem, warm = plan_groups(steps, max(2 * G, steps // 10), G)
    settle(ev, dist, every_eval)
    dt = run_groups(ev, full, warm, None, dist, every_eval)
    return ev, S_global * full * g / dt, dt * 1e6 / (full * g)


def training_step_ms(dims, lr, batch, dev, mode, samples=2, steps=300):
    """One optimiser step of bnn_hip.train.GraphedTrainStep (forward of `samples` MC samples, backward, fused Adam) in the
    math mode of the run, HIP events around `steps` replays on fresh synthetic minibatches."""
    import torch
    from bnn_hip.optim import FusedAdam
    from bnn_hip.train import GraphedTrainStep
    net, x, y = build_net(dims, lr, batch, dev, mode, n_minibatches=1)
    x, y = x[0].contiguous(), y[0].contiguous()                 # build_net stacks the minibatches: [1, batch, ...]
    opt = FusedAdam(net.parameters(), lr=1e-4, capturable=True)
    g = GraphedTrainStep(net, opt, x, y, samples)
    for _ in range(5):
        g.step(x, y, 0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        g.step(x, y, 0.5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return {"variant": "LR" if lr else "BBB", "mc_samples": samples, "batch": batch, "ms_per_step": ms, "steps_per_s": 1e3 / ms,
            "what": "stage minibatch + forward + backward + fused Adam, one hipGraph replay per step"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a mislabelled run", file=sys.stderr)
        sys.exit(2)

    import torch
    # rehearsal of the N>1 code path on a box with fewer GPUs than ranks (never a measurement):
    # BNN_BENCH_REHEARSAL=1 puts every rank on cuda:0 and exchanges over gloo instead of RCCL
    rehearsal = os.environ.get("BNN_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    if not rehearsal and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()}", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import bnn_hip
    from bnn_hip import engine
    bnn_hip.set_math(args.math)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            assert dist.get_backend() == "nccl" and dist.get_world_size() == args.gpus
        bnn_hip.shard_samples(True)

    dims, lr = DIMS[args.net], args.variant == "lr"
    # ONE STEP = one launch group: `--group` independent minibatches through one launch per layer (one hipGraph replay)
    G, full, rem, warm = max(1, args.group), args.steps, 0, args.warmup
    # the wide stack has no task attached in BASELINE: Gaussian NLL over its 4096 outputs
    mode = "regression" if args.net == "wide" else "classification"
    net, x, y = build_net(dims, lr, args.batch, dev, mode, n_minibatches=G)
    S_local, S_global = args.samples, args.samples * world
    ev = make_evaluator(engine, net, x, y, S_global, G, graph=not args.no_graph)
    assert ev.s_local == S_local and ev.n_local == G * S_local
    tail = make_evaluator(engine, net, x, y, S_global, rem, graph=not args.no_graph) if rem else None
    if args.roofline_only:
        torch.cuda.synchronize()
        roof = layer2_roofline(ev, net, dims, args.batch, lr, args.math)
        print(json.dumps({"roofline": roof, "variant": args.variant}), flush=True)
        return
    # W untimed + K timed steps straight after building the evaluator: a device coming out of idle.  The launch groups of the
    # first ~30 ms run up to 12 % slower than the steady state (clocks, power state, caches: tools/warmup_probe.py,
    # profiles/r04_warmup_probe.log -- windows of 20 steps: 684, 634, 617, 614, 611 ... us per step), and the driver's 5 + 20 steps
    # last 17 ms: that figure is kept as `cold_start`; `value` is the same W + K steps measured again after `--settle-ms` of
    # the same replays (every rank the same number of them: run_groups returns the maximum over the ranks).
    dt = dt_cold = run_groups(ev, full, warm, tail, dist)
    n_settle = 0
    if args.settle_ms > 0 and full > 0:
        n_settle = max(1, int(args.settle_ms * 1e-3 / (dt_cold / full) + 0.999))
        run_groups(ev, n_settle, 0, None, dist)
        dt = run_groups(ev, full, warm, tail, dist)
    if dist is not None and run_groups.last_reduced is not None and full > 0:
        got = run_groups.last_reduced[(full + warm - 1) & 1][..., 3]         # every all-reduced row: the GLOBAL sample count
        assert bool((got == float(S_global)).all()), f"all-reduced sample counts {got.flatten().tolist()} != {S_global}"
    value = S_global * G * args.steps / dt
    nst = n_stochastic(dims)
    layers = "-".join(map(str, (dims[0], dims[1], dims[1], dims[2])))

    out = {
        "metric": "MC-forward-samples/sec + KL-elements/sec, 784-1200-1200-10 BNN" if args.net == "mnist"
        else f"MC-forward-samples/sec + KL-elements/sec, {layers} BNN",
        "value": value, "unit": "MC-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "us_per_minibatch": dt * 1e6 / (args.steps * G), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "f32": "f32", "bf16x3": "bf16x3 (split-bf16 operands, fp32-equivalent products)"}[args.math], "data": "synthetic",
        **({"rehearsal": "all ranks on cuda:0 over gloo; NOT a measurement"} if rehearsal else {}),
        "config": {"workload": f"{layers} {'LR' if lr else 'BBB'} forward-only ELBO evaluation of a stream of independent "
                               f"minibatches (3-layer forward with freshly sampled weights + log p/log q reductions over "
                               f"{nst} stochastic params + NLL per MC sample), batch {args.batch}, {S_local} MC sample(s) "
                               f"per GPU per minibatch, Gaussian prior, on-chip Philox eps; ONE STEP = one launch group = {G} "
                               f"minibatches resident in HBM through one launch per layer (one hipGraph replay): "
                               f"{G * S_global} MC forward samples per step",
                   "minibatches_per_step": G, "mc_samples_per_step_all_gpus": G * S_global,
                   "batch": args.batch, "mc_samples_per_gpu_per_step": S_local, "mc_samples_per_step": S_global,
                   "stochastic_params": nst, "hipgraph": not args.no_graph, "minibatches_per_launch_group": G,
                   "launches_timed": full + (1 if rem else 0),
                   "parallelism": (f"mc-sample-shard x{world}: every minibatch's {S_global} MC samples split over the ranks; ONE "
                                   f"RCCL sum all-reduce of the [{G}, 4] ELBO scalars per launch group, asynchronous")
                   if world > 1 else "single GPU"},
        "kl_elements_per_s": value * nst,
        "cold_start": {"value": S_global * G * args.steps / dt_cold, "ms_per_step": dt_cold * 1e3 / args.steps, "settle_steps_before_value": n_settle,
                       "note": "the same W warm-up + K timed steps straight after building the evaluator (a device coming out of idle: its "
                               "first ~30 ms under load run up to 12 % slower); `value` was measured after `settle_steps_before_value` "
                               "further untimed launch groups and W more warm-up steps"},
    }

    if rank == 0:
        out["roofline"] = attach_traffic(layer2_roofline(ev, net, dims, args.batch, lr, args.math))
        if world == 1 and not rehearsal:
            out["roofline"]["board_power"] = board_power_while(ev.replay)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dims, lr, args.batch)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    del ev, tail

    if not args.no_extras and args.net == "mnist" and not lr:
        extras = {}
        if world == 1:
            # the training loop's regime: one minibatch, one MC sample, every evaluation waits for the previous one
            e1, rate, us = timed_config(engine, net, x, y, 1, 1, 400)
            out["single_evaluation_in_flight"] = {
                "samples_per_s": rate, "us_per_evaluation": us,
                "layer2": {k: v for k, v in layer2_roofline(e1, net, dims, args.batch, False, args.math).items()
                           if k in ("kernel", "avg_launch_us", "frac", "plan")},
                "note": "same network, ONE minibatch, one hipGraph replayed back to back on one stream (latency of one ELBO evaluation)"}
            out["single_evaluation_us"] = us                        # (flat copies: the driver's parser keeps scalars)
            out["single_evaluation_samples_per_s"] = rate
            del e1
            # the same chain with 8 evaluations per hipGraph replay: every evaluation still waits for the previous one (one
            # stream, dependent launches), the ~8 us between two replays of a graph is paid once per eight
            e1b = make_evaluator(engine, net, x, y, 1, 1, per_replay=8)
            g_, full_, _, warm_ = plan_groups(100, 10, 1)
            settle(e1b)
            dt8 = run_groups(e1b, full_, warm_, None)
            out["single_evaluation_in_flight"]["eight_per_replay"] = {"us_per_evaluation": dt8 * 1e6 / (full_ * 8), "samples_per_s": full_ * 8 / dt8,
                                                                      "note": "8 dependent evaluations per graph replay (fresh epsilon each): the replay gap amortised"}
            out["single_evaluation_us_8_per_replay"] = dt8 * 1e6 / (full_ * 8)
            del e1b
            # ... and as a recorded list of C-ABI launches called again (engine.GraphedElbo(capture="calls")): no hipGraph, so none
            # of the ~8 us a graph replay spends around its nodes; the host pays a few us per launch instead
            try:
                e1c = make_evaluator(engine, net, x, y, 1, 1, graph="calls")
                g_, full_, _, warm_ = plan_groups(800, 80, 1)
                settle(e1c)
                dtc = run_groups(e1c, full_, warm_, None)
                out["single_evaluation_in_flight"]["recorded_launches"] = {
                    "us_per_evaluation": dtc * 1e6 / full_, "samples_per_s": full_ / dtc, "launches_per_evaluation": len(e1c.calls),
                    "note": "the evaluation's launches recorded once and called again per evaluation (no graph): bit-identical results"}
                out["single_evaluation_us_recorded_launches"] = dtc * 1e6 / full_
                del e1c
            except Exception as e:                                   # (a configuration that allocates while recording)
                out["single_evaluation_in_flight"]["recorded_launches"] = {"error": repr(e)[:200]}
            # MC-batched evaluations of ONE minibatch (C4's per-GPU share is 8 samples; 64 = C4 on one GPU)
            mc = []
            for (S, steps) in ((8, 200), (64, 60), (256, 24)):
                e2, rate, us = timed_config(engine, net, x, y, S, 1, steps)
                r2 = layer2_roofline(e2, net, dims, args.batch, False, args.math)
                mc.append({"mc_samples_per_evaluation": S, "samples_per_s": rate, "kl_elements_per_s": rate * nst,
                           "us_per_evaluation": us, "layer2_kernel": r2["kernel"], "layer2_us_per_launch": r2["avg_launch_us"],
                           "layer2_bound": r2["bound"], "layer2_frac": r2["frac"],
                           "layer2_hbm_algorithmic_frac": r2.get("hbm_algorithmic", r2)["frac"]})
                del e2
                if S <= 8:                                          # few samples: the same evaluation as recorded launches
                    try:
                        e2c = make_evaluator(engine, net, x, y, S, 1, graph="calls")
                        g_, full_, _, warm_ = plan_groups(2 * steps, steps // 5, 1)
                        settle(e2c)
                        dtc = run_groups(e2c, full_, warm_, None)
                        mc[-1]["us_per_evaluation_recorded_launches"] = dtc * 1e6 / full_
                        del e2c
                    except Exception as e:
                        mc[-1]["recorded_launches_error"] = repr(e)[:200]
            extras["mc_batched_one_minibatch"] = mc
            # the headline workload in the two math modes that meet ELBO rtol 1e-4 against the reference's fp32 arithmetic at
            # EVERY beta (plain bf16 does for beta >= 2^-6 only: DESIGN.md 2): exact-fp32 matrix core, split-bf16 operands
            modes = {}
            for mname in ("f32", "bf16x3"):
                if mname == args.math:
                    continue
                bnn_hip.set_math(mname)
                em, rate, us = timed_config(engine, net, x, y, 1, G, 4 * G if mname == "f32" else 8 * G)
                rm = attach_traffic(layer2_roofline(em, net, dims, args.batch, False, mname))
                modes[mname] = {"samples_per_s": rate, "us_per_minibatch": us, "kl_elements_per_s": rate * nst, "vs_headline": rate / value,
                                "minibatches_per_launch_group": G, "roofline": rm}
                out[f"{mname}_math_samples_per_s"] = rate            # (flat copies: the driver's parser keeps scalars)
                if mname == "bf16x3" and not rehearsal:
                    modes[mname]["board_power"] = board_power_while(em.replay, seconds=1.0)
                del em
            # ... and the local-reparameterisation network in exact-fp32 math (what set_math('bf16x3') runs an LR network as)
            bnn_hip.set_math("f32")
            net_lr32, _, _ = build_net(dims, True, args.batch, dev, mode, n_minibatches=1)
            em, rate, us = timed_config(engine, net_lr32, x, y, 1, G, 4 * G)
            modes["lr_f32"] = {"samples_per_s": rate, "us_per_minibatch": us, "minibatches_per_launch_group": G,
                               "roofline": layer2_roofline(em, net_lr32, dims, args.batch, True, "f32")}
            del em
            bnn_hip.set_math("bf16x3")                 # ... and in split-bf16 math (hidden layers K3b<X3>, output layer exact fp32)
            em, rate, us = timed_config(engine, net_lr32, x, y, 1, G, 8 * G)
            modes["lr_bf16x3"] = {"samples_per_s": rate, "us_per_minibatch": us, "minibatches_per_launch_group": G,
                                  "roofline": layer2_roofline(em, net_lr32, dims, args.batch, True, "bf16x3")}
            out["lr_bf16x3_math_samples_per_s"] = rate
            del em, net_lr32
            bnn_hip.set_math(args.math)
            extras["math_modes"] = modes
            # the headline with rocRAND's generator (Philox4x32-10, epsilon map version 1): a second library built with
            # PHILOX_ROUNDS=10 (csrc/Makefile: make rounds10), run in a child process through BNN_HIP_LIB
            r10 = os.path.join(REPO, "bayesian-neural-network_amd", "bnn_hip", "libbnn_hip_r10.so")
            if os.path.exists(r10) and not os.environ.get("BNN_HIP_LIB"):
                try:
                    cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline", "--steps", str(args.steps),
                                         "--warmup", str(args.warmup), "--group", str(G), "--batch", str(args.batch)],
                                        env=dict(os.environ, BNN_HIP_LIB=r10), capture_output=True, text=True, timeout=300)
                    d10 = json.loads(cp.stdout.strip().splitlines()[-1])
                    extras["philox_10_rounds"] = {"samples_per_s": d10["value"], "vs_headline": d10["value"] / value, "ms_per_step": d10["ms_per_step"],
                                                  "layer2_us_per_launch": d10["roofline"]["avg_launch_us"],
                                                  "note": "the same workload on libbnn_hip_r10.so (BNN_PHILOX_ROUNDS = 10: rocRAND's PHILOX4_32_10)"}
                    out["philox_10_rounds_samples_per_s"] = d10["value"]
                except Exception as e:      # a diagnostic figure: never fails the bench
                    extras["philox_10_rounds"] = {"error": repr(e)[:200]}
            # C3: the local-reparameterisation variant, same workload as the headline and one evaluation at a time
            net_lr, _, _ = build_net(dims, True, args.batch, dev, mode, n_minibatches=1)
            e3, rate, us = timed_config(engine, net_lr, x, y, 1, G, max(4 * G, 1024))
            r3 = layer2_roofline(e3, net_lr, dims, args.batch, True, args.math)
            del e3
            e4, rate1, us1 = timed_config(engine, net_lr, x, y, 1, 1, 400)
            r4 = layer2_roofline(e4, net_lr, dims, args.batch, True, args.math)
            del e4
            # ... and MC-batched evaluations of ONE minibatch (sample_elbo_lr's own loop: networks.py:211-225): the first layer's two
            # products are made once for all samples up to 23, the hidden layer runs over prepared fragments from 8
            mc_lr = []
            for (S, steps) in ((2, 400), (8, 200), (64, 60)):
                e7, rate7, us7 = timed_config(engine, net_lr, x, y, S, 1, steps)
                mc_lr.append({"mc_samples_per_evaluation": S, "samples_per_s": rate7, "us_per_evaluation": us7})
                del e7
            extras["lr_variant"] = {"samples_per_s": rate, "us_per_minibatch": us, "minibatches_per_launch_group": G,
                                    "mc_batched_one_minibatch": mc_lr,
                                    "roofline": attach_traffic(r3),
                                    "single_evaluation_in_flight": {"samples_per_s": rate1, "us_per_evaluation": us1,
                                                                    "layer2_us_per_launch": r4["avg_launch_us"],
                                                                    "layer2_hbm_frac": r4["frac"], "layer2_kernel": r4["kernel"]}}
            del net_lr
            # C5: 4096-4096-4096, 4 MC samples (its per-GPU share of 32), batch 128 and 1024
            wide = []
            for B in (128, 1024, 4096):
                wnet, wx, wy = build_net(DIMS["wide"], False, B, dev, "regression", n_minibatches=1)
                e5, rate, us = timed_config(engine, wnet, wx, wy, 4, 1, 40 if B == 128 else 12)
                r5 = attach_traffic(layer2_roofline(e5, wnet, DIMS["wide"], B, False, args.math))
                wide.append({"batch": B, "mc_samples_per_evaluation": 4, "samples_per_s": rate, "us_per_evaluation": us,
                             "kl_elements_per_s": rate * n_stochastic(DIMS["wide"]), "roofline": r5})
                del e5, wnet, wx, wy
            extras["wide_4096"] = wide
            # F1 / F2: the body of the reference's training loop (class_task.py:66-79: zero_grad, sample_elbo with
            # train_samples = 2, backward, Adam) as one captured graph, BBB and the reference's own default (local_reparam)
            extras["training_step"] = [training_step_ms(dims, lr_, args.batch, dev, mode) for lr_ in (False, True)]
        # C4 as SURVEY 8(e) words it: ONE minibatch, 64 / 512 MC samples over the ranks, one all-reduce per evaluation
        c4 = []
        for S_tot in (64, 512):
            if S_tot % world:
                continue
            e6, rate, us = timed_config(engine, net, x, y, S_tot, 1, 200 if S_tot == 64 else 40, dist, every_eval=True)
            c4.append({"mc_samples_per_evaluation": S_tot, "mc_samples_per_gpu": S_tot // world, "samples_per_s": rate,
                       "us_per_evaluation": us,
                       "collective": "one RCCL all-reduce of the 4-vector per evaluation, awaited before the next" if world > 1 else "none (1 GPU)"})
            del e6
            if world > 1:
                # the same with the collective OVERLAPPED -- evaluation i + 1 is launched while the all-reduce
                # of evaluation i is in flight (two slab slots; nobody waits for an ELBO before starting the next evaluation)
                e6, rate, us = timed_config(engine, net, x, y, S_tot, 1, 200 if S_tot == 64 else 40, dist, every_eval=False)
                c4.append({"mc_samples_per_evaluation": S_tot, "mc_samples_per_gpu": S_tot // world, "samples_per_s": rate,
                           "us_per_evaluation": us, "collective": "one RCCL all-reduce of the 4-vector per evaluation, asynchronous: "
                                                                  "the next evaluation is launched behind it (pipelined)"})
                del e6
                if S_tot // world <= 16:
                    # few samples per GPU: the evaluation as a recorded launch list instead of a hipGraph (no ~8 us of graph
                    # replay around ~60 us of kernels), pipelined as above
                    try:
                        e6, rate, us = timed_config(engine, net, x, y, S_tot, 1, 200, dist, every_eval=False, graph="calls")
                        c4.append({"mc_samples_per_evaluation": S_tot, "mc_samples_per_gpu": S_tot // world, "samples_per_s": rate,
                                   "us_per_evaluation": us, "collective": "pipelined, the evaluation replayed as a recorded launch list (no hipGraph)"})
                        del e6
                    except Exception as e:
                        c4.append({"mc_samples_per_evaluation": S_tot, "recorded_launches_error": repr(e)[:200]})
        extras["c4"] = c4
        if rank == 0:
            out["extras"] = extras
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
