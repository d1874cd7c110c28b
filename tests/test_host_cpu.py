"""CPU-side checks (no GPU): the C ABI loads and exports every symbol the header declares,
struct layouts agree with the header, argument validation works without touching a device,
and the host logic (sample sharding, ELBO assembly, the 2-rank all-reduce path over gloo)
is correct.  The compute itself is only ever exercised on a GPU (tests/test_gpu_parity.py);
here the layer launches are replaced by an oracle-backed stand-in INSIDE THE TEST ONLY."""
import ctypes as C
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "bnn_hip.h")


def test_library_exports_every_declared_symbol():
    from bnn_hip import _lib
    lib = _lib.load()
    src = open(HEADER).read()
    declared = set(re.findall(r"\b(bnn_[a-z0-9_]+)\s*\(", src))
    declared -= {"bnn_status"}
    assert {"bnn_bbb_linear_fwd", "bnn_lr_linear_fwd", "bnn_gauss_kl", "bnn_elbo_finalize", "bnn_bbb_final_fwd",
            "bnn_philox_normal", "bnn_cast_bf16", "bnn_version", "bnn_status_string"} <= declared
    for name in declared:
        assert hasattr(lib, name), f"libbnn_hip.so does not export {name}"
    assert set(_lib.EXPORTS) == declared
    assert lib.bnn_version() == _lib.ABI_VERSION
    assert lib.bnn_status_string(-5).decode().startswith("struct_bytes")


def test_struct_layouts_match_the_header(tmp_path):
    """Every ctypes mirror in bnn_hip/_lib.py has the size and the field offsets of its C struct."""
    from bnn_hip import _lib
    pairs = [("bnn_bbb_fwd_args", _lib.BbbFwdArgs), ("bnn_lr_fwd_args", _lib.LrFwdArgs),
             ("bnn_finalize_args", _lib.FinalizeArgs), ("bnn_bbb_bwd_args", _lib.BbbBwdArgs),
             ("bnn_lr_bwd_args", _lib.LrBwdArgs), ("bnn_adam_args", _lib.AdamArgs), ("bnn_prior", _lib.Prior)]
    lines, want = [], []
    for cname, cls in pairs:
        lines.append('printf("%%zu\\n", sizeof(%s));' % cname)
        want.append(C.sizeof(cls))
        for fname, _t in cls._fields_:
            lines.append('printf("%%zu\\n", offsetof(%s, %s));' % (cname, fname))
            want.append(getattr(cls, fname).offset)
    prog = tmp_path / "sz.c"
    prog.write_text('#include "%s"\n#include <stdio.h>\n#include <stddef.h>\nint main(){%s return 0;}' % (HEADER, "".join(lines)))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", str(prog), "-o", str(exe)], check=True)   # header is plain C
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == want


def test_argument_validation_without_a_device():
    """Launch functions reject bad arguments before any HIP call."""
    from bnn_hip import _lib as L
    lib = L.load()
    a = L.BbbFwdArgs()
    assert lib.bnn_bbb_linear_fwd(C.byref(a), None) == -5          # struct_bytes mismatch
    a.struct_bytes = C.sizeof(L.BbbFwdArgs)
    assert lib.bnn_bbb_linear_fwd(C.byref(a), None) == -2          # shape
    a.n_samples = a.batch = a.in_features = a.out_features = 4
    assert lib.bnn_bbb_linear_fwd(C.byref(a), None) == -1          # NULL pointers
    f = L.FinalizeArgs()
    f.struct_bytes = C.sizeof(L.FinalizeArgs)
    assert lib.bnn_elbo_finalize(C.byref(f), None) == -2
    assert lib.bnn_gauss_kl(None, None, 10, 1.0, None, 0, None, None) == -1
    assert lib.bnn_philox_normal(None, 1, 0, 0, 1, 1, 1, None) == -1
    assert lib.bnn_bbb_linear_fwd_workspace_bytes(2, 1200) == (1 + 2 * 300) * 16
    assert lib.bnn_lr_linear_fwd_workspace_bytes(1200) == (1 + 300) * 16
    lb = L.LrBwdArgs()
    assert lib.bnn_lr_linear_bwd(C.byref(lb), None) == -5
    lb.struct_bytes = C.sizeof(L.LrBwdArgs)
    assert lib.bnn_lr_linear_bwd(C.byref(lb), None) == -2
    ad = L.AdamArgs()
    assert lib.bnn_adam_step(C.byref(ad), None) == -5
    ad.struct_bytes = C.sizeof(L.AdamArgs)
    assert lib.bnn_adam_step(C.byref(ad), None) == -2
    assert lib.bnn_nll_bwd(None, None, None, None, 1, 1, 1, 0, 1.0, None) == -1


def test_cpu_tensors_are_refused():
    import bnn_hip
    import networks
    layer = networks.BayesianLinear(4, 3, [-0.2, 0.2], [-5, -4], [1.0], False)
    with pytest.raises(bnn_hip.BnnHipError):
        layer(torch.rand(2, 4))
    net = networks.BayesianNetwork(dict(input_shape=4, classes=3, batch_size=2, hidden_units=8, mode="classification",
                                        mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False,
                                        local_reparam=True))
    with pytest.raises(bnn_hip.BnnHipError):
        net.sample_elbo_lr(torch.rand(2, 1, 2, 2), torch.zeros(2, dtype=torch.long), 0.5, 2)


def test_drop_in_surface():
    """Names, constructor keys, parameter names/shapes, attributes the reference's callers touch
    (SURVEY §8(b1))."""
    import copy
    import networks
    from config import DEVICE, ClassConfig, RegConfig, RLConfig   # noqa: F401
    mp = dict(input_shape=784, classes=10, batch_size=128, hidden_units=1200, mode="classification",
              mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0], mixture_prior=False, local_reparam=False)
    net = networks.BayesianNetwork(mp)
    assert list(net.state_dict().keys()) == [f"l{i}.{n}" for i in (1, 2, 3)
                                             for n in ("weight_mu", "weight_rho", "bias_mu", "bias_rho")]
    assert tuple(net.l1.weight_mu.shape) == (1200, 784) and tuple(net.l3.bias_rho.shape) == (10,)
    assert net.l1.weight.mu is net.l1.weight_mu and net.l1.bias.rho is net.l1.bias_rho
    assert all(isinstance(c, (networks.BayesianLinear, torch.nn.ReLU)) for c in net.children())
    assert net.batch_size == 128 and net.l1.log_prior == 0 and net.l1.log_variational_posterior == 0
    lr = networks.BayesianNetwork(dict(mp, local_reparam=True))
    assert tuple(lr.l1.weight_mu.shape) == (784, 1200) and lr.l1.kl_cost == 0 and lr.l1.weight_prior == [0, 1.0]
    with pytest.raises(AssertionError):
        networks.BayesianLinear(3, 2, [-0.2, 0.2], [-5, -4], [1.0], True)       # mixture needs 3 values
    with pytest.raises(AssertionError):
        networks.BayesianLinearLR(3, 2, [-0.2, 0.2], [-5, -4], [0.5, 0, -6])     # LR asserts 1 value
    with pytest.raises(AssertionError):
        net.sample_elbo_lr(None, None, 0.5, 1)
    copy.deepcopy(net)
    net.l1.weight_mu.data = torch.zeros_like(net.l1.weight_mu)                   # weight_pruning.py:111
    assert all(("mu" in n) or ("rho" in n) for n, _ in net.named_parameters())
    assert ClassConfig.hidden_units == 1200 and RegConfig.noise_tolerance == .1 and RLConfig.prior_init == [0.5, -0, -6]


def test_shard_range_partitions_the_samples():
    from bnn_hip.engine import shard_range
    for S in (1, 2, 7, 64, 65):
        for W in (1, 2, 3, 8):
            got = [shard_range(S, r, W) for r in range(W)]
            idx = [i for lo, n in got for i in range(lo, lo + n)]
            assert idx == list(range(S))
            assert max(n for _, n in got) - min(n for _, n in got) <= 1


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, os.path.join({repo!r}, "bayesian-neural-network_amd")); sys.path.insert(0, {repo!r})
    import numpy as np, torch, torch.distributed as dist
    import bnn_hip, networks
    from bnn_hip import engine, ops, synth
    from oracle import bnn_oracle as O
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bnn_hip.shard_samples(True)

    # TEST-ONLY stand-in for the device launches: the oracle evaluates the local samples and the
    # engine's own sharding / all-reduce / ELBO assembly code runs unchanged on top of it.
    S, B, dims = 5, 8, (6, 7, 3)
    sd = synth.synth_state_dict(*dims, False)
    p = O.NetParams.from_state_dict(sd, "classification", dims[0], False, O.Prior.from_init([1.0], False))
    x, y = synth.synth_batch("classification", B, dims[0], dims[2])
    eps_all = [[torch.from_numpy(a) for a in synth.synth_eps(p.eps_shapes(B), s)] for s in range(S)]
    def fake_run_layers(layers, xin, n_local, first, **kw):
        logits, lps, lqs = [], [], []
        for s in range(first, first + n_local):
            out, lp, lq = O.network_forward(p, xin, eps_all[s])
            logits.append(out); lps.append(lp); lqs.append(lq)
        return torch.stack(logits), dict(log_prior=torch.stack(lps), log_q=torch.stack(lqs),
                                         nll=torch.stack([O.nll(l, torch.from_numpy(y), "classification") for l in logits]), kl=None)
    engine.run_layers = fake_run_layers
    engine.collect_injected = lambda *a, **k: None
    net = networks.BayesianNetwork(dict(input_shape=dims[0], classes=dims[2], batch_size=B, hidden_units=dims[1],
                                        mode="classification", mu_init=[-0.2, 0.2], rho_init=[-5, -4], prior_init=[1.0],
                                        mixture_prior=False, local_reparam=False))
    with torch.no_grad():
        got = net.sample_elbo(torch.from_numpy(x).view(B, dims[0]), torch.from_numpy(y), 0.25, S)
    ref = O.sample_elbo(p, torch.from_numpy(x), torch.from_numpy(y), 0.25, S, eps=eps_all)
    for g, r in zip(got, ref):
        assert tuple(g.shape) == tuple(r.shape), (g.shape, r.shape)
        np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=2e-6)
    lo, n = engine.shard_range(S, rank, world)
    assert n in (2, 3) and bnn_hip.runtime.state.counter == S      # every rank advanced by the GLOBAL count
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
''')


def test_two_rank_gloo_sample_sharding(tmp_path):
    """world_size 2 over gloo: each rank evaluates its shard of the MC samples, the 3 ELBO sums
    are all-reduced, both ranks assemble the same loss tuple as the single-process oracle."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(repo=REPO))
    port = str(29500 + (os.getpid() % 2000))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


BENCH_WORKER = textwrap.dedent(r'''
    import os, sys
    sys.path.insert(0, r"{repo}"); sys.path.insert(0, os.path.join(r"{repo}", "bayesian-neural-network_amd"))
    import torch, torch.distributed as dist
    import bench
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class FakeEvaluator:
        """Stands in for engine.GraphedElbo: a replay runs `per_replay` evaluations, each depositing its
        4-vector {{evaluation index, rank, 0, 1}} at the device-side ring cursor (here a Python int)."""
        def __init__(self, slab, j, n, per_replay):
            self.slab, self.j, self.n, self.per_replay, self.stream, self.pos, self.count = slab, j, n, per_replay, None, 0, 0
        def replay(self):
            for _ in range(self.per_replay):
                self.slab[self.pos, self.j] = torch.tensor([float(self.count * self.n + self.j), float(rank), 0.0, 1.0])
                self.pos = (self.pos + 1) % self.slab.shape[0]
                self.count += 1

    for (nstr, E, ar_every, warmup, steps) in ((3, 4, 16, 24, 120), (4, 2, 8, 10, 46), (1, 1, 4, 3, 9)):
        slab = torch.zeros((2 * ar_every, nstr, 4))
        evs = [FakeEvaluator(slab, j, nstr, E) for j in range(nstr)]
        dt = bench.run_steps(evs, steps, warmup, dist, slab, ar_every)
        assert dt > 0
        h = bench.run_steps.last_flushed_half
        if h is not None:                                   # every row of the last fully reduced half: summed over the ranks
            rows = slab[h * ar_every:(h + 1) * ar_every]
            assert bool((rows[:, :, 3] == float(world)).all()), rows[:, :, 3]
            assert bool((rows[:, :, 1] == sum(range(world))).all())
            idx = rows[:, :, 0] / world                     # the same evaluation index on every rank
            assert bool((idx == idx.round()).all())
        total = sum(e.count for e in evs)
        assert total == warmup + steps, (total, warmup, steps)
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
''')


def test_bench_ring_allreduce_bookkeeping_two_ranks(tmp_path):
    """bench.run_steps' N>1 bookkeeping (ring halves, one all-reduce per half, partial flush at the barriers,
    matching collective counts on every rank) on 2 ranks over gloo with stand-in evaluators: no hang, every
    fully flushed row is the sum over the ranks, exactly warmup + steps evaluations ran."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER.format(repo=REPO))
    port = str(30500 + (os.getpid() % 2000))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


def test_bench_times_any_step_count_exactly():
    """bench.plan_steps: whole graph launches plus one tail launch always add up to the K the caller asked for, the
    warm-up is never shorter than asked, E divides the all-reduce period in the N>1 path."""
    import bench
    for steps in (1, 2, 3, 5, 7, 20, 50, 97, 100, 1001, 3000):
        for warmup in (0, 1, 5, 33, 300):
            for nstr in (1, 3, 4):
                for ar in (None, 16, 64):
                    E, main, tail, wrun = bench.plan_steps(steps, warmup, 4, nstr, ar)
                    assert E >= 1 and main % E == 0 and 0 <= tail < E and main + tail == steps
                    assert wrun >= warmup and wrun % E == 0 and wrun - warmup < E
                    assert ar is None or ar % E == 0
    assert bench.plan_steps(3000, 300, 4, 4)[0] == 4 and bench.plan_steps(20, 5, 4, 4)[0] == 5
