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
             ("bnn_lr_bwd_args", _lib.LrBwdArgs), ("bnn_adam_args", _lib.AdamArgs), ("bnn_prior", _lib.Prior),
             ("bnn_prepare_args", _lib.PrepareArgs), ("bnn_loss_args", _lib.LossArgs)]
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
    assert lib.bnn_bbb_linear_fwd_workspace_bytes(2, 1200) == (1 + 2 * 75 * 8) * 16    # 8 statistics writers per 16-feature tile
    assert lib.bnn_bbb_split_scratch_bytes(8, 128, 1200) == 768 + 152 * 8 * 32768 and \
        lib.bnn_bbb_split_scratch_zero_bytes(8, 128, 1200) == 768
    assert lib.bnn_lr_linear_fwd_workspace_bytes(1200) == (1 + 4096) * 16 and lib.bnn_lr_linear_fwd_workspace_bytes(20000) == (1 + 5000) * 16
    lb = L.LrBwdArgs()
    assert lib.bnn_lr_linear_bwd(C.byref(lb), None) == -5
    lb.struct_bytes = C.sizeof(L.LrBwdArgs)
    assert lib.bnn_lr_linear_bwd(C.byref(lb), None) == -2
    # the training step's optional arguments are validated on the host too (fake non-NULL device addresses: nothing
    # is dereferenced before the checks)
    fake = 0x1000
    lb.n_samples, lb.batch, lb.in_features, lb.out_features = 2, 4, 8, 24
    for fld in ("x", "gy", "w_mu", "w_rho", "b_mu", "b_rho", "g_w_mu", "g_w_rho", "g_b_mu", "g_b_rho"):
        setattr(lb, fld, fake)
    assert lib.bnn_lr_linear_bwd(C.byref(lb), None) == -1          # neither the forward's v nor its hfac
    lb.hfac, lb.relu, lb.y, lb.sigma_p = fake, 1, fake, 1.0
    assert lib.bnn_lr_linear_bwd(C.byref(lb), None) == -1          # a fused ReLU needs the mask pass, and that pass v
    lb.relu, lb.math = 0, 7
    assert lib.bnn_lr_linear_bwd(C.byref(lb), None) == L.load().bnn_lr_linear_bwd(C.byref(lb), None) < 0   # bad enum
    lf = L.LrFwdArgs()
    lf.struct_bytes = C.sizeof(L.LrFwdArgs)
    assert lib.bnn_lr_linear_fwd(C.byref(lf), None) == -2
    assert lib.bnn_stage_inputs_cast(fake, fake, 64, None, None, 0, None, 0.0, fake + 2, None) < 0      # misaligned bf16 copy
    assert lib.bnn_stage_inputs_cast(fake, fake, 60, None, None, 0, None, 0.0, fake, None) < 0         # the cast needs 16-byte sizes
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
    assert [n for n, _ in net.named_children()] == ["l1", "l1_act", "l2", "l2_act", "l3"]      # networks.py:160-164
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
    # the sharded ELBO is forward-only: nothing all-reduces parameter gradients, so a differentiable call is refused
    # rather than leaving every rank with the gradient of its own samples only
    try:
        net.sample_elbo(torch.from_numpy(x).view(B, dims[0]), torch.from_numpy(y), 0.25, S)
        raise SystemExit("sharded differentiable sample_elbo was not refused")
    except bnn_hip.BnnHipError as e:
        assert "forward-only" in str(e)
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
        # Stands in for engine.GraphedElbo: a replay evaluates G minibatches and leaves their 4-vectors
        # (replay index * G + minibatch, rank, 0, 1) in the static `sums` tensor.
        def __init__(self, G):
            self.G, self.count, self.sums = G, 0, torch.zeros((G, 4)) if G > 1 else torch.zeros(4)
        def replay(self):
            v = self.sums.view(self.G, 4)
            for m in range(self.G):
                v[m] = torch.tensor([float(self.count * self.G + m), float(rank), 0.0, 1.0])
            self.count += 1
            return self.sums

    for (steps, warmup, group, every) in ((120, 24, 16, False), (46, 10, 8, False), (9, 3, 4, False), (20, 5, 256, False),
                                          (12, 4, 1, True), (12, 4, 1, False)):     # the last: C4 pipelined (one minibatch per
                                                                                    # evaluation, its all-reduce behind the next launch)
        G, full, rem, warm = bench.plan_groups(steps, warmup, group)
        main, tail = FakeEvaluator(G), (FakeEvaluator(rem) if rem else None)
        dt = bench.run_groups(main, full, warm, tail, dist, every)
        assert dt > 0 and main.count == warm + full and (tail is None or tail.count == 1)
        assert (main.count - warm) * G + (rem if tail else 0) == steps
        if every:                                           # reduced in place, evaluation by evaluation
            assert bool((main.sums.view(G, 4)[:, 3] == float(world)).all())
        else:
            slot = bench.run_groups.last_reduced[(full + warm - 1) & 1].view(G, 4)   # the last launch group's collective
            assert bool((slot[:, 3] == float(world)).all()) and bool((slot[:, 1] == sum(range(world))).all())
            assert bool((slot[:, 0] == world * (torch.arange(G) + (warm + full - 1) * G)).all()), slot[:, 0]
            if tail is not None:
                assert bool((tail.sums.view(rem, 4)[:, 3] == float(world)).all())
    # the settle phase ahead of a timed region: every rank replays the same number of launch groups (its length comes from a
    # time that is the maximum over the ranks), collectives included
    for every in (False, True):
        ev = FakeEvaluator(4 if not every else 1)
        n = bench.settle(ev, dist, every, ms=15.0)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([ev.count], dtype=torch.int64))
        assert n == ev.count and n >= 6 and all(int(c) == ev.count for c in counts), (n, ev.count, counts)
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
''')


def test_bench_group_allreduce_bookkeeping_two_ranks(tmp_path):
    '''bench.run_groups' N>1 bookkeeping (two slab slots, one asynchronous all-reduce per launch group, the remainder
    group's own collective, the per-evaluation collective of the C4 mode) on 2 ranks over gloo with stand-in
    evaluators: no hang, every reduced row is the sum over the ranks, exactly K steps are timed.'''
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
    '''bench.plan_groups: whole launch groups plus one remainder launch always add up to the K the caller asked for and
    the warm-up is never shorter than asked.'''
    import bench
    for steps in (1, 2, 3, 5, 7, 20, 50, 97, 100, 1001, 3000, 8192):
        for warmup in (0, 1, 5, 33, 300, 512):
            for group in (1, 4, 16, 256):
                G, full, rem, warm = bench.plan_groups(steps, warmup, group)
                assert 1 <= G <= min(group, steps) and full * G + rem == steps and 0 <= rem < G
                assert warm * G >= warmup and warm * G - warmup < G
    assert bench.plan_groups(20, 5, 256) == (20, 1, 0, 1) and bench.plan_groups(8192, 512, 256) == (256, 32, 0, 2)


def test_bench_gpus_flag_is_honoured_or_refused():
    '''`bench.py --gpus N` never reports a run of another size: under a launcher whose WORLD_SIZE disagrees it exits 2
    before touching a device, and without a launcher it starts N rank processes itself (here, with no GPU, every
    rank refuses for want of devices and the parent relays that: exit 2, no JSON line).'''
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    if not torch.cuda.is_available():
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                           env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 2 and r.stderr.count("2 ranks need 2 GPUs") == 2 and not r.stdout.strip()


def _plan_args(cls, **kw):
    a = cls()
    a.struct_bytes = C.sizeof(cls)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_launch_plans_are_a_function_of_the_shape():
    '''bnn_bbb_plan / bnn_lr_plan (no device needed: they read shapes, dtypes and pointer alignment only): the forms,
    tile widths and wave counts the BASELINE shapes launch with, and the invariants every plan must keep -- whole
    waves in fours where there is enough work, LDS within 160 KiB, the requested form honoured when it applies.'''
    from bnn_hip import _lib as L
    lib = L.load()
    P = 0x10000                                             # any 16-byte aligned non-null address: plans never dereference

    def bbb(S, B, K, N, math=L.MATH_BF16, xdt=L.BF16, form=0, scratch=False, sigma=False, ydt=L.F32):
        x3 = math == L.MATH_BF16X3
        a = _plan_args(L.BbbFwdArgs, n_samples=S, batch=B, in_features=K, out_features=N, x=P, x_dtype=xdt, w_mu=P, w_rho=P,
                       b_mu=P, b_rho=P, math=math, y=P, y_dtype=ydt, form=form, split_scratch=P if scratch else None,
                       split_scratch_bytes=lib.bnn_bbb_split_scratch_bytes(S, B, N) if scratch else 0,
                       w_sigma=P if sigma else None, x_lo=P if (x3 and xdt == L.BF16) else None, y_lo=P if (x3 and ydt == L.BF16) else None)
        a.prior.sigma_p = 1.0
        pl = L.Plan()
        assert lib.bnn_bbb_plan(C.byref(a), C.byref(pl)) == 0
        return pl

    def lr(S, B, K, N, math=L.MATH_BF16, xdt=L.BF16, form=0, sq=False, frag=False, scratch=False, xps=0):
        a = _plan_args(L.LrFwdArgs, n_samples=S, batch=B, in_features=K, out_features=N, x=P, x_dtype=xdt, x_per_sample=xps, w_mu=P, w_rho=P,
                       b_mu=P, b_rho=P, math=math, y=P, form=form, x_sq=P if sq else None, w_frag=P if frag else None,
                       split_scratch=P if scratch else None,
                       split_scratch_bytes=lib.bnn_lr_split_scratch_bytes(S, B, N) if scratch else 0)
        pl = L.Plan()
        assert lib.bnn_lr_plan(C.byref(a), C.byref(pl)) == 0
        return pl

    # C2, one sample: 150 tiles of 8 features cover the chip, 12 waves split the 19 super-steps
    pl = bbb(1, 128, 1200, 1200)
    assert (pl.form, pl.k_classes, pl.waves, pl.blocks, pl.features_per_block) == (L.FORM_TILE, 2, 12, 150, 8)
    pl = bbb(1, 128, 784, 1200, xdt=L.F32)
    assert (pl.form, pl.k_classes, pl.blocks) == (L.FORM_TILE, 2, 150)
    # C4's per-GPU share (8 samples): the K-sliced block GEMM when a scratch is there -- 5 slices put 760 blocks of 8
    # k-steps on the 256 CUs, 3 x 8 = 24 steps per SIMD against the ideal 22.3 -- else whole 16-feature tiles, 4-wave blocks
    pl = bbb(8, 128, 1200, 1200, scratch=True)
    assert (pl.form, pl.k_slices, pl.waves, pl.blocks) == (L.FORM_GEMM_KSLICE, 5, 4, 760)
    assert [bbb(S, 128, 1200, 1200, scratch=True).k_slices for S in (4, 16, 32, 64)] == [6, 3, 2, 1]
    pl = bbb(8, 128, 1200, 1200)
    assert (pl.form, pl.k_classes, pl.waves, pl.blocks) == (L.FORM_TILE, 1, 4, 600)
    assert bbb(8, 128, 784, 1200, scratch=True).form == L.FORM_GEMM_KSLICE
    assert bbb(8, 128, 784, 1200, xdt=L.F32, scratch=True).form == L.FORM_TILE          # the block GEMM streams bf16 x
    # the throughput regime: block GEMM from 450 (64-feature group x sample) items; slices stop paying once the
    # whole-K blocks alone balance
    assert bbb(24, 128, 1200, 1200).form == L.FORM_GEMM and bbb(23, 128, 1200, 1200).form == L.FORM_TILE
    assert bbb(256, 128, 1200, 1200, scratch=True).form == L.FORM_GEMM
    pl = bbb(256, 128, 1200, 1200)
    assert (pl.form, pl.waves, pl.blocks, pl.features_per_block) == (L.FORM_GEMM, 4, 19 * 256, 64)
    assert bbb(256, 128, 1200, 1200, math=L.MATH_F32, xdt=L.F32).form == L.FORM_TILE      # fp32 math has no GEMM form
    # the headline launch: hoisted sigma -> K1b2, two pairs per 8-wave block, 64.5 KiB of LDS (two blocks per CU)
    pl = bbb(256, 128, 1200, 1200, sigma=True)
    assert (pl.form, pl.waves, pl.blocks, pl.lds_bytes) == (L.FORM_GEMM, 8, 19 * 128, 2 * (4 * 256 + 2 * 512) * 16 + 512)
    # split-bf16 math: the same form over (hi, lo) plane pairs, 48 KiB per staging buffer; no K-sliced form, and without the
    # hoisted sigma (or on fp32 x: split in registers) the tile form
    pl = bbb(256, 128, 1200, 1200, math=L.MATH_BF16X3, sigma=True, ydt=L.BF16)
    assert (pl.form, pl.waves, pl.blocks, pl.lds_bytes) == (L.FORM_GEMM, 8, 19 * 128, (4 * 256 + 2 * 2 * 1024) * 16)
    assert bbb(256, 128, 1200, 1200, math=L.MATH_BF16X3).form == L.FORM_TILE
    assert bbb(256, 128, 784, 1200, math=L.MATH_BF16X3, sigma=True, xdt=L.F32).form == L.FORM_TILE
    assert bbb(8, 128, 1200, 1200, math=L.MATH_BF16X3, sigma=True, scratch=True).form == L.FORM_TILE
    assert bbb(4, 128, 1200, 1200, math=L.MATH_BF16X3, sigma=True, form=L.FORM_GEMM).form == L.FORM_GEMM
    assert bbb(3, 128, 1200, 1200, math=L.MATH_BF16X3, sigma=True, form=L.FORM_GEMM).form == L.FORM_TILE     # fewer than two blocks' worth of pairs
    a = _plan_args(L.BbbFwdArgs, n_samples=8, batch=128, in_features=1200, out_features=1200, x=P, x_dtype=L.BF16, w_mu=P, w_rho=P, b_mu=P,
                   b_rho=P, math=L.MATH_BF16X3, y=P)
    a.prior.sigma_p = 1.0
    assert lib.bnn_bbb_plan(C.byref(a), C.byref(L.Plan())) == -1           # BNN_ERR_NULL: bf16 x in this mode is a plane PAIR
    # C5's per-GPU share: K-sliced GEMM for the 16.8 M-weight layer when a scratch is there, the tile form without
    pl = bbb(4, 128, 4096, 4096, scratch=True)
    assert pl.form == L.FORM_GEMM_KSLICE and pl.k_slices >= 2 and pl.blocks == 64 * 4 * pl.k_slices
    assert bbb(4, 128, 4096, 4096).form == L.FORM_TILE
    # below 4 samples and for small layers the narrow-tile form stays
    assert bbb(3, 128, 1200, 1200, scratch=True).form == L.FORM_TILE and bbb(8, 128, 1200, 64, scratch=True).form == L.FORM_TILE
    # a form preference is honoured when the arguments allow it, ignored otherwise
    assert bbb(2, 128, 1200, 1200, form=L.FORM_GEMM).form == L.FORM_GEMM
    assert bbb(2, 128, 1200, 1200, math=L.MATH_F32, xdt=L.F32, form=L.FORM_GEMM).form == L.FORM_TILE
    assert bbb(256, 128, 1200, 1200, form=L.FORM_TILE).form == L.FORM_TILE
    # LR: K3a until 300 block-GEMM items AND the bf16 x / x^2 pair; the output layer in 32-row blocks
    assert lr(1, 128, 1200, 1200).form == L.FORM_TILE and lr(64, 128, 1200, 1200).form == L.FORM_TILE
    pl = lr(64, 128, 1200, 1200, sq=True)
    assert (pl.form, pl.waves) == (L.FORM_GEMM, 4)
    assert lr(64, 128, 1200, 1200, sq=True, frag=True).waves == 16 and lr(64, 128, 1200, 256, sq=True, frag=True).waves == 4
    # over prepared fragments the block form runs from 7 samples (per-sample inputs: xps), in the narrowest blocks whose launch
    # is still one round of <= 256 (the measured optima of tools/lr_mid_sweep.py)
    for S, waves, blocks in ((7, 4, 133), (8, 4, 152), (10, 4, 190), (16, 8, 160), (24, 8, 240), (32, 16, 160), (64, 16, 320)):
        pl = lr(S, 128, 1200, 1200, sq=True, frag=True, xps=1)
        assert (pl.form, pl.waves, pl.blocks) == (L.FORM_GEMM, waves, blocks), (S, pl.form, pl.waves, pl.blocks)
    assert lr(6, 128, 1200, 1200, sq=True, frag=True, xps=1).form == L.FORM_TILE and lr(10, 128, 1200, 1200, sq=True, xps=1).form == L.FORM_TILE
    pl = lr(1, 128, 1200, 10)
    assert (pl.form, pl.batch_rows, pl.k_classes) == (L.FORM_TILE, 32, 4)
    # K3s: with a split scratch, 1-2 samples on the wide layers run as 32-feature groups x K slices in ONE round of blocks
    # (38 groups x 4 slices of <= 10 k-steps; 76 units x 3 slices of 13); more units than that would queue: K3a
    pl = lr(1, 128, 1200, 1200, scratch=True)
    assert (pl.form, pl.k_slices, pl.blocks, pl.features_per_block, pl.waves) == (L.FORM_GEMM_KSLICE, 4, 152, 32, 8)
    pl = lr(1, 128, 784, 1200, scratch=True)
    assert (pl.form, pl.k_slices, pl.blocks) == (L.FORM_GEMM_KSLICE, 4, 152)
    pl = lr(2, 128, 1200, 1200, scratch=True, xps=1)
    assert (pl.form, pl.k_slices, pl.blocks) == (L.FORM_GEMM_KSLICE, 3, 228)
    assert lr(3, 128, 1200, 1200, scratch=True, xps=1).form == L.FORM_TILE            # (K3a's 225 blocks are one round)
    # ... and up to 512 blocks (two fit a CU) where the tile form itself would need a second round of its 16-feature blocks:
    # 4 samples = 300 tile blocks -> 152 units x 3 slices; 5 samples = 190 units: K3a
    pl = lr(4, 128, 1200, 1200, scratch=True, xps=1)
    assert (pl.form, pl.k_slices, pl.blocks) == (L.FORM_GEMM_KSLICE, 3, 456)
    assert lr(5, 128, 1200, 1200, scratch=True, xps=1).form == L.FORM_TILE
    # 2 .. 64 samples on ONE input (x_per_sample = 0: the first layer of sample_elbo_lr / predict): the units count one sample --
    # their products are made once, the epilogue runs per sample; beyond 64 the per-sample forms
    for S in (2, 10, 23, 64):
        pl = lr(S, 128, 784, 1200, scratch=True, xdt=L.F32)
        assert (pl.form, pl.k_slices, pl.blocks) == (L.FORM_GEMM_KSLICE, 4, 152), S
    assert lr(65, 128, 784, 1200, scratch=True).form == L.FORM_TILE and lr(10, 128, 784, 1200).form == L.FORM_TILE
    assert lr(10, 128, 784, 1200, scratch=True, form=L.FORM_TILE).form == L.FORM_TILE
    assert lr(1, 128, 784, 1200, scratch=True, xdt=L.F32).form == L.FORM_GEMM_KSLICE        # the first layer reads the fp32 minibatch itself
    assert lr(1, 128, 1200, 1200, scratch=True, math=L.MATH_F32, xdt=L.F32).form == L.FORM_TILE and lr(1, 128, 1200, 10, scratch=True).form == L.FORM_TILE
    assert lr(1, 128, 1204, 1200, scratch=True).form == L.FORM_TILE and lr(1, 128, 1200, 1202, scratch=True).form == L.FORM_TILE
    assert lr(1, 128, 1200, 1200, scratch=True, form=L.FORM_TILE).form == L.FORM_TILE
    assert lib.bnn_lr_split_scratch_zero_bytes(1, 128, 1200) == 256 and lib.bnn_lr_split_scratch_bytes(1, 128, 1200) == 256 + 38 * 8 * 32768
    # invariants over a sweep of shapes
    for S in (1, 2, 3, 8, 31, 64):
        for B in (1, 16, 100, 128, 300):
            for K in (1, 8, 50, 784, 1200, 4096):
                for N in (1, 10, 50, 1200, 4096):
                    for pl in (bbb(S, B, K, N), lr(S, B, K, N), bbb(S, B, K, N, math=L.MATH_F32, xdt=L.F32),
                               bbb(S, B, K, N, math=L.MATH_BF16X3, sigma=True, ydt=L.BF16)):
                        assert 1 <= pl.waves <= 16 and pl.lds_bytes <= 160 * 1024 and pl.blocks >= 1
                        assert pl.form in (L.FORM_TILE, L.FORM_GEMM, L.FORM_GEMM_KSLICE)
                        assert pl.k_classes in (1, 2, 4) and pl.batch_rows in (32, 128)


def test_committed_instruction_mix_matches_the_kernel_sources():
    """`roofline.frac` of the headline is priced from profiles/isa_mix.json, and bench.py drops an entry whose source hash is not
    the current one (a stale mix must not price a changed kernel): a kernel-source change without `python tools/make_isa_mix.py`
    would leave the driver's bench line without its fraction.  (traffic.json / pmc.json / valu_floor.json come from GPU passes --
    tools/profile_final.sh -- and only lose their fields when stale.)"""
    import json
    import bench
    mix = json.load(open(os.path.join(REPO, "profiles", "isa_mix.json")))
    for key, fam in (("bbb_fwd_gemm2_kernel<4,2,philox>", "bbb"), ("bbb_fwd_gemm2_kernel<4,2,philox,x3>", "bbb"),
                     ("bbb_fwd_gemm_kernel<4,true,philox>", "bbb"), ("lr_fwd_gemm_kernel<16,true,2>", "lr")):
        assert mix[key]["source_hash"] == bench.source_hash(bench.KERNEL_SOURCES[fam]), f"{key}: run tools/make_isa_mix.py"


def test_bench_roofline_bookkeeping_without_a_device(tmp_path, monkeypatch):
    """bench.valu_bound / attach_traffic: the dominant kernel is priced against the vector-issue roof from the committed
    instruction mix (profiles/isa_mix.json) with the SURVEY 8(d) HBM figure beside it, and profile entries measured on
    other kernel sources (hash mismatch) are dropped instead of being attached to a kernel that has changed."""
    import json
    import bench
    roof = {"bound": "hbm", "achieved": 7000.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.875, "traffic": None,
            "traffic_key": "bbb_1200_n256_b128_bf16", "plan": {"waves": 8}, "kernel": "k"}
    out = bench.valu_bound(dict(roof), 1200, 1200, 256, 128, 350.0, True)
    assert out["bound"] == "valu" and out["hbm_algorithmic"]["frac"] == 0.875 and out["peak"] == pytest.approx(78.6432)
    mix = json.load(open(os.path.join(REPO, "profiles", "isa_mix.json")))["bbb_fwd_gemm2_kernel<4,2,philox>"]
    if mix["source_hash"] == bench.source_hash(bench.KERNEL_SOURCES["bbb"]):      # the committed mix is current
        steps = 75 * 256 * 38                                                    # 16-feature tiles x pairs x k-steps
        assert out["valu"]["wave_ksteps_per_launch"] == steps
        want = steps * mix["valu_issue_cycles_per_iteration"] * 32.0 / 350e-6 / 1e12
        assert out["achieved"] == pytest.approx(want) and 0.0 < out["frac"] < 1.0
    else:
        assert out["achieved"] is None and "valu_note" in out
    # the algorithmic count (from the epsilon map, not from the compiled loop) sits beside the instruction-mix figure
    if out["achieved"] is not None:
        alg = out["valu"]["algorithmic"]
        assert alg["philox_rounds"] == 7 and alg["wave_instructions_per_weight"]["imul"] == pytest.approx(13 / 4)
        assert alg["floor_us_at_2p4GHz"] == pytest.approx(1200 * 1200 * 256 / 64 * alg["issue_cycles_per_weight"] / (1024 * 2.4e9) * 1e6)
        assert 0.0 < alg["frac"] < 1.0
    # K1s' bytes: (mu, rho) once per group of four samples + every sample's bf16 weights -- never above the HBM figure
    assert bench.sampling_bytes(4096, 4096, 4) == (8 + 2 * 4) * 4096 * 4096 + (8 + 4 * 4) * 4096
    assert bench.sampling_bytes(4096, 4096, 4) / 55.7e-6 / 1e9 / bench.HBM_PEAK_GBS < 1.0
    assert "../../include/bnn_hip.h" in bench.KERNEL_SOURCES["bbb"]         # BNN_PHILOX_ROUNDS is part of every family's hash
    # a traffic entry is attached only while its family's sources hash to the recorded value
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "bayesian-neural-network_amd" / "csrc")
    os.makedirs(tmp_path / "include")
    for f in bench.KERNEL_SOURCES["bbb"]:
        (tmp_path / "bayesian-neural-network_amd" / "csrc" / f).write_text("// " + f)
    good = bench.source_hash(bench.KERNEL_SOURCES["bbb"])
    for h, want in ((good, 123.0), ("0" * 16, None)):
        json.dump({"bbb_1200_n256_b128_bf16": {"hbm_bytes_per_launch": 123.0, "source_hash": h}}, open(tmp_path / "profiles" / "traffic.json", "w"))
        got = bench.attach_traffic({"traffic": None, "traffic_key": "bbb_1200_n256_b128_bf16"})
        assert got["traffic"] == want and ("traffic_note" in got) == (want is None)


def test_kernels_with_hand_issued_loads_keep_their_registers(tmp_path):
    """The block-GEMM / K-sliced kernels issue some of their loads in inline assembly and wait for them by hand (the
    compiler's own wait-count bookkeeping drains the prefetch: DESIGN 4).  The compiler does not know that the destination
    of such a load is busy until the wait: it may spill it (the round-3 memory fault: an address register overwritten by a
    late-landing load) or hand it to another value on a path that does not run through the wait statement.  Two checks
    over the gfx950 assembly of both sources (compiled here, no GPU needed):
      (1) tools/lint_hand_loads.py -- the invariant itself: between a hand-issued load with VGPR destinations and the
          s_waitcnt that retires it, no instruction reads or writes those registers (scratch traffic included), in EVERY
          kernel, K3b's 16-wave form (which carries 16 bytes of scratch for values of its epilogue) included;
      (2) the kernels whose hand-issued loads stay in flight across whole loop iterations compile without scratch."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import lint_hand_loads as lint
    # the lint detects what it is for: a touch of an in-flight destination, a spill of one, a counted wait that does not
    # reach the load yet -- and accepts the same code with the wait in place
    bad = tmp_path / "bad.s"
    bad.write_text("""_ZN3bnn4testEv:
\t;;#ASMSTART
\tds_read_b128 v[4:7], v1
\tds_read_b128 v[8:11], v1 offset:1024
\t;;#ASMEND
\tv_add_f32_e32 v20, v21, v22
\ts_waitcnt lgkmcnt(1)
\tv_mov_b32_e32 v30, v5
\tv_mov_b32_e32 v31, v9
\t;;#ASMSTART
\tglobal_load_dwordx4 v[12:15], v[2:3], off sc1
\t;;#ASMEND
\tscratch_store_dword off, v13, s0
\ts_waitcnt vmcnt(0) lgkmcnt(0)
\tv_mov_b32_e32 v32, v12
\ts_endpgm
""")
    v, seen = lint.check(str(bad))
    assert [(x[1], x[5]) for x in v] == [(9, ["v9"]), (13, ["v13"])] and seen == {"_ZN3bnn4testEv": 3}, v
    csrc = os.path.join(REPO, "bayesian-neural-network_amd", "csrc")
    watched = ("bbb_fwd_gemm_kernel", "bbb_fwd_gemm2_kernel", "bbb_fwd_gemm_rider_kernel", "bbb_block_gemm_kernel",
               "lr_fwd_kslice_kernel")
    seen = set()
    procs = []
    for src in ("bbb_linear.hip", "lr_linear.hip"):
        out = tmp_path / (src + ".s")
        procs.append((out, subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-S",
                                             "--offload-device-only", os.path.join(csrc, src), "-o", str(out)],
                                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    hand_kernels = {}
    for out, pr in procs:
        assert pr.wait() == 0
        violations, hand = lint.check(str(out))
        assert not violations, violations[:5]
        hand_kernels.update(hand)
        name = None
        for line in open(out):
            m = re.match(r"^(_ZN3bnn\S+):", line)
            if m:
                name = m.group(1)
            m = re.match(r"^; ScratchSize: (\d+)", line)
            if m and name:
                hit = [w for w in watched if re.search(r"\d+" + w + r"I", name)]
                if hit:
                    seen.add(hit[0])
                    assert int(m.group(1)) == 0, (name, int(m.group(1)))
                name = None
    assert seen == set(watched), seen
    # the lint saw the kernels it is meant for (their hand-issued loads carry register destinations)
    # (K1g's hand-issued loads are all LDS-DMA -- no register destination --, its LDS reads are the compiler's)
    for w in ("bbb_fwd_gemm_kernel", "bbb_fwd_gemm2_kernel", "bbb_fwd_gemm_rider_kernel", "lr_fwd_kslice_kernel", "lr_fwd_gemm_kernel"):
        assert any(re.search(r"\d+" + w + r"I", k) for k in hand_kernels), w
