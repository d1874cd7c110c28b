"""Thin tensor-level wrappers over the C ABI.  torch is used only for device memory and
the current HIP stream; every operation here is one or two kernels of libbnn_hip.so."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib as L
from ._lib import BnnHipError


@dataclass(frozen=True)
class PriorSpec:
    """Prior of a layer: Gaussian N(0, sigma_p) or the scale mixture of networks.py:14-27."""
    mixture: bool = False
    sigma_p: float = 1.0
    pi: float = 0.5
    sigma1: float = 1.0
    sigma2: float = 1.0

    @staticmethod
    def from_init(prior_init: Sequence[float], mixture: bool) -> "PriorSpec":
        if mixture:
            assert len(prior_init) == 3, "Scale Mixture Prior requires three values in prior initialisation"
            return PriorSpec(True, 1.0, float(prior_init[0]), math.exp(prior_init[1]), math.exp(prior_init[2]))
        assert len(prior_init) == 1, "Gaussian Prior requires one value in prior initialisation"
        return PriorSpec(False, float(prior_init[0]))

    def c(self) -> L.Prior:
        return L.Prior(L.PRIOR_MIXTURE if self.mixture else L.PRIOR_GAUSS, self.sigma_p, self.pi,
                       self.sigma1, self.sigma2)


def _stream() -> int:
    """Raw hipStream_t of the current stream of the current device.  (torch.cuda.current_stream() without a
    device walks through is_available() and an os.getenv on every call: ~20-60 us, more than a launch.)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def require_device(*tensors: torch.Tensor):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise BnnHipError(
                "bnn_hip: the Bayes-by-backprop hot path runs on a ROCm device only (tensor is on "
                f"{t.device}); there is no CPU fallback.  Move the module and its inputs to DEVICE.")


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise BnnHipError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise BnnHipError(f"activations must be float32 or bfloat16, got {t.dtype}")


def _exact_unless_bf16(math_mode: int) -> int:
    """The split-bf16 mode (MATH_BF16X3) exists for the BBB forward kernels; every other launch of a job in that mode --
    the local-reparameterisation layers, the backward kernels -- runs the exact-fp32 matrix core."""
    return L.MATH_F32 if math_mode == L.MATH_BF16X3 else math_mode


def _x3(x: torch.Tensor, n_samples: int):
    """Returns (contiguous x, batch, in_features, x_per_sample): 0 = one x for all samples, g >= 1 = sample s reads
    x[s // g] (x [rows, batch, in] with rows * g == n_samples: g = 1 is one x per sample, g = S is one x per minibatch
    of S MC samples)."""
    if x.dim() == 2:
        xs = x if x.is_contiguous() else x.contiguous()
        return xs, x.shape[0], x.shape[1], 0
    if x.dim() == 3:
        if x.shape[0] < 1 or n_samples % x.shape[0]:
            raise BnnHipError(f"x has {x.shape[0]} row blocks, which does not divide {n_samples} samples")
        xs = x if x.is_contiguous() else x.contiguous()
        return xs, x.shape[1], x.shape[2], n_samples // x.shape[0]
    raise BnnHipError(f"x must be [batch,in] or [rows,batch,in], got {tuple(x.shape)}")


def bbb_workspace(n_samples: int, out_features: int, device) -> torch.Tensor:
    nbytes = L.load().bnn_bbb_linear_fwd_workspace_bytes(n_samples, out_features)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


def split_scratch(n_samples: int, batch: int, out_features: int, device) -> torch.Tensor:
    """Scratch for the K-sliced GEMM form of K1: arrival counters (zeroed here, once; launches leave them zero) followed
    by the fp32 partial tiles (uninitialised).  One scratch serves one launch at a time."""
    lib = L.load()
    nbytes = lib.bnn_bbb_split_scratch_bytes(n_samples, batch, out_features)
    t = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=device)
    t[:lib.bnn_bbb_split_scratch_zero_bytes(n_samples, batch, out_features) // 4].zero_()
    return t


_split_cache = {}


def split_scratch_cached(n_samples: int, batch: int, out_features: int, device) -> torch.Tensor:
    """The eager path's scratch: one per (device, current stream, shape), allocated and zeroed once -- launches on one
    stream run one after the other and each leaves the counters at zero.  (Captured evaluators own theirs.)"""
    if torch.cuda.is_current_stream_capturing():
        return split_scratch(n_samples, batch, out_features, device)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, n_samples, batch, out_features)
    t = _split_cache.get(key)
    if t is None:
        if len(_split_cache) >= 16:
            _split_cache.clear()
        t = _split_cache[key] = split_scratch(n_samples, batch, out_features, device)
    return t


def final_scratch(n_samples: int, device) -> torch.Tensor:
    """Zeroed scratch for the fused last layer (K-range slices per sample)."""
    nbytes = L.load().bnn_bbb_final_scratch_bytes(n_samples)
    return torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=device)


def lr_split_scratch(n_samples: int, batch: int, out_features: int, device) -> torch.Tensor:
    """Scratch for the K-sliced form of K3 (1-3 samples on a wide LR layer): arrival counters (zeroed here, once; launches
    leave them zero) followed by the fp32 partial tiles.  One scratch serves one launch at a time."""
    lib = L.load()
    nbytes = lib.bnn_lr_split_scratch_bytes(n_samples, batch, out_features)
    t = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=device)
    t[:lib.bnn_lr_split_scratch_zero_bytes(n_samples, batch, out_features) // 4].zero_()
    return t


_lr_split_cache = {}


def lr_split_scratch_cached(n_samples: int, batch: int, out_features: int, device) -> torch.Tensor:
    """The eager path's K3s scratch: one per (device, current stream, shape), as split_scratch_cached."""
    if torch.cuda.is_current_stream_capturing():
        return lr_split_scratch(n_samples, batch, out_features, device)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, n_samples, batch, out_features)
    t = _lr_split_cache.get(key)
    if t is None:
        if len(_lr_split_cache) >= 16:
            _lr_split_cache.clear()
        t = _lr_split_cache[key] = lr_split_scratch(n_samples, batch, out_features, device)
    return t


def lr_workspace(out_features: int, device) -> torch.Tensor:
    nbytes = L.load().bnn_lr_linear_fwd_workspace_bytes(out_features)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


def _y16(a, y: torch.Tensor) -> torch.Tensor:
    """y_bf16_copy of a layer call: fp32 y for the backward, bf16 y for the next layer's forward (tile forms)."""
    if y.dtype != torch.float32:
        raise BnnHipError("want_y16 goes with fp32 y")
    y16 = torch.empty(tuple(y.shape), dtype=torch.bfloat16, device=y.device)
    a.y_bf16_copy = y16.data_ptr()
    return y16


def _bbb_build(x, w_mu, w_rho, b_mu, b_rho, *, n_samples: int, prior: PriorSpec, math_mode: int,
               relu: bool, y_dtype: torch.dtype, eps_mode: int, eps_w=None, eps_b=None, seed: int = 0,
               layer_id: int = 0, sample_offset: int = 0, want_stats: bool = True,
               want_scalars: bool = False, dump_eps: bool = False, workspace=None, sample_counter=None,
               out=None, split_scratch=None, w_sigma=None, form: int = 0, sample_group: int = 0,
               sample_group_stride: int = 0, w_sampled=None, b_sampled=None, rider=None, want_y16: bool = False, wt_out=None,
               x_lo=None, out_lo=None):
    """Argument block of K1 + the tensors it points at (kept alive by the caller).  `w_sampled` / `b_sampled` (bf16
    [S,out,in] / fp32 [S,out] from bbb_sample_weights): the matmul-only form, the parameter tensors may then be None.
    `rider` = the (args, results, keep) of build_sample_job: an independent sampling job carried by the launch."""
    require_device(x, w_mu, w_rho, b_mu, b_rho, eps_w, eps_b, w_sampled, b_sampled)
    if w_sampled is not None:
        if w_sampled.dtype != torch.bfloat16 or b_sampled is None or b_sampled.dtype != torch.float32 or \
                not w_sampled.is_contiguous() or not b_sampled.is_contiguous() or w_sampled.dim() != 3:
            raise BnnHipError("w_sampled must be contiguous bf16 [S,out,in], b_sampled contiguous fp32 [S,out]")
        if w_sampled.shape[0] != n_samples or b_sampled.numel() != n_samples * w_sampled.shape[1]:
            raise BnnHipError("w_sampled / b_sampled do not match n_samples")
        N, K = w_sampled.shape[1], w_sampled.shape[2]
        xs, B, Kx, per_sample = _x3(x, n_samples)
        if Kx != K:
            raise BnnHipError(f"shape mismatch: x[...,{Kx}] sampled weight {tuple(w_sampled.shape)}")
        y = out if out is not None else torch.empty((n_samples, B, N), dtype=y_dtype, device=xs.device)
        a = L.BbbFwdArgs()
        a.struct_bytes = C.sizeof(L.BbbFwdArgs)
        a.n_samples, a.batch, a.in_features, a.out_features = n_samples, B, K, N
        a.x, a.x_dtype, a.x_per_sample = xs.data_ptr(), _dt(xs), per_sample
        a.eps_mode, a.math = L.EPS_ZERO, L.MATH_BF16
        a.want_stats, a.relu = 0, int(relu)
        a.y, a.y_dtype = y.data_ptr(), _dt(y)
        a.w_sampled, a.b_sampled = w_sampled.data_ptr(), b_sampled.data_ptr()
        if wt_out is not None:                       # the same weights once more, transposed (for the layer's input gradient)
            require_device(wt_out)
            if wt_out.dtype != torch.bfloat16 or not wt_out.is_contiguous() or tuple(wt_out.shape) != (n_samples, K, N):
                raise BnnHipError("wt_out must be a contiguous bf16 [samples,in,out] tensor")
            a.w_sampled_t_out = wt_out.data_ptr()
        a.form = int(form)
        y16 = _y16(a, y) if want_y16 else None
        return a, dict(y=y, y16=y16, workspace=None, log_prior=None, log_q=None, eps_w=None, eps_b=None), (xs, w_sampled, b_sampled, y)
    w_mu, w_rho = _f32c(w_mu, "weight_mu"), _f32c(w_rho, "weight_rho")
    b_mu, b_rho = _f32c(b_mu, "bias_mu"), _f32c(b_rho, "bias_rho")
    N, K = w_mu.shape
    xs, B, Kx, per_sample = _x3(x, n_samples)
    if Kx != K or tuple(w_rho.shape) != (N, K) or tuple(b_mu.shape) != (N,) or tuple(b_rho.shape) != (N,):
        raise BnnHipError(f"shape mismatch: x[...,{Kx}] weight {tuple(w_mu.shape)} bias {tuple(b_mu.shape)}")
    dev = xs.device
    y = out if out is not None else torch.empty((n_samples, B, N), dtype=y_dtype, device=dev)
    if want_stats and workspace is None:
        workspace = bbb_workspace(n_samples, N, dev)
    lp = torch.empty(n_samples, dtype=torch.float32, device=dev) if want_scalars else None
    lq = torch.empty(n_samples, dtype=torch.float32, device=dev) if want_scalars else None
    if eps_mode == L.EPS_MEMORY:
        eps_w, eps_b = _f32c(eps_w, "eps_w"), _f32c(eps_b, "eps_b")
        if eps_w.numel() != n_samples * N * K or eps_b.numel() != n_samples * N:
            raise BnnHipError("eps_w/eps_b must be [samples,out,in] / [samples,out]")
    dw = torch.empty((n_samples, N, K), dtype=torch.float32, device=dev) if dump_eps else None
    db = torch.empty((n_samples, N), dtype=torch.float32, device=dev) if dump_eps else None
    a = L.BbbFwdArgs()
    a.struct_bytes = C.sizeof(L.BbbFwdArgs)
    a.n_samples, a.batch, a.in_features, a.out_features = n_samples, B, K, N
    a.x, a.x_dtype, a.x_per_sample = xs.data_ptr(), _dt(xs), per_sample
    a.w_mu, a.w_rho, a.b_mu, a.b_rho = w_mu.data_ptr(), w_rho.data_ptr(), b_mu.data_ptr(), b_rho.data_ptr()
    a.eps_mode, a.math = eps_mode, math_mode
    a.eps_w, a.eps_b = _ptr(eps_w) if eps_mode == L.EPS_MEMORY else None, _ptr(eps_b) if eps_mode == L.EPS_MEMORY else None
    a.seed, a.layer_id, a.sample_offset = seed & 0xFFFFFFFFFFFFFFFF, layer_id, sample_offset & 0xFFFFFFFF
    a.sample_counter = _ptr(sample_counter)
    a.sample_group, a.sample_group_stride = int(sample_group), int(sample_group_stride)
    a.form = int(form)
    a.eps_w_dump, a.eps_b_dump = _ptr(dw), _ptr(db)
    a.prior = prior.c()
    a.want_stats, a.relu = int(want_stats), int(relu)
    a.workspace = _ptr(workspace) if want_stats else None
    a.workspace_bytes = workspace.numel() * 4 if (want_stats and workspace is not None) else 0
    a.log_prior, a.log_q = _ptr(lp), _ptr(lq)
    a.y, a.y_dtype = y.data_ptr(), _dt(y)
    y_lo = None
    if math_mode == L.MATH_BF16X3:
        # split-bf16 math: a bf16 activation is a PAIR of planes (hi = x / y, lo = x_lo / out_lo), fp32 ones are split on chip
        if xs.dtype == torch.bfloat16:
            if x_lo is None or x_lo.dtype != torch.bfloat16 or tuple(x_lo.shape) != tuple(x.shape) or not x_lo.is_contiguous():
                raise BnnHipError("bf16x3 math on bf16 x needs x_lo: the contiguous bf16 low plane, shaped like x")
            require_device(x_lo)
            a.x_lo = x_lo.data_ptr()
        if y.dtype == torch.bfloat16:
            y_lo = out_lo if out_lo is not None else torch.empty(tuple(y.shape), dtype=torch.bfloat16, device=dev)
            if y_lo.dtype != torch.bfloat16 or y_lo.numel() != y.numel() or not y_lo.is_contiguous():
                raise BnnHipError("out_lo must be a contiguous bf16 tensor shaped like y")
            a.y_lo = y_lo.data_ptr()
    if w_sigma is not None:
        a.w_sigma = w_sigma.data_ptr()
    if split_scratch is not None:
        a.split_scratch = split_scratch.data_ptr()
        a.split_scratch_bytes = split_scratch.numel() * split_scratch.element_size()
    if rider is not None:
        a.rider = C.addressof(rider[0])
    res = dict(y=y, y16=_y16(a, y) if want_y16 else None, workspace=workspace, log_prior=lp, log_q=lq, eps_w=dw, eps_b=db, y_lo=y_lo)
    keep = (xs, w_mu, w_rho, b_mu, b_rho, eps_w, eps_b, sample_counter, split_scratch, w_sigma, rider, x_lo, y_lo)
    a._keep = keep                    # (the structure owns what its pointers refer to: engine.GraphedElbo(capture="calls") replays it)
    return a, res, keep


def bbb_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw):
    """K1.  Returns dict(y, workspace, log_prior, log_q, eps_w, eps_b)."""
    lib = L.load()
    a, res, keep = _bbb_build(x, w_mu, w_rho, b_mu, b_rho, **kw)
    L.check(lib.bnn_bbb_linear_fwd(C.byref(a), _stream()), "bnn_bbb_linear_fwd")
    return res


def _plan_dict(pl: L.Plan) -> dict:
    return {k: getattr(pl, k) for k, _ in L.Plan._fields_}


def bbb_plan(x, w_mu, w_rho, b_mu, b_rho, **kw) -> dict:
    """bnn_bbb_plan: the launch geometry bbb_linear_fwd would use for these arguments (no launch)."""
    a, res, keep = _bbb_build(x, w_mu, w_rho, b_mu, b_rho, **kw)
    pl = L.Plan()
    L.check(L.load().bnn_bbb_plan(C.byref(a), C.byref(pl)), "bnn_bbb_plan")
    return _plan_dict(pl)


def sample_workspace(n_samples: int, fin: int, fout: int, device) -> torch.Tensor:
    """Statistics workspace of one layer that serves both K1 (fused) and K1s (split) forms."""
    nbytes = L.load().bnn_bbb_sample_workspace_bytes(n_samples, fin, fout) or \
        L.load().bnn_bbb_linear_fwd_workspace_bytes(n_samples, fout)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


def build_sample_job(layers, *, n_samples: int, seed: int = 0, sample_offset: int = 0, sample_counter=None, cast=None,
                     sample_group: int = 0, sample_group_stride: int = 0):
    """Argument block of K1s (bnn_bbb_sample_weights) + its results + the tensors it points at: launched by
    bbb_sample_weights, or handed to a layer launch as its `rider`.  `layers` = list of dicts(w_mu [out,in], w_rho,
    b_mu, b_rho, prior, layer_id, workspace=None, w_out=None, b_out=None); results = list of dicts(w [S,out,in] bf16,
    b [S,out] fp32, workspace).  `cast` = (fp32 tensor, bf16 tensor): also converts the input batch in that launch."""
    if not 1 <= len(layers) <= L.SAMPLE_MAX_LAYERS:
        raise BnnHipError(f"bbb_sample_weights: 1..{L.SAMPLE_MAX_LAYERS} layers per launch")
    a = L.SampleArgs()
    a.struct_bytes = C.sizeof(L.SampleArgs)
    a.n_layers, a.n_samples = len(layers), int(n_samples)
    a.seed, a.sample_offset = seed & 0xFFFFFFFFFFFFFFFF, sample_offset & 0xFFFFFFFF
    a.sample_counter = _ptr(sample_counter)
    a.sample_group, a.sample_group_stride = int(sample_group), int(sample_group_stride)
    res, keep = [], [sample_counter]
    for i, ly in enumerate(layers):
        w_mu, w_rho = _f32c(ly["w_mu"], "weight_mu"), _f32c(ly["w_rho"], "weight_rho")
        b_mu, b_rho = _f32c(ly["b_mu"], "bias_mu"), _f32c(ly["b_rho"], "bias_rho")
        require_device(w_mu, w_rho, b_mu, b_rho)
        N, K = w_mu.shape
        if K % 8:
            raise BnnHipError("bbb_sample_weights: in_features must be a multiple of 8")
        if tuple(w_rho.shape) != (N, K) or tuple(b_mu.shape) != (N,) or tuple(b_rho.shape) != (N,):
            raise BnnHipError("bbb_sample_weights: parameter shapes disagree")
        dev = w_mu.device
        ws = ly.get("workspace")
        if ws is None:
            ws = sample_workspace(n_samples, K, N, dev)
        w = ly.get("w_out")
        if w is None:
            w = torch.empty((n_samples, N, K), dtype=torch.bfloat16, device=dev)
        b = ly.get("b_out")
        if b is None:
            b = torch.empty((n_samples, N), dtype=torch.float32, device=dev)
        if w.dtype != torch.bfloat16 or w.numel() != n_samples * N * K or b.dtype != torch.float32 or b.numel() != n_samples * N:
            raise BnnHipError("bbb_sample_weights: w_out must be bf16 [S,out,in], b_out fp32 [S,out]")
        e = a.layer[i]
        e.in_features, e.out_features, e.layer_id = K, N, int(ly.get("layer_id", i))
        e.w_mu, e.w_rho, e.b_mu, e.b_rho = w_mu.data_ptr(), w_rho.data_ptr(), b_mu.data_ptr(), b_rho.data_ptr()
        e.w_out, e.b_out = w.data_ptr(), b.data_ptr()
        e.workspace, e.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        e.prior = ly["prior"].c()
        res.append(dict(w=w, b=b, workspace=ws))
        keep.append((w_mu, w_rho, b_mu, b_rho, w, b, ws))
    if cast is not None:                      # (fp32 src, bf16 dst): the input batch cast rides on the launch
        src, dst = cast
        require_device(src, dst)
        if src.dtype != torch.float32 or dst.dtype != torch.bfloat16 or src.numel() != dst.numel() or \
                not src.is_contiguous() or not dst.is_contiguous():
            raise BnnHipError("bbb_sample_weights: cast = (contiguous fp32 source, contiguous bf16 destination) of one size")
        a.cast_src, a.cast_dst, a.cast_n = src.data_ptr(), dst.data_ptr(), src.numel()
        keep.append((src, dst))
    a._keep = keep                    # (the structure owns what its pointers refer to: engine.GraphedElbo(capture="calls") replays it)
    return a, res, keep


def bbb_sample_weights(layers, **kw):
    """K1s (bnn_bbb_sample_weights): one launch samples every layer of `layers`; see build_sample_job."""
    a, res, keep = build_sample_job(layers, **kw)
    L.check(L.load().bnn_bbb_sample_weights(C.byref(a), _stream()), "bnn_bbb_sample_weights")
    return res


def bbb_sampled_matmul(x, w, b, *, n_samples: int, relu: bool, y_dtype: torch.dtype, out=None, want_y16: bool = False, wt_out=None,
                       form: int = 0):
    """Matmul half of K1 over weights sampled by bbb_sample_weights: y[s] = act(x[s] . w[s]^T + b[s]).  `want_y16`:
    returns (y fp32, y in bf16).  `wt_out` (bf16 [S,in,out]): the launch also leaves w transposed there."""
    a, res, keep = _bbb_build(x, None, None, None, None, n_samples=n_samples, prior=PriorSpec(), math_mode=L.MATH_BF16, relu=relu,
                              y_dtype=y_dtype, eps_mode=L.EPS_ZERO, want_stats=False, out=out, w_sampled=w, b_sampled=b,
                              want_y16=want_y16, wt_out=wt_out, form=form)
    L.check(L.load().bnn_bbb_linear_fwd(C.byref(a), _stream()), "bnn_bbb_linear_fwd")
    return (res["y"], res["y16"]) if want_y16 else res["y"]


def _lr_build(x, w_mu, w_rho, b_mu, b_rho, *, n_samples: int, sigma_p: float, math_mode: int, relu: bool,
                  y_dtype: torch.dtype, eps_mode: int, eps_act=None, eps_b=None, seed: int = 0, layer_id: int = 0,
                  sample_offset: int = 0, want_kl: bool = True, want_scalars: bool = False,
                  dump_eps: bool = False, workspace=None, sample_counter=None, out=None, x_sq=None,
                  out_sq=None, w_frag=None, want_v: bool = False, want_y16: bool = False, want_hfac: bool = False, form: int = 0,
                  sample_group: int = 0,
                  sample_group_stride: int = 0, split_scratch=None, rider=None, x_lo=None, out_lo=None):
    """Argument block of K3 + the result dict + the tensors it points at.  `rider`: dict(w_mu, w_rho, b_mu, b_rho, w_frag,
    workspace) of a narrow LR layer whose operands this launch prepares on the side (bnn_lr_rider).
    math_mode MATH_BF16X3 (the block form over lr_prepare(x3=True) fragments, bf16 x with `x_lo` and `x_sq`): taken as asked when
    those operands are there, else the launch runs exact fp32 (the mode's promise is the reference's arithmetic)."""
    require_device(x, w_mu, w_rho, b_mu, b_rho, eps_act, eps_b)
    if math_mode == L.MATH_BF16X3 and (x_lo is None or x_sq is None or w_frag is None or x.dtype != torch.bfloat16):
        math_mode = L.MATH_F32
    w_mu, w_rho = _f32c(w_mu, "weight_mu"), _f32c(w_rho, "weight_rho")
    b_mu, b_rho = _f32c(b_mu, "bias_mu"), _f32c(b_rho, "bias_rho")
    K, N = w_mu.shape
    xs, B, Kx, per_sample = _x3(x, n_samples)
    if Kx != K or tuple(w_rho.shape) != (K, N) or tuple(b_mu.shape) != (N,) or tuple(b_rho.shape) != (N,):
        raise BnnHipError(f"shape mismatch: x[...,{Kx}] weight {tuple(w_mu.shape)} bias {tuple(b_mu.shape)}")
    dev = xs.device
    y = out if out is not None else torch.empty((n_samples, B, N), dtype=y_dtype, device=dev)
    if want_kl and workspace is None:
        workspace = lr_workspace(N, dev)
    kl3 = torch.empty(3, dtype=torch.float32, device=dev) if want_scalars else None
    if eps_mode == L.EPS_MEMORY:
        eps_act, eps_b = _f32c(eps_act, "eps_act"), _f32c(eps_b, "eps_b")
        if eps_act.numel() != n_samples * B * N or eps_b.numel() != n_samples * N:
            raise BnnHipError("eps_act/eps_b must be [samples,batch,out] / [samples,out]")
    da = torch.empty((n_samples, B, N), dtype=torch.float32, device=dev) if dump_eps else None
    db = torch.empty((n_samples, N), dtype=torch.float32, device=dev) if dump_eps else None
    a = L.LrFwdArgs()
    a.struct_bytes = C.sizeof(L.LrFwdArgs)
    a.n_samples, a.batch, a.in_features, a.out_features = n_samples, B, K, N
    a.x, a.x_dtype, a.x_per_sample = xs.data_ptr(), _dt(xs), per_sample
    a.w_mu, a.w_rho, a.b_mu, a.b_rho = w_mu.data_ptr(), w_rho.data_ptr(), b_mu.data_ptr(), b_rho.data_ptr()
    a.eps_mode, a.math = eps_mode, math_mode
    a.eps_act = _ptr(eps_act) if eps_mode == L.EPS_MEMORY else None
    a.eps_b = _ptr(eps_b) if eps_mode == L.EPS_MEMORY else None
    a.seed, a.layer_id, a.sample_offset = seed & 0xFFFFFFFFFFFFFFFF, layer_id, sample_offset & 0xFFFFFFFF
    a.sample_counter = _ptr(sample_counter)
    a.sample_group, a.sample_group_stride = int(sample_group), int(sample_group_stride)
    a.form = int(form)
    a.eps_act_dump, a.eps_b_dump = _ptr(da), _ptr(db)
    a.sigma_p, a.want_kl, a.relu = float(sigma_p), int(want_kl), int(relu)
    a.workspace = _ptr(workspace) if want_kl else None
    a.workspace_bytes = workspace.numel() * 4 if (want_kl and workspace is not None) else 0
    a.kl_out = _ptr(kl3)
    a.y, a.y_dtype = y.data_ptr(), _dt(y)
    if x_sq is not None:
        if x_sq.dtype != torch.bfloat16 or tuple(x_sq.shape) != tuple(xs.shape) or not x_sq.is_contiguous():
            raise BnnHipError("x_sq must be a contiguous bfloat16 tensor shaped like x")
        a.x_sq = x_sq.data_ptr()
    if w_frag is not None:
        a.w_frag = w_frag.data_ptr()
    if out_sq is not None:
        if out_sq.dtype != torch.bfloat16 or out_sq.numel() != y.numel():
            raise BnnHipError("out_sq must be bfloat16 shaped like y")
        a.y_sq = out_sq.data_ptr()
    v = None
    if want_v:                                       # the variance the kernel sampled from: saved for bnn_lr_linear_bwd
        v = torch.empty(tuple(y.shape), dtype=torch.float32, device=y.device)
        a.v_out = v.data_ptr()
    hfac = None
    if want_hfac:                                    # eps_act / (2 sqrt(v)): lets bnn_lr_linear_bwd skip its preparation launch
        hfac = torch.empty(tuple(y.shape), dtype=torch.float32, device=y.device)
        a.hfac_out = hfac.data_ptr()
    y16 = None
    if want_y16:                                     # fp32 y for the backward, bf16 y for the next layer's forward
        if y.dtype != torch.float32:
            raise BnnHipError("want_y16 goes with fp32 y")
        y16 = torch.empty(tuple(y.shape), dtype=torch.bfloat16, device=y.device)
        a.y_bf16_copy = y16.data_ptr()
    if split_scratch is not None:
        a.split_scratch = split_scratch.data_ptr()
        a.split_scratch_bytes = split_scratch.numel() * split_scratch.element_size()
    y_lo = None
    if math_mode == L.MATH_BF16X3:
        if x_lo.dtype != torch.bfloat16 or tuple(x_lo.shape) != tuple(xs.shape) or not x_lo.is_contiguous():
            raise BnnHipError("x_lo must be a contiguous bfloat16 tensor shaped like x")
        require_device(x_lo)
        a.x_lo = x_lo.data_ptr()
        if y.dtype == torch.bfloat16:
            y_lo = out_lo if out_lo is not None else torch.empty(tuple(y.shape), dtype=torch.bfloat16, device=dev)
            if y_lo.dtype != torch.bfloat16 or y_lo.numel() != y.numel() or not y_lo.is_contiguous():
                raise BnnHipError("out_lo must be a contiguous bf16 tensor shaped like y")
            a.y_lo = y_lo.data_ptr()
    rd = None
    if rider is not None:
        rw = [_f32c(rider[k], k) for k in ("w_mu", "w_rho", "b_mu", "b_rho")]
        require_device(*rw, rider["w_frag"], rider["workspace"])
        rd = L.LrRider()
        rd.struct_bytes = C.sizeof(L.LrRider)
        rd.in_features, rd.out_features = int(rw[0].shape[0]), int(rw[0].shape[1])
        rd.w_mu, rd.w_rho, rd.b_mu, rd.b_rho = (t.data_ptr() for t in rw)
        rd.w_frag, rd.w_frag_bytes = rider["w_frag"].data_ptr(), rider["w_frag"].numel() * rider["w_frag"].element_size()
        rd.kl_workspace = rider["workspace"].data_ptr()
        rd.kl_workspace_bytes = rider["workspace"].numel() * rider["workspace"].element_size()
        a.rider = C.pointer(rd)
        rd = (rd, rw, rider["w_frag"], rider["workspace"])
    res = dict(y=y, y_sq=out_sq, workspace=workspace, kl3=kl3, eps_act=da, eps_b=db, v=v, y16=y16, hfac=hfac, y_lo=y_lo)
    keep = (xs, w_mu, w_rho, b_mu, b_rho, eps_act, eps_b, sample_counter, x_sq, w_frag, out_sq, workspace, split_scratch, rd, x_lo, y_lo)
    a._keep = keep                    # (the structure owns what its pointers refer to: engine.GraphedElbo(capture="calls") replays it)
    return a, res, keep


def lr_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, **kw):
    """K3.  Weights are [in, out].  Returns dict(y, workspace, kl3, eps_act, eps_b)."""
    lib = L.load()
    a, res, keep = _lr_build(x, w_mu, w_rho, b_mu, b_rho, **kw)
    L.check(lib.bnn_lr_linear_fwd(C.byref(a), _stream()), "bnn_lr_linear_fwd")
    return res


def lr_final_fwd(layer_args: tuple, layer_kw: dict, fin_kw: dict):
    """Last LR layer + ELBO finalize through bnn_lr_final_fwd (one launch when the layer is narrow and the evaluation
    has few samples, else the two launches).  `fin_kw["workspaces"]` lists the KL workspaces of ALL layers (the last
    one's is filled only by the two-launch form); `logits` is taken from the layer call.
    Returns (layer result dict, finalize result dict)."""
    lib = L.load()
    a, res, keep1 = _lr_build(*layer_args, **layer_kw)
    fin_kw = dict(fin_kw)
    fin_kw["logits"] = res["y"]
    f, out, keep2 = _fin_build(**fin_kw)
    L.check(lib.bnn_lr_final_fwd(C.byref(a), C.byref(f), _stream()), "bnn_lr_final_fwd")
    return res, out


def lr_plan(x, w_mu, w_rho, b_mu, b_rho, **kw) -> dict:
    """bnn_lr_plan: the launch geometry lr_linear_fwd would use for these arguments (no launch)."""
    a, res, keep = _lr_build(x, w_mu, w_rho, b_mu, b_rho, **kw)
    pl = L.Plan()
    L.check(L.load().bnn_lr_plan(C.byref(a), C.byref(pl)), "bnn_lr_plan")
    return _plan_dict(pl)


def gauss_kl(mu: torch.Tensor, rho: torch.Tensor, sigma_p: float) -> torch.Tensor:
    """K2.  Returns float32[4] = (KL, sum log sigma, sum sigma^2, sum mu^2) on the device."""
    lib = L.load()
    require_device(mu, rho)
    mu, rho = _f32c(mu, "mu"), _f32c(rho, "rho")
    if mu.numel() != rho.numel():
        raise BnnHipError("mu and rho must have the same number of elements")
    n = mu.numel()
    ws = torch.empty(lib.bnn_gauss_kl_workspace_bytes(n) // 4, dtype=torch.float32, device=mu.device)
    out = torch.empty(4, dtype=torch.float32, device=mu.device)
    L.check(lib.bnn_gauss_kl(mu.data_ptr(), rho.data_ptr(), n, float(sigma_p), ws.data_ptr(), ws.numel() * 4,
                             out.data_ptr(), _stream()), "bnn_gauss_kl")
    return out


def _fin_build(*, workspaces, layer_in, layer_out, local_reparam: bool, prior: PriorSpec, n_samples: int,
               logits: Optional[torch.Tensor], target: Optional[torch.Tensor], mode: Optional[str],
               nll_sigma: float = 1.0, sample_counter=None, sample_counter_inc: int = 0, out=None, sums=None,
               ticket=None, scratch=None, group_samples: int = 0, loss=None):
    """`group_samples` g > 0: the n_samples are G = n_samples / g independent minibatches of g MC samples each;
    `target` may then hold one target block per minibatch ([G, batch] / [G, batch, classes]) and `sums` is [G, 4].
    `loss` = dict(beta=device scalar, total_samples=, grad_scale=): the training step's tail (bnn_loss_args); its
    results come back as out["loss"] = (out4, g_a, g_b, g_kl3, g_logits)."""
    n_layers = len(workspaces)
    dev = logits.device if logits is not None else workspaces[0].device
    a = L.FinalizeArgs()
    a.struct_bytes = C.sizeof(L.FinalizeArgs)
    a.n_layers, a.local_reparam, a.n_samples = n_layers, int(local_reparam), n_samples
    for i, (w, ki, ko) in enumerate(zip(workspaces, layer_in, layer_out)):
        require_device(w)
        a.layer_workspace[i] = w.data_ptr()
        a.layer_in[i], a.layer_out[i] = ki, ko
    a.prior = prior.c()
    out = {**dict(log_prior=None, log_q=None, kl=None, nll=None), **(out or {})}
    preset = {k for k, v in out.items() if v is not None}
    if n_layers:
        for key in (("kl",) if local_reparam else ("log_prior", "log_q")):
            if key not in preset:
                out[key] = torch.empty(n_samples, dtype=torch.float32, device=dev)
    keep = []
    if logits is not None:
        require_device(logits, target)
        lg = _f32c(logits, "logits")
        S, B, Cc = lg.shape
        if S != n_samples:
            raise BnnHipError("logits must be [samples,batch,classes]")
        G = n_samples // group_samples if group_samples else 1
        if group_samples and n_samples % group_samples:
            raise BnnHipError("group_samples must divide n_samples")
        if mode == "classification":
            tg = target.to(torch.int64).contiguous()
            if tg.numel() not in (B, G * B):
                raise BnnHipError("classification target must have `batch` elements (per minibatch)")
            a.nll_mode = L.NLL_CLASSIFICATION
            per_group = tg.numel() == G * B and G > 1
        elif mode == "regression":
            tg = _f32c(target.to(torch.float32), "target")
            if tg.numel() not in (B * Cc, G * B * Cc):
                raise BnnHipError("regression target must match the output shape (per minibatch)")
            a.nll_mode = L.NLL_REGRESSION
            per_group = tg.numel() == G * B * Cc and G > 1
        else:
            raise Exception("Training mode must be either 'regression' or 'classification'")
        a.target_per_group = int(per_group)
        keep += [lg, tg]
        a.batch, a.classes = B, Cc
        a.logits, a.target, a.nll_sigma = lg.data_ptr(), tg.data_ptr(), float(nll_sigma)
        if "nll" not in preset:
            out["nll"] = torch.empty(n_samples, dtype=torch.float32, device=dev)
    a.sample_counter, a.sample_counter_inc = _ptr(sample_counter), int(sample_counter_inc)
    a.sums = _ptr(sums)
    a.group_samples = int(group_samples)
    a.ticket = _ptr(ticket)
    a.scratch = _ptr(scratch)
    a.scratch_bytes = scratch.numel() * scratch.element_size() if scratch is not None else 0
    a.log_prior, a.log_q, a.kl, a.nll = _ptr(out["log_prior"]), _ptr(out["log_q"]), _ptr(out["kl"]), _ptr(out["nll"])
    keep += [sample_counter, sums, ticket, scratch] + list(workspaces)
    if loss is not None:
        if logits is None or group_samples:
            raise BnnHipError("the loss tail needs the logits of ONE evaluation")
        require_device(loss["beta"])
        la = L.LossArgs()
        res = (torch.empty(4, dtype=torch.float32, device=dev), torch.empty(n_samples, dtype=torch.float32, device=dev),
               torch.empty(n_samples, dtype=torch.float32, device=dev), torch.empty(3, dtype=torch.float32, device=dev),
               torch.empty_like(lg))
        la.beta, la.total_samples, la.grad_scale = loss["beta"].data_ptr(), float(loss["total_samples"]), float(loss.get("grad_scale", 1.0))
        la.out4, la.g_a, la.g_b, la.g_kl3, la.g_logits = (t.data_ptr() for t in res)
        a.loss = C.pointer(la)
        out["loss"] = res
        keep += [la, loss["beta"]] + list(res)
    a._keep = keep
    return a, out, keep


def elbo_finalize(**kw):
    """K4.  Returns dict(log_prior, log_q, kl, nll): float32[n_samples] tensors (or None)."""
    lib = L.load()
    a, out, keep = _fin_build(**kw)
    L.check(lib.bnn_elbo_finalize(C.byref(a), _stream()), "bnn_elbo_finalize")
    return out


def bbb_final_fwd(layer_args: tuple, layer_kw: dict, fin_kw: dict):
    """Last BBB layer + ELBO finalize through bnn_bbb_final_fwd (one launch when the layer is a
    single feature tile).  `fin_kw` must not carry `logits`/`workspaces` for the last layer:
    they are taken from the layer call.  Returns (layer result dict, finalize result dict)."""
    lib = L.load()
    a, res, keep1 = _bbb_build(*layer_args, **layer_kw)
    fin_kw = dict(fin_kw)
    if layer_kw.get("w_sampled") is None:              # (a pre-sampled layer: its statistics workspace, the sampler's, is
        fin_kw["workspaces"] = list(fin_kw["workspaces"]) + [res["workspace"]]     # already the last of fin_kw's)
    fin_kw["logits"] = res["y"]
    f, out, keep2 = _fin_build(**fin_kw)
    L.check(lib.bnn_bbb_final_fwd(C.byref(a), C.byref(f), _stream()), "bnn_bbb_final_fwd")
    return res, out


def philox_normal(seed: int, tensor_id: int, sample_offset: int, n_samples: int, rows: int, cols: int,
                  device) -> torch.Tensor:
    """The on-chip epsilon stream, materialised: float32[n_samples, rows, cols]."""
    lib = L.load()
    if torch.device(device).type != "cuda":
        raise BnnHipError("bnn_hip.philox_normal needs a ROCm device")
    eps = torch.empty((n_samples, rows, cols), dtype=torch.float32, device=device)
    L.check(lib.bnn_philox_normal(eps.data_ptr(), seed & 0xFFFFFFFFFFFFFFFF, tensor_id, sample_offset & 0xFFFFFFFF,
                                  n_samples, rows, cols, _stream()), "bnn_philox_normal")
    return eps


def cast_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None, out_sq: Optional[torch.Tensor] = None,
              want_sq: bool = False):
    """fp32 -> bf16 copy of a contiguous device tensor (one tiny kernel); with `want_sq` also
    x*x in bf16.  Returns out, or (out, out_sq)."""
    lib = L.load()
    require_device(x)
    x = _f32c(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    if want_sq and out_sq is None:
        out_sq = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.check(lib.bnn_cast_bf16(x.data_ptr(), out.data_ptr(), _ptr(out_sq), x.numel(), _stream()), "bnn_cast_bf16")
    return (out, out_sq) if (want_sq or out_sq is not None) else out


def lr_prepare(w_mu, w_rho, b_mu, b_rho, workspace=None, out=None, x3: bool = False):
    """bnn_lr_prepare: bf16 (M, sigma^2) in MFMA fragment order + the KL sums into `workspace` (`x3`: bnn_lr_prepare_x3 --
    the fragments of the split-bf16 math mode, with the low part of M as a third plane).  Returns (w_frag, workspace)."""
    lib = L.load()
    require_device(w_mu, w_rho, b_mu, b_rho)
    w_mu, w_rho = _f32c(w_mu, "weight_mu"), _f32c(w_rho, "weight_rho")
    b_mu, b_rho = _f32c(b_mu, "bias_mu"), _f32c(b_rho, "bias_rho")
    K, N = w_mu.shape
    nbytes = (lib.bnn_lr_prepare_x3_bytes if x3 else lib.bnn_lr_prepare_bytes)(K, N)
    if out is None:
        out = torch.empty(nbytes // 4, dtype=torch.float32, device=w_mu.device)
    if workspace is None:
        workspace = lr_workspace(N, w_mu.device)
    fn = lib.bnn_lr_prepare_x3 if x3 else lib.bnn_lr_prepare
    L.check(fn(w_mu.data_ptr(), w_rho.data_ptr(), b_mu.data_ptr(), b_rho.data_ptr(), K, N,
               out.data_ptr(), out.numel() * 4, workspace.data_ptr(), workspace.numel() * 4, _stream()),
            "bnn_lr_prepare")
    return out, workspace


def lr_prepare_many(jobs: Sequence[dict], x3: bool = False):
    """bnn_lr_prepare_many: the prepared operands of several layers in ONE launch.  Each job: dict(w_mu, w_rho, b_mu, b_rho,
    workspace, out) with `out` / `workspace` as in lr_prepare (allocated when absent).  Returns [(w_frag, workspace), ...]."""
    lib = L.load()
    if not 1 <= len(jobs) <= L.PREPARE_MANY_MAX:
        raise BnnHipError(f"lr_prepare_many: 1 .. {L.PREPARE_MANY_MAX} layers per launch")
    arr = (L.LrPrepareJob * len(jobs))()
    keep, res = [], []
    for j, q in enumerate(jobs):
        require_device(q["w_mu"], q["w_rho"], q["b_mu"], q["b_rho"])
        w_mu, w_rho = _f32c(q["w_mu"], "weight_mu"), _f32c(q["w_rho"], "weight_rho")
        b_mu, b_rho = _f32c(q["b_mu"], "bias_mu"), _f32c(q["b_rho"], "bias_rho")
        K, N = w_mu.shape
        nbytes = (lib.bnn_lr_prepare_x3_bytes if x3 else lib.bnn_lr_prepare_bytes)(K, N)
        out = q.get("out")
        if out is None:
            out = torch.empty(nbytes // 4, dtype=torch.float32, device=w_mu.device)
        ws = q.get("workspace")
        if ws is None:
            ws = lr_workspace(N, w_mu.device)
        keep += [w_mu, w_rho, b_mu, b_rho, out, ws]
        a = arr[j]
        a.w_mu, a.w_rho, a.b_mu, a.b_rho = w_mu.data_ptr(), w_rho.data_ptr(), b_mu.data_ptr(), b_rho.data_ptr()
        a.in_features, a.out_features = K, N
        a.w_frag, a.w_frag_bytes = out.data_ptr(), out.numel() * 4
        a.kl_workspace, a.kl_workspace_bytes = ws.data_ptr(), ws.numel() * 4
        res.append((out, ws))
    L.check(lib.bnn_lr_prepare_many(arr, len(jobs), int(bool(x3)), _stream()), "bnn_lr_prepare_many")
    return res


def _grad_outputs(out, w_mu, w_rho, b_mu, b_rho):
    """Gradient destinations: fresh tensors, or the caller's (e.g. views of one flat all-reduce bucket)."""
    if out is None:
        return torch.empty_like(w_mu), torch.empty_like(w_rho), torch.empty_like(b_mu), torch.empty_like(b_rho)
    for o, p_ in zip(out, (w_mu, w_rho, b_mu, b_rho)):
        if o.dtype != torch.float32 or tuple(o.shape) != tuple(p_.shape) or not o.is_contiguous() or o.device != p_.device:
            raise BnnHipError("gradient outputs must be contiguous float32 tensors shaped like their parameters")
    return tuple(out)


def bbb_linear_bwd(x, gy, y, w_mu, w_rho, b_mu, b_rho, *, n_samples: int, prior: PriorSpec, math_mode: int, relu: bool,
                   eps_mode: int, eps_w=None, eps_b=None, seed: int = 0, layer_id: int = 0, sample_offset: int = 0,
                   g_log_prior=None, g_log_q=None, want_gx: bool = True, sample_counter=None, out=None,
                   gx_relu_mask: bool = False, w_sampled=None, w_sampled_t=None, gy16=None, want_gx16: bool = False):
    """F1: backward of K1 (bnn_bbb_linear_bwd).  All tensors fp32.  Returns
    (g_w_mu, g_w_rho, g_b_mu, g_b_rho, g_x[S,B,K] | None), with `want_gx16` also g_x in bf16 as a sixth element.
    `w_sampled_t` (bf16 [S,in,out], the forward's wt_out) / `gy16` (bf16 copy of gy): the input gradient as the
    forward's matmul-only launch."""
    lib = L.load()
    require_device(x, gy, y, w_mu, w_rho, b_mu, b_rho, eps_w, eps_b, g_log_prior, g_log_q)
    w_mu, w_rho = _f32c(w_mu, "weight_mu"), _f32c(w_rho, "weight_rho")
    b_mu, b_rho = _f32c(b_mu, "bias_mu"), _f32c(b_rho, "bias_rho")
    N, K = w_mu.shape
    xs, B, Kx, per_sample = _x3(_f32c(x, "x"), n_samples)
    gy = _f32c(gy, "gy")
    if Kx != K or gy.numel() != n_samples * B * N:
        raise BnnHipError("bbb_linear_bwd: shape mismatch")
    dev = xs.device
    a = L.BbbBwdArgs()
    a.struct_bytes = C.sizeof(L.BbbBwdArgs)
    a.n_samples, a.batch, a.in_features, a.out_features = n_samples, B, K, N
    a.x, a.x_per_sample, a.relu = xs.data_ptr(), per_sample, int(relu)
    a.gy = gy.data_ptr()
    if relu:
        y = _f32c(y, "y")
        a.y = y.data_ptr()
    a.w_mu, a.w_rho, a.b_mu, a.b_rho = w_mu.data_ptr(), w_rho.data_ptr(), b_mu.data_ptr(), b_rho.data_ptr()
    a.eps_mode, a.math = eps_mode, _exact_unless_bf16(math_mode)
    if eps_mode == L.EPS_MEMORY:
        eps_w, eps_b = _f32c(eps_w, "eps_w"), _f32c(eps_b, "eps_b")
        a.eps_w, a.eps_b = eps_w.data_ptr(), eps_b.data_ptr()
    a.seed, a.layer_id, a.sample_offset = seed & 0xFFFFFFFFFFFFFFFF, layer_id, sample_offset & 0xFFFFFFFF
    a.prior = prior.c()
    glp = _f32c(g_log_prior, "g_log_prior") if g_log_prior is not None else None
    glq = _f32c(g_log_q, "g_log_q") if g_log_q is not None else None
    a.g_log_prior, a.g_log_q = _ptr(glp), _ptr(glq)
    g_wmu, g_wrho, g_bmu, g_brho = _grad_outputs(out, w_mu, w_rho, b_mu, b_rho)
    gx = torch.empty((n_samples, B, K), dtype=torch.float32, device=dev) if want_gx else None
    a.g_w_mu, a.g_w_rho, a.g_b_mu, a.g_b_rho = g_wmu.data_ptr(), g_wrho.data_ptr(), g_bmu.data_ptr(), g_brho.data_ptr()
    a.g_x = _ptr(gx)
    a.gx_relu_mask = int(bool(gx_relu_mask) and want_gx)
    if w_sampled is not None and want_gx:
        require_device(w_sampled)
        if w_sampled.dtype != torch.bfloat16 or not w_sampled.is_contiguous() or w_sampled.numel() != n_samples * N * K:
            raise BnnHipError("bbb_linear_bwd: w_sampled must be contiguous bf16 [samples,out,in]")
        a.w_sampled = w_sampled.data_ptr()
    if w_sampled_t is not None and want_gx:
        require_device(w_sampled_t, gy16)
        if w_sampled_t.dtype != torch.bfloat16 or not w_sampled_t.is_contiguous() or tuple(w_sampled_t.shape) != (n_samples, K, N):
            raise BnnHipError("bbb_linear_bwd: w_sampled_t must be contiguous bf16 [samples,in,out]")
        a.w_sampled_t = w_sampled_t.data_ptr()
        if gy16 is not None:
            if gy16.dtype != torch.bfloat16 or not gy16.is_contiguous() or gy16.numel() != gy.numel():
                raise BnnHipError("bbb_linear_bwd: gy16 must be a contiguous bf16 copy of gy")
            a.gy_bf16 = gy16.data_ptr()
    gx16 = None
    if want_gx16 and want_gx:
        gx16 = torch.empty((n_samples, B, K), dtype=torch.bfloat16, device=dev)
        a.g_x_bf16 = gx16.data_ptr()
    ws = torch.empty(lib.bnn_bbb_linear_bwd_workspace_bytes(n_samples, B, N) // 4, dtype=torch.float32, device=dev)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    a.sample_counter = _ptr(sample_counter)
    L.check(lib.bnn_bbb_linear_bwd(C.byref(a), _stream()), "bnn_bbb_linear_bwd")
    if want_gx16:
        return g_wmu, g_wrho, g_bmu, g_brho, gx, gx16
    return g_wmu, g_wrho, g_bmu, g_brho, gx


def lr_linear_bwd(x, gy, y, v, w_mu, w_rho, b_mu, b_rho, *, n_samples: int, sigma_p: float, relu: bool, eps_mode: int,
                  eps_act=None, eps_b=None, seed: int = 0, layer_id: int = 0, sample_offset: int = 0, g_kl=None,
                  want_gx: bool = True, sample_counter=None, out=None, gx_relu_mask: bool = False, math_mode: int = L.MATH_F32,
                  hfac=None):
    """F1: backward of K3 (bnn_lr_linear_bwd).  All tensors fp32; `v` is the variance the forward
    saved (lr_linear_fwd(want_v=True)), or None with `hfac` (lr_linear_fwd(want_hfac=True); relu must be False: no
    preparation launch then); g_kl float[3] = upstream grads of (kl, weight_kl, bias_kl).
    Returns (g_w_mu, g_w_rho, g_b_mu, g_b_rho, g_x[S,B,K] | None)."""
    lib = L.load()
    require_device(x, gy, y, v, hfac, w_mu, w_rho, b_mu, b_rho, eps_act, eps_b, g_kl)
    w_mu, w_rho = _f32c(w_mu, "weight_mu"), _f32c(w_rho, "weight_rho")
    b_mu, b_rho = _f32c(b_mu, "bias_mu"), _f32c(b_rho, "bias_rho")
    K, N = w_mu.shape
    xs, B, Kx, per_sample = _x3(_f32c(x, "x"), n_samples)
    gy = _f32c(gy, "gy")
    v = _f32c(v, "v") if v is not None else None
    hfac = _f32c(hfac, "hfac") if hfac is not None else None
    if v is None and (hfac is None or relu):
        raise BnnHipError("lr_linear_bwd: needs the forward's v (or its hfac for a layer without a fused ReLU)")
    if Kx != K or gy.numel() != n_samples * B * N or any(t_ is not None and t_.numel() != gy.numel() for t_ in (v, hfac)):
        raise BnnHipError("lr_linear_bwd: shape mismatch")
    dev = xs.device
    a = L.LrBwdArgs()
    a.struct_bytes = C.sizeof(L.LrBwdArgs)
    a.n_samples, a.batch, a.in_features, a.out_features = n_samples, B, K, N
    a.x, a.x_per_sample, a.relu = xs.data_ptr(), per_sample, int(relu)
    a.gy, a.v, a.hfac = gy.data_ptr(), _ptr(v), _ptr(hfac)
    if relu:
        y = _f32c(y, "y")
        a.y = y.data_ptr()
    a.w_mu, a.w_rho, a.b_mu, a.b_rho = w_mu.data_ptr(), w_rho.data_ptr(), b_mu.data_ptr(), b_rho.data_ptr()
    a.eps_mode, a.math = eps_mode, _exact_unless_bf16(int(math_mode))
    if eps_mode == L.EPS_MEMORY:
        eps_act, eps_b = _f32c(eps_act, "eps_act"), _f32c(eps_b, "eps_b")
        a.eps_act, a.eps_b = eps_act.data_ptr(), eps_b.data_ptr()
    a.seed, a.layer_id, a.sample_offset = seed & 0xFFFFFFFFFFFFFFFF, layer_id, sample_offset & 0xFFFFFFFF
    a.sigma_p = float(sigma_p)
    gk = _f32c(g_kl, "g_kl") if g_kl is not None else None
    a.g_kl = _ptr(gk)
    g_wmu, g_wrho, g_bmu, g_brho = _grad_outputs(out, w_mu, w_rho, b_mu, b_rho)
    gx = torch.empty((n_samples, B, K), dtype=torch.float32, device=dev) if want_gx else None
    a.g_w_mu, a.g_w_rho, a.g_b_mu, a.g_b_rho = g_wmu.data_ptr(), g_wrho.data_ptr(), g_bmu.data_ptr(), g_brho.data_ptr()
    a.g_x = _ptr(gx)
    a.gx_relu_mask = int(bool(gx_relu_mask) and want_gx)
    ws = torch.empty(lib.bnn_lr_linear_bwd_workspace_bytes(n_samples, B, K, N, int(want_gx)) // 4, dtype=torch.float32,
                     device=dev)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    a.sample_counter = _ptr(sample_counter)
    L.check(lib.bnn_lr_linear_bwd(C.byref(a), _stream()), "bnn_lr_linear_bwd")
    return g_wmu, g_wrho, g_bmu, g_brho, gx


def mc_softmax_mean(logits: torch.Tensor, scale: float, want_preds: bool = True, out_probs: Optional[torch.Tensor] = None,
                    out_preds: Optional[torch.Tensor] = None):
    """F3: probs[B,C] = scale * sum_s softmax(logits[s]), preds[B] = argmax (bnn_mc_softmax_mean).  `out_probs` / `out_preds`:
    static buffers of a captured evaluation."""
    lib = L.load()
    require_device(logits, out_probs, out_preds)
    lg = _f32c(logits, "logits")
    S, B, Cc = lg.shape
    if out_probs is not None and (out_probs.dtype != torch.float32 or tuple(out_probs.shape) != (B, Cc) or not out_probs.is_contiguous()):
        raise BnnHipError("mc_softmax_mean: out_probs must be a contiguous float32 [batch, classes] tensor")
    if out_preds is not None and (out_preds.dtype != torch.int64 or tuple(out_preds.shape) != (B,) or not out_preds.is_contiguous()):
        raise BnnHipError("mc_softmax_mean: out_preds must be a contiguous int64 [batch] tensor")
    probs = out_probs if out_probs is not None else torch.empty((B, Cc), dtype=torch.float32, device=lg.device)
    preds = out_preds if out_preds is not None else (torch.empty(B, dtype=torch.int64, device=lg.device) if want_preds else None)
    L.check(lib.bnn_mc_softmax_mean(lg.data_ptr(), S, B, Cc, float(scale), probs.data_ptr(), _ptr(preds), _stream()),
            "bnn_mc_softmax_mean")
    return probs, preds


def elbo_loss(a, b, nll, beta, total_samples: int, local_reparam: bool, grad_scale: float = 1.0):
    """bnn_elbo_loss: returns (out4 {loss, mean a, mean b, mean nll}, g_a, g_b, g_nll, g_kl3)."""
    lib = L.load()
    require_device(a, b, nll, beta)
    S = nll.numel()
    dev = nll.device
    out4 = torch.empty(4, dtype=torch.float32, device=dev)
    g_a = torch.empty(S, dtype=torch.float32, device=dev)
    g_b = torch.empty(S, dtype=torch.float32, device=dev)
    g_nll = torch.empty(S, dtype=torch.float32, device=dev)
    g_kl3 = torch.empty(3, dtype=torch.float32, device=dev)
    L.check(lib.bnn_elbo_loss(a.data_ptr(), _ptr(b), nll.data_ptr(), beta.data_ptr(), S, float(total_samples),
                              float(grad_scale), int(local_reparam), out4.data_ptr(), g_a.data_ptr(), g_b.data_ptr(), g_nll.data_ptr(),
                              g_kl3.data_ptr(), _stream()), "bnn_elbo_loss")
    return out4, g_a, g_b, g_nll, g_kl3


def _nll_target(target, mode: str, B: int, Cc: int):
    if mode == "classification":
        tg, m = target.to(torch.int64).contiguous(), L.NLL_CLASSIFICATION
        if tg.numel() != B:
            raise BnnHipError("classification target must have `batch` elements")
    elif mode == "regression":
        tg, m = _f32c(target.to(torch.float32), "target"), L.NLL_REGRESSION
        if tg.numel() != B * Cc:
            raise BnnHipError("regression target must match the output shape")
    else:
        raise Exception("Training mode must be either 'regression' or 'classification'")
    return tg, m


def elbo_loss_nll_bwd(a, b, nll, beta, total_samples: int, local_reparam: bool, logits, target, mode: str,
                      sigma: float = 1.0, grad_scale: float = 1.0):
    """bnn_elbo_loss_nll_bwd: elbo_loss + nll_bwd in one launch.  Returns (out4, g_a, g_b, g_kl3, g_logits)."""
    lib = L.load()
    require_device(a, b, nll, beta, logits, target)
    lg = _f32c(logits, "logits")
    S, B, Cc = lg.shape
    if nll.numel() != S:
        raise BnnHipError("nll must have one element per MC sample")
    tg, m = _nll_target(target, mode, B, Cc)
    dev = nll.device
    out4 = torch.empty(4, dtype=torch.float32, device=dev)
    g_a = torch.empty(S, dtype=torch.float32, device=dev)
    g_b = torch.empty(S, dtype=torch.float32, device=dev)
    g_kl3 = torch.empty(3, dtype=torch.float32, device=dev)
    g_logits = torch.empty_like(lg)
    L.check(lib.bnn_elbo_loss_nll_bwd(a.data_ptr(), _ptr(b), nll.data_ptr(), beta.data_ptr(), S, float(total_samples),
                                      float(grad_scale), int(local_reparam), out4.data_ptr(), g_a.data_ptr(), g_b.data_ptr(),
                                      g_kl3.data_ptr(), lg.data_ptr(), tg.data_ptr(), g_logits.data_ptr(), B, Cc, m,
                                      float(sigma), _stream()), "bnn_elbo_loss_nll_bwd")
    return out4, g_a, g_b, g_kl3, g_logits


def stage_inputs(src0, dst0, src1=None, dst1=None, word=None, value: float = 0.0, cast0=None):
    """bnn_stage_inputs: dst0 <- src0, dst1 <- src1 (same dtype, shape; contiguous device tensors), *word = value,
    one launch.  `cast0` (bf16, src0's shape, src0 fp32): also the bf16 copy of src0 (bnn_stage_inputs_cast)."""
    lib = L.load()
    require_device(src0, dst0, src1, dst1, word)
    for s_, d_ in ((src0, dst0), (src1, dst1)):
        if s_ is None:
            continue
        if s_.dtype != d_.dtype or s_.numel() != d_.numel() or not s_.is_contiguous() or not d_.is_contiguous():
            raise BnnHipError("stage_inputs: source and destination must be contiguous, same dtype and size")
    nb = lambda t: 0 if t is None else t.numel() * t.element_size()
    if cast0 is not None:
        require_device(cast0)
        if src0.dtype != torch.float32 or cast0.dtype != torch.bfloat16 or cast0.numel() != src0.numel() or not cast0.is_contiguous():
            raise BnnHipError("stage_inputs: cast0 must be a contiguous bf16 tensor of the fp32 src0's size")
        L.check(lib.bnn_stage_inputs_cast(_ptr(src0), _ptr(dst0), nb(src0), _ptr(src1), _ptr(dst1), nb(src1), _ptr(word),
                                          float(value), cast0.data_ptr(), _stream()), "bnn_stage_inputs_cast")
        return
    L.check(lib.bnn_stage_inputs(_ptr(src0), _ptr(dst0), nb(src0), _ptr(src1), _ptr(dst1), nb(src1), _ptr(word), float(value),
                                 _stream()), "bnn_stage_inputs")


def nll_bwd(logits, target, g_nll, mode: str, sigma: float = 1.0):
    """Gradient of the per-sample summed NLL w.r.t. logits[S,B,C] scaled by g_nll[S] (bnn_nll_bwd)."""
    lib = L.load()
    require_device(logits, target, g_nll)
    lg = _f32c(logits, "logits")
    S, B, Cc = lg.shape
    gn = _f32c(g_nll.reshape(-1), "g_nll")
    if gn.numel() != S:
        raise BnnHipError("g_nll must have one element per MC sample")
    tg, m = _nll_target(target, mode, B, Cc)
    out = torch.empty_like(lg)
    L.check(lib.bnn_nll_bwd(lg.data_ptr(), tg.data_ptr(), gn.data_ptr(), out.data_ptr(), S, B, Cc, m, float(sigma), _stream()),
            "bnn_nll_bwd")
    return out


def adam_step(params, grads, exp_avgs, exp_avg_sqs, *, lr: float, betas, eps: float, weight_decay: float, step: int = 0,
              lr_device=None, step_device=None, ticket=None, bump_counter=None, bump_by: int = 0):
    """F2: bnn_adam_step over lists of fp32 tensors (any number; 16 per launch; the gradients fp32 or bf16).  With `ticket` (zeroed device array
    of 16 uint32) the device step advances inside the first launch, which also adds bump_by to *bump_counter."""
    lib = L.load()
    n = len(params)
    for lo in range(0, n, L.ADAM_MAX_TENSORS):
        hi = min(n, lo + L.ADAM_MAX_TENSORS)
        a = L.AdamArgs()
        a.struct_bytes = C.sizeof(L.AdamArgs)
        a.n_tensors = hi - lo
        for j in range(lo, hi):
            p, g, m, v = params[j], grads[j], exp_avgs[j], exp_avg_sqs[j]
            require_device(p, g, m, v)
            for t_ in (p, m, v):
                if t_.dtype != torch.float32 or not t_.is_contiguous() or t_.numel() != p.numel():
                    raise BnnHipError("adam_step: tensors must be contiguous float32 of the parameter's size")
            if g.dtype not in (torch.float32, torch.bfloat16) or g.dtype != grads[lo].dtype or not g.is_contiguous() or \
                    g.numel() != p.numel():
                raise BnnHipError("adam_step: gradients must be contiguous float32 or bfloat16 (one dtype per launch) of the "
                                  "parameter's size")
            a.param[j - lo], a.grad[j - lo] = p.data_ptr(), g.data_ptr()
            a.exp_avg[j - lo], a.exp_avg_sq[j - lo] = m.data_ptr(), v.data_ptr()
            a.numel[j - lo] = p.numel()
        a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
        a.step = int(step)
        a.grad_dtype = _dt(grads[lo])
        a.lr_device = _ptr(lr_device)
        a.step_device = _ptr(step_device)
        a.step_advance = int(lo == 0)                   # the first launch of the step advances the device counter
        if lo == 0 and ticket is not None and step_device is not None:
            if ticket.numel() < 16 or ticket.element_size() != 4:
                raise BnnHipError("adam_step: ticket must be a zeroed array of 16 32-bit words")
            a.ticket = ticket.data_ptr()
            if bump_counter is not None:
                a.bump_counter, a.bump_by = bump_counter.data_ptr(), int(bump_by)
        elif bump_counter is not None and lo == 0:
            raise BnnHipError("adam_step: bump_counter needs a device step and a ticket word")
        L.check(lib.bnn_adam_step(C.byref(a), _stream()), "bnn_adam_step")


def softplus(rho: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sigma = log1p(exp(rho)) by bnn_softplus (one elementwise kernel)."""
    lib = L.load()
    require_device(rho)
    rho = _f32c(rho, "rho")
    if out is None:
        out = torch.empty_like(rho)
    L.check(lib.bnn_softplus(rho.data_ptr(), out.data_ptr(), rho.numel(), _stream()), "bnn_softplus")
    return out


def eval_prepare(rhos=(), sigmas=None, cast: Optional[torch.Tensor] = None, cast_out: Optional[torch.Tensor] = None,
                 cast_out_sq: Optional[torch.Tensor] = None, want_sq: bool = False, cast_out_lo: Optional[torch.Tensor] = None):
    """bnn_eval_prepare: sigma = softplus(rho) for every tensor of `rhos` and (optionally) the bf16 cast of `cast`
    (+ its squares; + `cast_out_lo`, the low plane bf16(x - bf16(x)) of the split-bf16 math mode) in ONE launch --
    everything of an evaluation that depends on no activation.
    Returns (list of sigma tensors, cast_out, cast_out_sq)."""
    lib = L.load()
    rhos = [_f32c(r, "rho") for r in rhos]
    require_device(*rhos, cast)
    if len(rhos) > L.PREPARE_MAX:
        raise BnnHipError(f"eval_prepare: at most {L.PREPARE_MAX} tensors per launch")
    if sigmas is None:
        sigmas = [torch.empty_like(r) for r in rhos]
    a = L.PrepareArgs()
    a.struct_bytes = C.sizeof(L.PrepareArgs)
    a.n_softplus = len(rhos)
    for i, (r, sg) in enumerate(zip(rhos, sigmas)):
        if sg.shape != r.shape or sg.dtype != torch.float32 or not sg.is_contiguous():
            raise BnnHipError("eval_prepare: sigma must be a contiguous fp32 tensor of rho's shape")
        a.rho[i], a.sigma[i], a.n[i] = r.data_ptr(), sg.data_ptr(), r.numel()
    if cast is not None:
        cast = _f32c(cast, "x")
        if cast_out is None:
            cast_out = torch.empty(cast.shape, dtype=torch.bfloat16, device=cast.device)
        if want_sq and cast_out_sq is None:
            cast_out_sq = torch.empty(cast.shape, dtype=torch.bfloat16, device=cast.device)
        a.cast_src, a.cast_dst, a.cast_dst_sq, a.cast_n = cast.data_ptr(), cast_out.data_ptr(), _ptr(cast_out_sq), cast.numel()
        if cast_out_lo is not None:
            if cast_out_lo.dtype != torch.bfloat16 or cast_out_lo.numel() != cast.numel() or not cast_out_lo.is_contiguous():
                raise BnnHipError("eval_prepare: cast_out_lo must be a contiguous bf16 tensor of the input's size")
            require_device(cast_out_lo)
            a.cast_dst_lo = cast_out_lo.data_ptr()
    if not rhos and cast is None:
        return [], None, None
    L.check(lib.bnn_eval_prepare(C.byref(a), _stream()), "bnn_eval_prepare")
    return list(sigmas), cast_out, cast_out_sq


def ece_bins(probs: torch.Tensor, labels: torch.Tensor, bin_edges) -> torch.Tensor:
    """F3: bnn_ece.  probs fp32 [n, classes], labels int64[n] on the device, bin_edges a host float64 sequence.
    Returns float32 [1 + 3 * nbins] = (ece, then per bin count, corrects, confidence sum)."""
    lib = L.load()
    require_device(probs, labels)
    pr = _f32c(probs, "probs")
    if pr.dim() != 2 or labels.numel() != pr.shape[0]:
        raise BnnHipError("ece: probs must be [n, classes] and labels [n]")
    lb = labels.to(torch.int64).contiguous()
    edges = (C.c_double * len(bin_edges))(*[float(e) for e in bin_edges])
    ws = torch.empty(lib.bnn_ece_workspace_bytes() // 8, dtype=torch.float64, device=pr.device)
    out = torch.empty(1 + 3 * (len(bin_edges) - 1), dtype=torch.float32, device=pr.device)
    L.check(lib.bnn_ece(pr.data_ptr(), lb.data_ptr(), pr.shape[0], pr.shape[1], edges, len(bin_edges), ws.data_ptr(),
                        ws.numel() * 8, out.data_ptr(), _stream()), "bnn_ece")
    return out


def snr_db(mu: torch.Tensor, rho: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F4: bnn_snr_db.  10 log10(|mu| / softplus(rho)) elementwise (fp32), shaped like mu."""
    lib = L.load()
    require_device(mu, rho)
    mu, rho = _f32c(mu, "mu"), _f32c(rho, "rho")
    if mu.numel() != rho.numel():
        raise BnnHipError("snr_db: mu and rho must have the same number of elements")
    if out is None:
        out = torch.empty_like(mu)
    L.check(lib.bnn_snr_db(mu.data_ptr(), rho.data_ptr(), mu.numel(), out.data_ptr(), _stream()), "bnn_snr_db")
    return out


def snr_prune_(mu: torch.Tensor, rho: torch.Tensor, threshold: float, kept: Optional[torch.Tensor] = None):
    """F4: bnn_snr_prune, IN PLACE on contiguous fp32 tensors: mu, rho *= (snr_db > threshold)."""
    lib = L.load()
    require_device(mu, rho)
    if mu.dtype != torch.float32 or rho.dtype != torch.float32 or not mu.is_contiguous() or not rho.is_contiguous() or \
            mu.numel() != rho.numel():
        raise BnnHipError("snr_prune_: contiguous float32 mu and rho of one size")
    L.check(lib.bnn_snr_prune(mu.data_ptr(), rho.data_ptr(), mu.numel(), float(threshold), _ptr(kept), _stream()),
            "bnn_snr_prune")
