// F2  fused Adam over all parameter tensors of the network in one launch (the update the
// reference's trainers run after loss.backward(): torch.optim.Adam, class_task.py:60/:79,
// reg_task.py:53) and the NLL backward that seeds the layer backward kernels.
//
// Adam is pure streaming: 16 B read + 12 B written per parameter (p, g, m, v -> p, m, v), no
// reuse: HBM-bound, 28 B/param.  One block = one 4096-element chunk of one tensor; the
// chunk -> tensor table rides in the kernel arguments; 16-byte accesses.
#include "bnn_device.h"
#include "../../include/bnn_hip.h"
#include <math.h>

namespace bnn {

constexpr int kAdamChunk = 4096;

struct AdamK {
  float* p[BNN_ADAM_MAX_TENSORS];
  const float* g[BNN_ADAM_MAX_TENSORS];
  float* m[BNN_ADAM_MAX_TENSORS];
  float* v[BNN_ADAM_MAX_TENSORS];
  long numel[BNN_ADAM_MAX_TENSORS];
  int first_chunk[BNN_ADAM_MAX_TENSORS + 1];
  int n;
  double lr, beta1, beta2;
  float beta2f, omb1, omb2, eps, wd;
  uint32_t step;
  const float* lr_dev;
  uint32_t* step_dev;         // already advanced by adam_tick_kernel when set and ticket == nullptr
  uint32_t* ticket;           // set: this launch advances *step_dev itself (last arriver), no tick launch
  uint32_t* bump;             // optional word the last arriver adds bump_by to (the step's MC-sample counter)
  uint32_t bump_by;
};

__global__ void adam_tick_kernel(uint32_t* step) { *step += 1u; }

// GBF16: the gradients arrive in bf16 (a data-parallel step's all-reduced 2-byte bucket)
template <bool GBF16>
__global__ __launch_bounds__(256) void adam_kernel(const AdamK k) {
  int t = 0;
#pragma unroll 1
  while (t + 1 < k.n && (int)blockIdx.x >= k.first_chunk[t + 1]) ++t;
  const long base = (long)((int)blockIdx.x - k.first_chunk[t]) * kAdamChunk;
  const long n = k.numel[t];
  float* __restrict__ p = k.p[t];
  const float* __restrict__ g = k.g[t];
  const __bf16* __restrict__ g16 = reinterpret_cast<const __bf16*>(k.g[t]);
  auto ld_g4 = [&](long i) -> float4 {
    if (!GBF16) return *reinterpret_cast<const float4*>(g + i);
    const bf16x4 h = *reinterpret_cast<const bf16x4*>(g16 + i);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
  };
  auto ld_g1 = [&](long i) -> float { return GBF16 ? (float)g16[i] : g[i]; };
  float* __restrict__ m = k.m[t];
  float* __restrict__ v = k.v[t];
  // a whole chunk (all but the last block of a tensor): its 16 loads go out before anything else -- the double-precision
  // bias corrections below are a few hundred instructions per thread, and behind them the first load round trip of every
  // wave was exposed at the head of the launch
  constexpr int kIt = kAdamChunk / (256 * 4);
  const bool whole = base + kAdamChunk <= n;
  float4 qa[kIt], qb[kIt], qc[kIt], qd[kIt];
  if (whole) {
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const long i = base + ((long)it * 256 + threadIdx.x) * 4;
      qa[it] = *reinterpret_cast<const float4*>(p + i);
      qb[it] = ld_g4(i);
      qc[it] = *reinterpret_cast<const float4*>(m + i);
      qd[it] = *reinterpret_cast<const float4*>(v + i);
    }
  }
  const double lr = k.lr_dev ? (double)*k.lr_dev : k.lr;
  // with a ticket word every block reads the old step and uses step + 1; the block that arrives last (all have
  // read it by then) stores it: the optimiser step is ONE launch, nobody waits
  const uint32_t step = k.step_dev ? *k.step_dev + (k.ticket ? 1u : 0u) : k.step;
  // bias corrections in double, as torch computes them on the host (1 - beta ** step)
  const double bc1 = 1.0 - pow(k.beta1, (double)step);
  const double bc2 = 1.0 - pow(k.beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float sqrt_bc2 = (float)sqrt(bc2);
  const float omb1 = k.omb1, omb2 = k.omb2;
  if (whole) {
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const long i = base + ((long)it * 256 + threadIdx.x) * 4;
      float pv[4] = {qa[it].x, qa[it].y, qa[it].z, qa[it].w}, gv[4] = {qb[it].x, qb[it].y, qb[it].z, qb[it].w};
      float mv[4] = {qc[it].x, qc[it].y, qc[it].z, qc[it].w}, vv[4] = {qd[it].x, qd[it].y, qd[it].z, qd[it].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gg = k.wd != 0.f ? __builtin_fmaf(k.wd, pv[j], gv[j]) : gv[j];
        mv[j] = mv[j] + (gg - mv[j]) * omb1;                       // the arithmetic of the general loop below
        vv[j] = vv[j] * k.beta2f + omb2 * gg * gg;
        const float denom = __builtin_sqrtf(vv[j]) / sqrt_bc2 + k.eps;
        pv[j] = pv[j] - step_size * (mv[j] / denom);
      }
      *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
  } else
#pragma unroll
  for (int it = 0; it < kAdamChunk / (256 * 4); ++it) {
    const long i = base + ((long)it * 256 + threadIdx.x) * 4;
    if (i >= n) break;
    float pv[4], gv[4], mv[4], vv[4];
    const bool full = i + 3 < n;
    if (full) {
      const float4 a = *reinterpret_cast<const float4*>(p + i), b = ld_g4(i);
      const float4 c = *reinterpret_cast<const float4*>(m + i), d = *reinterpret_cast<const float4*>(v + i);
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w;
      gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w;
      vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = i + j < n;
        pv[j] = ok ? p[i + j] : 0.f; gv[j] = ok ? ld_g1(i + j) : 0.f; mv[j] = ok ? m[i + j] : 0.f; vv[j] = ok ? v[i + j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = k.wd != 0.f ? __builtin_fmaf(k.wd, pv[j], gv[j]) : gv[j];
      mv[j] = mv[j] + (gg - mv[j]) * omb1;                       // exp_avg.lerp_(grad, 1 - beta1)
      vv[j] = vv[j] * k.beta2f + omb2 * gg * gg;     // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
      const float denom = __builtin_sqrtf(vv[j]) / sqrt_bc2 + k.eps;
      pv[j] = pv[j] - step_size * (mv[j] / denom);
    }
    if (full) {
      *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < n) { p[i + j] = pv[j]; m[i + j] = mv[j]; v[i + j] = vv[j]; }
    }
  }
  // every wave of the block has read *step_dev (top of the kernel) before thread 0 announces the block's arrival:
  // the last arriver overwrites the word, and a wave that had not loaded it yet would use step + 2
  if (k.ticket) __syncthreads();
  if (k.ticket && threadIdx.x == 0) {
    // two-level arrival count: device-scope atomics on ONE word serialise at the memory side (1170 of them were a
    // ~4 us tail on this 24 us kernel); blocks count in 8 groups (word 1 + blockIdx % 8), each group's last arriver
    // counts once on word 0
    const uint32_t g = blockIdx.x & 7u;
    const uint32_t in_group = (gridDim.x - g + 7u) >> 3;
    const uint32_t tk = __hip_atomic_fetch_add(k.ticket + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == in_group - 1u) {
      k.ticket[1 + g] = 0u;                               // ready for the next launch
      const uint32_t groups = gridDim.x < 8u ? gridDim.x : 8u;
      const uint32_t top = __hip_atomic_fetch_add(k.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (top == groups - 1u) {
        *k.step_dev = step;
        if (k.bump) *k.bump += k.bump_by;
        k.ticket[0] = 0u;
      }
    }
  }
}

// one thread per (sample, batch row)
__global__ void nll_bwd_kernel(const float* __restrict__ logits, const void* __restrict__ target,
                               const float* __restrict__ g_nll, float* __restrict__ g_logits, int S, int B, int C, int mode,
                               float inv_var) {
  const long total = (long)S * B;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int b = (int)(idx % B), s = (int)(idx / B);
    const float* row = logits + idx * C;
    float* out = g_logits + idx * C;
    const float gs = g_nll[s];
    if (mode == BNN_NLL_CLASSIFICATION) {
      const long long tc = reinterpret_cast<const long long*>(target)[b];
      float mx = row[0];
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
      // a label outside [0, C) poisons the row's gradient (and the forward's NLL) with NaN instead of training
      // silently on a wrong loss (nn.CrossEntropyLoss, networks.py:186, raises there; ignore_index is not supported)
      const float inv = (tc >= 0 && tc < C) ? 1.0f / se : __builtin_nanf("");
      for (int c = 0; c < C; ++c) out[c] = (expf(row[c] - mx) * inv - (c == tc ? 1.f : 0.f)) * gs;
    } else {
      const float* tg = reinterpret_cast<const float*>(target) + (size_t)b * C;
      for (int c = 0; c < C; ++c) out[c] = (row[c] - tg[c]) * inv_var * gs;
    }
  }
}

}  // namespace bnn

using namespace bnn;

extern "C" int bnn_adam_step(const bnn_adam_args* a, void* stream_) {
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_adam_args)) return BNN_ERR_ABI;
  if (a->n_tensors <= 0 || a->n_tensors > BNN_ADAM_MAX_TENSORS) return BNN_ERR_SHAPE;
  if (!a->step_device && a->step == 0) return BNN_ERR_SHAPE;
  if (!(a->beta1 >= 0.0 && a->beta1 < 1.0) || !(a->beta2 >= 0.0 && a->beta2 < 1.0) || !(a->eps >= 0.0)) return BNN_ERR_SHAPE;
  AdamK k;
  long chunks = 0;
  for (int t = 0; t < a->n_tensors; ++t) {
    if (!a->param[t] || !a->grad[t] || !a->exp_avg[t] || !a->exp_avg_sq[t]) return BNN_ERR_NULL;
    if (a->numel[t] <= 0) return BNN_ERR_SHAPE;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a->param[t]) | reinterpret_cast<uintptr_t>(a->grad[t]) |
                         reinterpret_cast<uintptr_t>(a->exp_avg[t]) | reinterpret_cast<uintptr_t>(a->exp_avg_sq[t]);
    if (al & 15) return BNN_ERR_ALIGN;
    k.p[t] = a->param[t]; k.g[t] = a->grad[t]; k.m[t] = a->exp_avg[t]; k.v[t] = a->exp_avg_sq[t];
    k.numel[t] = (long)a->numel[t];
    k.first_chunk[t] = (int)chunks;
    chunks += (a->numel[t] + kAdamChunk - 1) / kAdamChunk;
    if (chunks > 0x3fffffff) return BNN_ERR_SHAPE;
  }
  for (int t = a->n_tensors; t <= BNN_ADAM_MAX_TENSORS; ++t) k.first_chunk[t] = (int)chunks;
  for (int t = a->n_tensors; t < BNN_ADAM_MAX_TENSORS; ++t) {
    k.p[t] = nullptr; k.g[t] = nullptr; k.m[t] = nullptr; k.v[t] = nullptr; k.numel[t] = 0;
  }
  k.n = a->n_tensors;
  k.lr = a->lr; k.beta1 = a->beta1; k.beta2 = a->beta2; k.eps = (float)a->eps; k.wd = (float)a->weight_decay;
  k.beta2f = (float)a->beta2; k.omb1 = (float)(1.0 - a->beta1); k.omb2 = (float)(1.0 - a->beta2);
  k.step = a->step; k.lr_dev = a->lr_device; k.step_dev = a->step_device;
  const bool ticketed = a->step_device && a->step_advance && a->ticket;
  if (a->bump_counter && !ticketed) return BNN_ERR_NULL;   // the counter rides on the ticketed hand-off only
  k.ticket = ticketed ? a->ticket : nullptr;
  k.bump = a->bump_counter; k.bump_by = a->bump_by;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (a->step_device && a->step_advance && !ticketed)
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, stream, a->step_device);
  if ((unsigned)a->grad_dtype > 1u) return BNN_ERR_ENUM;
  if (a->grad_dtype == BNN_BF16) hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)chunks), dim3(256), 0, stream, k);
  else hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)chunks), dim3(256), 0, stream, k);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_nll_bwd(const float* logits, const void* target, const float* g_nll, float* g_logits, int32_t n_samples,
                           int32_t batch, int32_t classes, int32_t nll_mode, float nll_sigma, void* stream_) {
  if (!logits || !target || !g_nll || !g_logits) return BNN_ERR_NULL;
  if (n_samples <= 0 || batch <= 0 || classes <= 0) return BNN_ERR_SHAPE;
  if ((unsigned)nll_mode > 1u) return BNN_ERR_ENUM;
  if (nll_mode == BNN_NLL_REGRESSION && !(nll_sigma > 0.f)) return BNN_ERR_SHAPE;
  const long total = (long)n_samples * batch;
  long nb = (total + 127) / 128;
  nb = nb > 2048 ? 2048 : nb;
  const float inv_var = nll_mode == BNN_NLL_REGRESSION ? (float)(1.0 / ((double)nll_sigma * nll_sigma)) : 0.f;
  hipLaunchKernelGGL(nll_bwd_kernel, dim3((unsigned)nb), dim3(128), 0, reinterpret_cast<hipStream_t>(stream_), logits, target,
                     g_nll, g_logits, n_samples, batch, classes, nll_mode, inv_var);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}
