// K2 gauss_kl (streaming KL / log-sigma reduction over (mu, rho)), K4 elbo_finalize
// (per-sample log p / log q / KL / NLL scalars of one ELBO evaluation), the Philox
// epsilon materialiser, and the library's version/status entry points.
#include "bnn_device.h"
#include "bnn_fin.h"
#include "../../include/bnn_hip.h"

namespace bnn {

// ----------------------------------------------------------------------------- K2
// HBM-bound: 8 algorithmic bytes per element (mu, rho read once as 16-byte vectors),
// per-lane fp32 partials over a grid-stride loop, wave shuffle, LDS, ONE float4 partial per
// block.  No atomics.
constexpr int kKlBlock = 256;
constexpr int kKlMaxBlocks = 2048;

__global__ __launch_bounds__(kKlBlock) void gauss_kl_partial_kernel(const float* __restrict__ mu,
                                                                  const float* __restrict__ rho, long n,
                                                                  int vec_ok, float4* __restrict__ partial) {
  __shared__ float scratch[3 * (kKlBlock / 64)];
  float ls = 0.f, s2 = 0.f, m2 = 0.f;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nthreads = (long)gridDim.x * blockDim.x;
  if (vec_ok) {
    const long n4 = n >> 2;
    const float4* mu4 = reinterpret_cast<const float4*>(mu);
    const float4* rho4 = reinterpret_cast<const float4*>(rho);
    for (long i = tid; i < n4; i += nthreads) {
      const float4 m = mu4[i], r = rho4[i];
      const float a = softplus(r.x), b = softplus(r.y), c = softplus(r.z), d = softplus(r.w);
      ls += (fast_log(a) + fast_log(b)) + (fast_log(c) + fast_log(d));
      s2 += (a * a + b * b) + (c * c + d * d);
      m2 += (m.x * m.x + m.y * m.y) + (m.z * m.z + m.w * m.w);
    }
    for (long i = (n4 << 2) + tid; i < n; i += nthreads) {
      const float a = softplus(rho[i]);
      ls += fast_log(a);
      s2 += a * a;
      m2 += mu[i] * mu[i];
    }
  } else {
    for (long i = tid; i < n; i += nthreads) {
      const float a = softplus(rho[i]);
      ls += fast_log(a);
      s2 += a * a;
      m2 += mu[i] * mu[i];
    }
  }
  ls = wave_sum(ls);
  s2 = wave_sum(s2);
  m2 = wave_sum(m2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    scratch[wave * 3 + 0] = ls;
    scratch[wave * 3 + 1] = s2;
    scratch[wave * 3 + 2] = m2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int w = 0; w < kKlBlock / 64; ++w) {
      a += scratch[w * 3 + 0];
      b += scratch[w * 3 + 1];
      c += scratch[w * 3 + 2];
    }
    partial[blockIdx.x] = make_float4(a, b, c, 0.f);
  }
}

__global__ void gauss_kl_final_kernel(const float4* __restrict__ partial, int nblocks, long n, float sigma_p,
                                      float* __restrict__ out4) {
  __shared__ double scratch[16];
  double ls = 0, s2 = 0, m2 = 0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
    const float4 v = partial[i];
    ls += v.x;
    s2 += v.y;
    m2 += v.z;
  }
  ls = block_sum(ls, scratch);
  s2 = block_sum(s2, scratch);
  m2 = block_sum(m2, scratch);
  if (threadIdx.x == 0) {
    const double cnt = (double)n, sp = sigma_p;
    out4[0] = (float)(0.5 * (2.0 * cnt * log(sp) - 2.0 * ls - cnt + (s2 + m2) / (sp * sp)));
    out4[1] = (float)ls;
    out4[2] = (float)s2;
    out4[3] = (float)m2;
  }
}

static inline int kl_blocks(long n) {
  long b = (n / 4 + kKlBlock - 1) / kKlBlock;
  if (b < 1) b = 1;
  if (b > kKlMaxBlocks) b = kKlMaxBlocks;
  return (int)b;
}

// ----------------------------------------------------------------------------- K4
__global__ void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, __bf16* __restrict__ dsq,
                                 long n, int vec_ok) {
  cast_bf16_span(src, dst, dsq, n, vec_ok, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// grid = n_samples blocks (one sample each), or ONE block looping over all samples when `single`: then the
// block also writes the 4-vector(s) of sums, in sample order.  With a `ticket` word the one-block-per-sample form
// does that too: the last-arriving block folds the per-sample scalars (release / ticket / acquire, nobody
// waits), so a handful of samples is finalized in parallel and still in one launch (8 samples: 31.7 us serially).
__global__ __launch_bounds__(256) void elbo_finalize_kernel(const FinK p, const FinC cst, int single, float* sums,
                                                            uint32_t* ticket) {
  __shared__ __attribute__((aligned(8))) float part[4 * kFinNV];
  int T[8];
#pragma unroll
  for (int l = 0; l < 8; ++l)
    T[l] = (l < p.n_layers) ? __float_as_int(reinterpret_cast<const float4*>(p.ws[l])[0].x) : 0;
  const int s_begin = single ? 0 : blockIdx.x, s_end = single ? p.S : blockIdx.x + 1;
  const int g = p.group > 0 ? p.group : p.S;
  double tot_a = 0, tot_b = 0, tot_n = 0;
  for (int s = s_begin; s < s_end; ++s) {
    float a = 0.f, b = 0.f, nll = 0.f;
    const float* lg = p.logits ? p.logits + (size_t)s * p.B * p.C : nullptr;
    fin_sample(p, cst, s, T, lg, p.C, -1, 0.f, 0.f, 0.f, part, a, b, nll);
    if (threadIdx.x == 0) {
      fin_store(p, s, a, b, nll);
      tot_a += a;
      tot_b += b;
      tot_n += nll;
      if (single && (s + 1) % g == 0) {                       // a minibatch (or the one evaluation) is complete
        if (sums) {
          float* so = sums + 4 * (s / g);
          so[0] = (float)tot_a; so[1] = (float)tot_b; so[2] = (float)tot_n; so[3] = (float)g;
        }
        tot_a = tot_b = tot_n = 0;
      }
    }
    if (s + 1 < s_end) __syncthreads();
  }
  if (single) {
    if (threadIdx.x == 0 && p.sample_counter) *p.sample_counter += p.sample_counter_inc;
    return;
  }
  if (!ticket) {                                             // the sums (if any) come from a follow-up launch
    if (p.sample_counter && blockIdx.x == 0 && threadIdx.x == 0) *p.sample_counter += p.sample_counter_inc;
    return;
  }
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == (uint32_t)p.S - 1u) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (sums) fin_fold_sums(p, sums);                      // sample order: the sums do not depend on who arrived when
      *ticket = 0u;
      if (p.sample_counter) *p.sample_counter += p.sample_counter_inc;
    }
  }
}

// NLL of wide outputs, split by rows: one block per (row block, sample) sums the NLL of its rows into
// partial[s][rb]; elbo_finalize_kernel then adds the row blocks up instead of walking B x C logits through ONE CU per
// sample (4096 outputs, batch 128, 4 samples: 101 us of a 485 us evaluation, 40 GB/s per CU).
__global__ __launch_bounds__(256) void nll_rows_kernel(const FinK p, const FinC cst, int rows_per_block, float* __restrict__ partial) {
  __shared__ float red[4];
  const int rb = blockIdx.x, s = blockIdx.y, nrb = gridDim.x;
  const int row0 = rb * rows_per_block, row1 = min(p.B, row0 + rows_per_block);
  const float acc = fin_nll(p, cst, s, p.logits + (size_t)s * p.B * p.C, p.C, row0, row1);
  const float w = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)s * nrb + rb] = ((red[0] + red[1]) + red[2]) + red[3];
}

// Sums of the per-sample scalars over the local samples: the 4-vector a sharded job all-reduces (one per minibatch
// of `g` samples).  One evaluation: index order per thread and a fixed tree across threads; several minibatches: a
// thread per minibatch, sample order.
__global__ void sample_sums_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                   const float* __restrict__ c, int S, int g, float* sums, uint32_t* counter,
                                   uint32_t counter_inc) {
  __shared__ double scratch[16];
  if (g > 0 && g < S) {
    const int G = S / g;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < G; m += gridDim.x * blockDim.x) {
      double x = 0, y = 0, z = 0;
      for (int i = m * g; i < (m + 1) * g; ++i) {
        if (a) x += a[i];
        if (b) y += b[i];
        if (c) z += c[i];
      }
      if (sums) {
        sums[4 * m + 0] = (float)x;
        sums[4 * m + 1] = (float)y;
        sums[4 * m + 2] = (float)z;
        sums[4 * m + 3] = (float)g;
      }
    }
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) *counter += counter_inc;
    return;
  }
  double x = 0, y = 0, z = 0;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    if (a) x += a[i];
    if (b) y += b[i];
    if (c) z += c[i];
  }
  x = block_sum(x, scratch);
  y = block_sum(y, scratch);
  z = block_sum(z, scratch);
  if (threadIdx.x == 0) {
    if (sums) {
      sums[0] = (float)x;
      sums[1] = (float)y;
      sums[2] = (float)z;
      sums[3] = (float)S;
    }
    if (counter) *counter += counter_inc;
  }
}

// sigma = softplus(rho), once per ELBO evaluation: the throughput form of K1 re-reads the weight
// tile for every MC sample, so hoisting the softplus out of the sample dimension removes ~20 %
// of its VALU work (the reference recomputes sigma three times per forward, networks.py:37-46).
__global__ void softplus_kernel(const float* __restrict__ rho, float* __restrict__ sigma, long n, int vec_ok) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
  if (vec_ok) {
    for (long i = tid; i < (n >> 2); i += nt) {
      const float4 r = reinterpret_cast<const float4*>(rho)[i];
      reinterpret_cast<float4*>(sigma)[i] = make_float4(softplus(r.x), softplus(r.y), softplus(r.z), softplus(r.w));
    }
    for (long i = ((n >> 2) << 2) + tid; i < n; i += nt) sigma[i] = softplus(rho[i]);
  } else {
    for (long i = tid; i < n; i += nt) sigma[i] = softplus(rho[i]);
  }
}

// ----------------------------------------------------------------------------- evaluation prologue
// Everything of an evaluation that depends on no activation, in ONE launch: sigma = softplus(rho) of up to
// BNN_PREPARE_MAX tensors (read by the block-GEMM forms of K1 instead of a softplus per sampled weight) and the bf16
// cast (+ squares) of the input batch.  Each block owns 4096 consecutive elements of one job; elementwise, so the
// results are those of bnn_softplus / bnn_cast_bf16.
struct PrepK {
  const float* rho[BNN_PREPARE_MAX];
  float* sigma[BNN_PREPARE_MAX];
  long n[BNN_PREPARE_MAX];
  int first_block[BNN_PREPARE_MAX + 1];   // of each softplus job; [n_softplus] = first block of the cast
  int n_softplus;
  const float* cast_src;
  __bf16* cast_dst;
  __bf16* cast_dsq;
  __bf16* cast_dlo;
  long cast_n;
  int cast_vec;
};
constexpr int kPrepPerBlock = 4096;

__global__ __launch_bounds__(256) void eval_prepare_kernel(const PrepK p) {
  const int b = blockIdx.x;
  if (b >= p.first_block[p.n_softplus]) {                  // block-uniform
    const long lo = (long)(b - p.first_block[p.n_softplus]) * kPrepPerBlock;
    const long cnt = min((long)kPrepPerBlock, p.cast_n - lo);
    // a block-local span: the vector path needs the span's start 16-byte aligned in all three arrays (lo is a
    // multiple of 4096 elements)
    cast_bf16_span(p.cast_src + lo, p.cast_dst + lo, p.cast_dsq ? p.cast_dsq + lo : nullptr, cnt, p.cast_vec, threadIdx.x, 256,
                   p.cast_dlo ? p.cast_dlo + lo : nullptr);
    return;
  }
  int j = 0;
#pragma unroll 1
  while (j + 1 < p.n_softplus && b >= p.first_block[j + 1]) ++j;
  const long lo = (long)(b - p.first_block[j]) * kPrepPerBlock;
  const long cnt = min((long)kPrepPerBlock, p.n[j] - lo);
  const float* rho = p.rho[j] + lo;
  float* sg = p.sigma[j] + lo;
  const bool vec = !((reinterpret_cast<uintptr_t>(rho) | reinterpret_cast<uintptr_t>(sg)) & 15);
  if (vec) {
    for (long i = threadIdx.x; i < (cnt >> 2); i += 256) {
      const float4 r = reinterpret_cast<const float4*>(rho)[i];
      reinterpret_cast<float4*>(sg)[i] = make_float4(softplus(r.x), softplus(r.y), softplus(r.z), softplus(r.w));
    }
    for (long i = ((cnt >> 2) << 2) + threadIdx.x; i < cnt; i += 256) sg[i] = softplus(rho[i]);
  } else {
    for (long i = threadIdx.x; i < cnt; i += 256) sg[i] = softplus(rho[i]);
  }
}


// ----------------------------------------------------------------------------- Philox fill
__global__ void philox_normal_kernel(float* __restrict__ eps, uint32_t k0, uint32_t k1, uint32_t tensor_id,
                                     uint32_t sample_offset, int S, int rows, int cols) {
  const int gpr = (cols + 3) >> 2;
  const long groups = (long)rows * gpr;
  const long total = groups * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int s = (int)(i / groups);
    const long g = i - (long)s * groups;
    const int row = (int)(g / gpr), c0 = (int)(g - (long)row * gpr) * 4;
    float e[4];
    philox_normal4((uint32_t)g, sample_offset + (uint32_t)s, tensor_id, k0, k1, e);
    float* out = eps + ((size_t)s * rows + row) * cols + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < cols) out[j] = e[j];
  }
}

// ELBO assembly of one training step (reference networks.py:205-208 / :222-224) from the per-sample
// scalars, and the seeds of the backward chain: d loss / d log p[s] = -beta/S, d loss / d log q[s] =
// beta/S, d loss / d nll[s] = 1/S; LR: d loss / d (a layer's KL) = beta.  One block.
__device__ __forceinline__ void elbo_loss_block(const float* __restrict__ a, const float* __restrict__ b,
                                                const float* __restrict__ nll, const float* __restrict__ beta_p, int S,
                                                float total, float grad_scale, int local_reparam, float* __restrict__ out4,
                                                float* __restrict__ g_a, float* __restrict__ g_b, float* __restrict__ g_nll,
                                                float* __restrict__ g_kl3) {
  __shared__ double scratch[16];
  double x = 0, y = 0, z = 0;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    x += a[i];
    if (b) y += b[i];
    z += nll[i];
  }
  x = block_sum(x, scratch);
  y = block_sum(y, scratch);
  z = block_sum(z, scratch);
  const float beta = *beta_p;
  const float inv = grad_scale / total;                 // grad_scale = 1 / (data-parallel ranks): the seeds of a SUM all-reduce
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    if (g_a) g_a[i] = local_reparam ? 0.f : -beta * inv;
    if (g_b) g_b[i] = beta * inv;
    if (g_nll) g_nll[i] = inv;
  }
  if (threadIdx.x == 0) {
    const float am = (float)x / total, bm = (float)y / total, nm = (float)z / total;
    // fp32, in the reference's order: beta * log_q_mean - beta * log_prior_mean + nll  |  beta * kl_mean + nll
    out4[0] = local_reparam ? beta * am + nm : beta * bm - beta * am + nm;
    out4[1] = am;
    out4[2] = bm;
    out4[3] = nm;
    if (g_kl3) { g_kl3[0] = beta * grad_scale; g_kl3[1] = 0.f; g_kl3[2] = 0.f; }
  }
}

__global__ void elbo_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ nll,
                                 const float* __restrict__ beta_p, int S, float total, float grad_scale, int local_reparam,
                                 float* __restrict__ out4, float* __restrict__ g_a, float* __restrict__ g_b,
                                 float* __restrict__ g_nll, float* __restrict__ g_kl3) {
  elbo_loss_block(a, b, nll, beta_p, S, total, grad_scale, local_reparam, out4, g_a, g_b, g_nll, g_kl3);
}

// The same, and in the same launch the gradient of the summed NLL w.r.t. the logits (bnn_nll_bwd with the constant
// seed grad_scale / total): block 0 assembles the loss and the seeds, blocks 1.. take one (sample, batch row) per thread.
__global__ __launch_bounds__(256) void elbo_loss_nll_bwd_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ nll, const float* __restrict__ beta_p,
    int S, float total, float grad_scale, int local_reparam, float* __restrict__ out4, float* __restrict__ g_a,
    float* __restrict__ g_b, float* __restrict__ g_kl3, const float* __restrict__ logits, const void* __restrict__ target,
    float* __restrict__ g_logits, int B, int C, int mode, float inv_var) {
  if (blockIdx.x == 0) {
    elbo_loss_block(a, b, nll, beta_p, S, total, grad_scale, local_reparam, out4, g_a, g_b, nullptr, g_kl3);
    return;
  }
  const float gs = grad_scale / total;
  const long rows = (long)S * B;
  for (long idx = (long)(blockIdx.x - 1) * blockDim.x + threadIdx.x; idx < rows; idx += (long)(gridDim.x - 1) * blockDim.x) {
    const int brow = (int)(idx % B);
    const float* row = logits + idx * C;
    float* out = g_logits + idx * C;
    if (mode == BNN_NLL_CLASSIFICATION) {
      const long long tc = reinterpret_cast<const long long*>(target)[brow];
      float mx = row[0];
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
      const float inv = (tc >= 0 && tc < C) ? 1.0f / se : __builtin_nanf("");   // bad label: NaN, as in nll_bwd_kernel
      for (int c = 0; c < C; ++c) out[c] = (expf(row[c] - mx) * inv - (c == tc ? 1.f : 0.f)) * gs;
    } else {
      const float* tg = reinterpret_cast<const float*>(target) + (size_t)brow * C;
      for (int c = 0; c < C; ++c) out[c] = (row[c] - tg[c]) * inv_var * gs;
    }
  }
}

// Inputs of a captured training step staged into its static buffers in one launch: up to two device-to-device
// copies (16-byte words when everything is aligned) and one float word (the step's KL weight).
__global__ __launch_bounds__(256) void stage_inputs_kernel(const char* __restrict__ s0, char* __restrict__ d0, size_t n0,
                                                           const char* __restrict__ s1, char* __restrict__ d1, size_t n1,
                                                           int vec, float* __restrict__ word, float value,
                                                           __bf16* __restrict__ cast0) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  if (tid == 0 && word) *word = value;
  if (vec) {
    for (size_t i = tid; i < (n0 >> 4); i += nt) {
      const float4 v = reinterpret_cast<const float4*>(s0)[i];
      reinterpret_cast<float4*>(d0)[i] = v;
      if (cast0) {                                          // src0 is fp32: also its bf16 copy (vec only)
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        reinterpret_cast<bf16x4*>(cast0)[i] = o;
      }
    }
    for (size_t i = tid; i < (n1 >> 4); i += nt) reinterpret_cast<float4*>(d1)[i] = reinterpret_cast<const float4*>(s1)[i];
  } else {
    for (size_t i = tid; i < n0; i += nt) d0[i] = s0[i];
    for (size_t i = tid; i < n1; i += nt) d1[i] = s1[i];
  }
}

// ----------------------------------------------------------------------------- F3
// MC-averaged class probabilities (reference classification/class_task.py:81-87): one wave per
// batch row walks the samples; lanes stride over the classes.
__global__ __launch_bounds__(256) void mc_softmax_mean_kernel(const float* __restrict__ logits, int S, int B, int C,
                                                              float scale, float* __restrict__ probs,
                                                              long long* __restrict__ preds) {
  // A wave per batch row; lane c owns classes c, c + 64, ...  The mean is kept in REGISTERS per 64-class chunk and the
  // samples' logits are fetched eight at a time (independent loads): the first version added every sample into probs in
  // global memory -- a dependent load / store round trip per sample, 11.8 us for 10 samples of 128 x 10 logits.
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B) return;                                    // wave-uniform
  float* out = probs + (size_t)row * C;
  float best = -1.f;
  int bi = 0x7fffffff;
  if (C <= 64) {
    // the common case (10 classes): one class per lane, the whole softmax of a sample inside the wave
    float acc = 0.f;
    for (int s0 = 0; s0 < S; s0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = (s0 + j < S && lane < C) ? logits[((size_t)(s0 + j) * B + row) * C + lane] : -3.0e38f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (s0 + j < S) {                                  // wave-uniform
          float mx = v[j];
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
          const float e = lane < C ? expf(v[j] - mx) : 0.f;
          const float se = wave_sum(e);
          acc += e * (scale / se);                         // out / test_samples (class_task.py:85)
        }
      }
    }
    if (lane < C) {
      out[lane] = acc;
      best = acc;
      bi = lane;
    }
  } else {
    for (int c = lane; c < C; c += 64) out[c] = 0.f;
    for (int s = 0; s < S; ++s) {
      const float* lg = logits + ((size_t)s * B + row) * C;
      float mx = -3.0e38f;
      for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lg[c]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      float se = 0.f;
      for (int c = lane; c < C; c += 64) se += expf(lg[c] - mx);
      se = wave_sum(se);
      const float inv = scale / se;
      for (int c = lane; c < C; c += 64) out[c] += expf(lg[c] - mx) * inv;   // lane c owns out[c]: no race
    }
    for (int c = lane; c < C; c += 64) {
      const float v = out[c];
      if (v > best) { best = v; bi = c; }
    }
  }
  if (preds) {                                             // argmax, lowest index on ties
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) preds[row] = bi;
  }
}

}  // namespace bnn

using namespace bnn;

extern "C" size_t bnn_gauss_kl_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return (size_t)kl_blocks((long)n) * sizeof(float4);
}

extern "C" int bnn_gauss_kl(const float* mu, const float* rho, int64_t n, float sigma_p, void* workspace,
                            size_t workspace_bytes, float* out4, void* stream_) {
  if (!mu || !rho || !out4) return BNN_ERR_NULL;
  if (n <= 0 || !(sigma_p > 0.f)) return BNN_ERR_SHAPE;
  if (!workspace || workspace_bytes < bnn_gauss_kl_workspace_bytes(n)) return BNN_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return BNN_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(mu) & 3) || (reinterpret_cast<uintptr_t>(rho) & 3)) return BNN_ERR_ALIGN;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int nb = kl_blocks((long)n);
  const int vec_ok = !((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(rho)) & 15);
  hipLaunchKernelGGL(gauss_kl_partial_kernel, dim3(nb), dim3(kKlBlock), 0, stream, mu, rho, (long)n, vec_ok,
                     reinterpret_cast<float4*>(workspace));
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(gauss_kl_final_kernel, dim3(1), dim3(256), 0, stream,
                     reinterpret_cast<const float4*>(workspace), nb, (long)n, sigma_p, out4);
  err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_elbo_loss(const float* a, const float* b, const float* nll, const float* beta, int32_t n_samples,
                             float total_samples, float grad_scale, int32_t local_reparam, float* out4, float* g_a, float* g_b,
                             float* g_nll, float* g_kl3, void* stream_) {
  if (!a || !nll || !beta || !out4) return BNN_ERR_NULL;
  if (!local_reparam && !b) return BNN_ERR_NULL;
  if (n_samples <= 0 || !(total_samples > 0.f)) return BNN_ERR_SHAPE;
  hipLaunchKernelGGL(elbo_loss_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), a, b, nll, beta,
                     n_samples, total_samples, grad_scale, local_reparam, out4, g_a, g_b, g_nll, g_kl3);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_elbo_loss_nll_bwd(const float* a, const float* b, const float* nll, const float* beta, int32_t n_samples,
                                     float total_samples, float grad_scale, int32_t local_reparam, float* out4, float* g_a,
                                     float* g_b, float* g_kl3, const float* logits, const void* target, float* g_logits,
                                     int32_t batch, int32_t classes, int32_t nll_mode, float nll_sigma, void* stream_) {
  if (!a || !nll || !beta || !out4 || !logits || !target || !g_logits) return BNN_ERR_NULL;
  if (!local_reparam && !b) return BNN_ERR_NULL;
  if (n_samples <= 0 || batch <= 0 || classes <= 0 || !(total_samples > 0.f)) return BNN_ERR_SHAPE;
  if ((unsigned)nll_mode > 1u) return BNN_ERR_ENUM;
  if (nll_mode == BNN_NLL_REGRESSION && !(nll_sigma > 0.f)) return BNN_ERR_SHAPE;
  const long rows = (long)n_samples * batch;
  long nb = (rows + 255) / 256;
  nb = nb > 2048 ? 2048 : nb;
  const float inv_var = nll_mode == BNN_NLL_REGRESSION ? (float)(1.0 / ((double)nll_sigma * nll_sigma)) : 0.f;
  hipLaunchKernelGGL(elbo_loss_nll_bwd_kernel, dim3((unsigned)nb + 1u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                     a, b, nll, beta, n_samples, total_samples, grad_scale, local_reparam, out4, g_a, g_b, g_kl3, logits,
                     target, g_logits, batch, classes, nll_mode, inv_var);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

// the training step's tail (bnn_finalize_args.loss) as its own launch, for the launch functions whose finalizing
// launch did not carry it
extern "C" int bnn_loss_tail_(const bnn_finalize_args* f, void* stream_) {
  const bnn_loss_args* t = f->loss;
  if (!t) return BNN_OK;
  return bnn_elbo_loss_nll_bwd(f->local_reparam ? f->kl : f->log_prior, f->local_reparam ? nullptr : f->log_q, f->nll, t->beta,
                               f->n_samples, t->total_samples, t->grad_scale, f->local_reparam, t->out4, t->g_a, t->g_b, t->g_kl3,
                               f->logits, f->target, t->g_logits, f->batch, f->classes, f->nll_mode, f->nll_sigma, stream_);
}

extern "C" int bnn_stage_inputs_cast(const void* src0, void* dst0, size_t bytes0, const void* src1, void* dst1, size_t bytes1,
                                     float* word, float value, void* cast0_bf16, void* stream_);

extern "C" int bnn_stage_inputs(const void* src0, void* dst0, size_t bytes0, const void* src1, void* dst1, size_t bytes1,
                                float* word, float value, void* stream_) {
  return bnn_stage_inputs_cast(src0, dst0, bytes0, src1, dst1, bytes1, word, value, nullptr, stream_);
}

extern "C" int bnn_stage_inputs_cast(const void* src0, void* dst0, size_t bytes0, const void* src1, void* dst1, size_t bytes1,
                                     float* word, float value, void* cast0_bf16, void* stream_) {
  if ((bytes0 && (!src0 || !dst0)) || (bytes1 && (!src1 || !dst1))) return BNN_ERR_NULL;
  if (!bytes0 && !bytes1 && !word) return BNN_OK;
  const uintptr_t al = reinterpret_cast<uintptr_t>(src0) | reinterpret_cast<uintptr_t>(dst0) | (uintptr_t)bytes0 |
                       reinterpret_cast<uintptr_t>(src1) | reinterpret_cast<uintptr_t>(dst1) | (uintptr_t)bytes1;
  const int vec = (al & 15) == 0;
  if (cast0_bf16 && (!vec || (reinterpret_cast<uintptr_t>(cast0_bf16) & 7))) return BNN_ERR_ALIGN;
  const size_t words = vec ? ((bytes0 > bytes1 ? bytes0 : bytes1) >> 4) : (bytes0 > bytes1 ? bytes0 : bytes1);
  size_t nb = (words + 255) / 256;
  nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
  hipLaunchKernelGGL(stage_inputs_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                     reinterpret_cast<const char*>(src0), reinterpret_cast<char*>(dst0), bytes0,
                     reinterpret_cast<const char*>(src1), reinterpret_cast<char*>(dst1), bytes1, vec, word, value,
                     reinterpret_cast<__bf16*>(cast0_bf16));
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_mc_softmax_mean(const float* logits, int32_t n_samples, int32_t batch, int32_t classes, float scale,
                                  float* probs, long long* preds, void* stream_) {
  if (!logits || !probs) return BNN_ERR_NULL;
  if (n_samples <= 0 || batch <= 0 || classes <= 0) return BNN_ERR_SHAPE;
  hipLaunchKernelGGL(mc_softmax_mean_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream_), logits, n_samples, batch, classes, scale, probs, preds);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" size_t bnn_bbb_final_scratch_bytes(int32_t n_samples);   // bbb_linear.hip

extern "C" int bnn_elbo_finalize(const bnn_finalize_args* a, void* stream_) {
  FinK k;
  FinC cst;
  const int rc = make_fin(a, k, cst);
  if (rc != BNN_OK) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  // one sample, or a few with small logits and no ticket word to fold them in parallel: one block walks them all
  // and writes the sums; otherwise a block per sample, the sums folded by the last arriver (ticket) or by a
  // follow-up launch
  // wide outputs: the rows' NLL first, spread over the chip (needs the caller's scratch for the row-block sums)
  if (a->nll && a->classes > 32 && (long)a->batch * a->classes >= 32768 && a->scratch &&
      a->scratch_bytes >= bnn_bbb_final_scratch_bytes(a->n_samples) && !(reinterpret_cast<uintptr_t>(a->scratch) & 15)) {
    const int rpb = 4;                                        // a wave per row
    const int nrb = (a->batch + rpb - 1) / rpb;
    if (nrb <= 8 * 2048) {                                    // the scratch's partial-tile region: 64 KiB per sample
      float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(a->scratch) + (((size_t)a->n_samples * 4 + 255) / 256) * 256 +
                                                (size_t)a->n_samples * 8 * 16);
      // [sample][row block], 16384 floats of room per sample
      hipLaunchKernelGGL(nll_rows_kernel, dim3((unsigned)nrb, (unsigned)a->n_samples), dim3(256), 0, stream, k, cst, rpb, partial);
      const hipError_t e0 = hipGetLastError();
      if (e0 != hipSuccess) return (int)e0;
      k.nll_partial = partial;
      k.nll_rb = nrb;
    }
  }
  const bool small = (long)a->n_samples * a->batch * a->classes <= 65536;
  const bool ticketed = a->n_samples > 1 && a->n_samples <= 64 && a->ticket != nullptr;
  const int single = a->n_samples == 1 || (a->n_samples <= 16 && small && !ticketed);
  hipLaunchKernelGGL(elbo_finalize_kernel, dim3((unsigned)(single ? 1 : a->n_samples)), dim3(256), 0, stream, k, cst, single,
                     a->sums, (single || !ticketed) ? (uint32_t*)nullptr : a->ticket);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  if (a->sums && !single && !ticketed) {
    const float* first = a->local_reparam ? a->kl : a->log_prior;
    const float* second = a->local_reparam ? nullptr : a->log_q;
    hipLaunchKernelGGL(sample_sums_kernel, dim3(1), dim3(256), 0, stream, first, second, a->nll, a->n_samples,
                       a->group_samples, a->sums, (uint32_t*)nullptr, 0u);
    err = hipGetLastError();
  }
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_philox_normal(float* eps, uint64_t seed, uint32_t tensor_id, uint32_t sample_offset,
                                 int32_t n_samples, int32_t rows, int32_t cols, void* stream_) {
  if (!eps) return BNN_ERR_NULL;
  if (n_samples <= 0 || rows <= 0 || cols <= 0) return BNN_ERR_SHAPE;
  const long total = (long)n_samples * rows * ((cols + 3) / 4);
  long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), eps,
                     (uint32_t)seed, (uint32_t)(seed >> 32), tensor_id, sample_offset, n_samples, rows, cols);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_elbo_sums_(const bnn_finalize_args* f, void* stream_) {
  const float* first = f->local_reparam ? f->kl : f->log_prior;
  const float* second = f->local_reparam ? nullptr : f->log_q;
  hipLaunchKernelGGL(sample_sums_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), first, second,
                     f->nll, f->n_samples, f->group_samples, f->sums, f->sample_counter, f->sample_counter_inc);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_softplus(const float* rho, float* sigma, int64_t n, void* stream_) {
  if (!rho || !sigma) return BNN_ERR_NULL;
  if (n <= 0) return BNN_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(rho) | reinterpret_cast<uintptr_t>(sigma)) & 3) return BNN_ERR_ALIGN;
  const int vec_ok = !((reinterpret_cast<uintptr_t>(rho) | reinterpret_cast<uintptr_t>(sigma)) & 15);
  long nb = (n / 4 + 255) / 256;
  nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  hipLaunchKernelGGL(softplus_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), rho, sigma,
                     (long)n, vec_ok);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_cast_bf16(const float* src, void* dst, void* dst_sq, int64_t n, void* stream_) {
  if (!src || !dst) return BNN_ERR_NULL;
  if (n <= 0) return BNN_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(src) & 3) || (reinterpret_cast<uintptr_t>(dst) & 1)) return BNN_ERR_ALIGN;
  const int vec_ok = !((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) |
                        reinterpret_cast<uintptr_t>(dst_sq)) & 15);
  long nb = (n / 8 + 255) / 256;
  nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), src,
                     reinterpret_cast<__bf16*>(dst), reinterpret_cast<__bf16*>(dst_sq), (long)n, vec_ok);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_eval_prepare(const bnn_prepare_args* a, void* stream_) {
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_prepare_args)) return BNN_ERR_ABI;
  if (a->n_softplus < 0 || a->n_softplus > BNN_PREPARE_MAX || a->cast_n < 0) return BNN_ERR_SHAPE;
  PrepK k{};
  long blocks = 0;
  for (int i = 0; i < a->n_softplus; ++i) {
    if (!a->rho[i] || !a->sigma[i]) return BNN_ERR_NULL;
    if (a->n[i] <= 0) return BNN_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(a->rho[i]) | reinterpret_cast<uintptr_t>(a->sigma[i])) & 3) return BNN_ERR_ALIGN;
    k.rho[i] = a->rho[i]; k.sigma[i] = a->sigma[i]; k.n[i] = (long)a->n[i];
    k.first_block[i] = (int)blocks;
    blocks += (a->n[i] + kPrepPerBlock - 1) / kPrepPerBlock;
    if (blocks > 0x3fffffff) return BNN_ERR_SHAPE;
  }
  k.n_softplus = a->n_softplus;
  for (int i = a->n_softplus; i <= BNN_PREPARE_MAX; ++i) k.first_block[i] = (int)blocks;
  if (a->cast_n > 0) {
    if (!a->cast_src || !a->cast_dst) return BNN_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(a->cast_src) & 3) || (reinterpret_cast<uintptr_t>(a->cast_dst) & 1) ||
        (reinterpret_cast<uintptr_t>(a->cast_dst_sq) & 1) || (reinterpret_cast<uintptr_t>(a->cast_dst_lo) & 1))
      return BNN_ERR_ALIGN;
    k.cast_src = a->cast_src; k.cast_dst = reinterpret_cast<__bf16*>(a->cast_dst);
    k.cast_dsq = reinterpret_cast<__bf16*>(a->cast_dst_sq); k.cast_n = (long)a->cast_n;
    k.cast_dlo = reinterpret_cast<__bf16*>(a->cast_dst_lo);
    k.cast_vec = !((reinterpret_cast<uintptr_t>(a->cast_src) | reinterpret_cast<uintptr_t>(a->cast_dst) |
                    reinterpret_cast<uintptr_t>(a->cast_dst_sq) | reinterpret_cast<uintptr_t>(a->cast_dst_lo)) & 15);
    blocks += (a->cast_n + kPrepPerBlock - 1) / kPrepPerBlock;
    if (blocks > 0x3fffffff) return BNN_ERR_SHAPE;
  }
  if (blocks == 0) return BNN_ERR_SHAPE;
  hipLaunchKernelGGL(eval_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), k);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_version(void) { return BNN_HIP_ABI_VERSION; }
extern "C" int bnn_philox_rounds(void) { return BNN_PHILOX_ROUNDS; }

extern "C" const char* bnn_status_string(int status) {
  switch (status) {
    case BNN_OK: return "ok";
    case BNN_ERR_NULL: return "required pointer is NULL";
    case BNN_ERR_SHAPE: return "non-positive or unsupported dimension";
    case BNN_ERR_ENUM: return "unknown dtype / mode / prior kind";
    case BNN_ERR_WORKSPACE: return "workspace missing or too small";
    case BNN_ERR_ABI: return "struct_bytes mismatch (header/library version skew)";
    case BNN_ERR_ALIGN: return "pointer not aligned";
    default: return status > 0 ? "hip runtime error (status is a hipError_t)" : "unknown status";
  }
}
