// K2 gauss_kl (streaming KL / log-sigma reduction over (mu, rho)), K4 elbo_finalize
// (per-sample log p / log q / KL / NLL scalars of one ELBO evaluation), the Philox
// epsilon materialiser, and the library's version/status entry points.
#include "bnn_device.h"
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
struct FinK {
  const float* ws[8];
  int lin[8], lout[8];
  int n_layers, local_reparam, S, B, C;
  bnn_prior prior;
  const float* logits;
  const void* target;
  int nll_mode;
  float nll_sigma;
  float *log_prior, *log_q, *kl, *nll;
  uint32_t* sample_counter;
  uint32_t sample_counter_inc;
};

// grid = n_samples blocks (one sample each), or ONE block looping over all samples when
// `single` (small n_samples): then the block also writes the 4-vector of sums, in sample order.
// Latency-lean: every global load is issued before anything waits (headers, then partials and
// logits together), partial sums go through fp32 wave shuffles (each thread holds at most a
// few partials) and fp64 only across the 4 waves; all transcendental constants of the priors
// arrive precomputed from the host.
struct FinC {
  double cnt_c0[8];      // count * c0                                  (log q constant)
  double lp_const[8];    // count * (c0 - log sigma_p)                  (Gaussian log p constant)
  double kl_const[8];    // 0.5 * (2 count log sigma_p - count)         (LR)
  double inv2var;        // 1 / (2 sigma_p^2)
  double reg_const;      // log(nll_sigma) - c0                          (regression NLL per element)
  double reg_inv2var;    // 1 / (2 nll_sigma^2)
};

__global__ __launch_bounds__(256) void elbo_finalize_kernel(const FinK p, const FinC cst, int single, float* sums) {
  constexpr int NV = 25;                       // 3 sums x 8 layers + nll
  __shared__ float part[4 * NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int T[8];
#pragma unroll
  for (int l = 0; l < 8; ++l)
    T[l] = (l < p.n_layers) ? __float_as_int(reinterpret_cast<const float4*>(p.ws[l])[0].x) : 0;
  const int s_begin = single ? 0 : blockIdx.x, s_end = single ? p.S : blockIdx.x + 1;
  double tot_a = 0, tot_b = 0, tot_n = 0;
  for (int s = s_begin; s < s_end; ++s) {
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      if (l < p.n_layers) {
        const float4* ws = reinterpret_cast<const float4*>(p.ws[l]);
        for (int t = threadIdx.x; t < T[l]; t += blockDim.x) {
          if (p.local_reparam) {
            const float4 q = ws[1 + t];
            v[3 * l + 0] += q.x;                // sum log sigma
            v[3 * l + 1] += q.y;                // sum sigma^2
            v[3 * l + 2] += q.z;                // sum mu^2
          } else {
            const float4 q = ws[1 + (size_t)s * T[l] + t];
            v[3 * l + 0] += q.x;                // sum eps^2
            v[3 * l + 1] += q.y;                // sum w^2 | sum log p_mix
            v[3 * l + 2] += ws[1 + t].z;        // sum log sigma (stored with sample 0)
          }
        }
      }
    }
    if (p.nll && p.logits) {
      const float* lg = p.logits + (size_t)s * p.B * p.C;
      float acc = 0.f;
      if (p.nll_mode == BNN_NLL_CLASSIFICATION) {
        const long long* tgt = reinterpret_cast<const long long*>(p.target);
        for (int b = threadIdx.x; b < p.B; b += blockDim.x) {
          const float* row = lg + (size_t)b * p.C;
          float mx = row[0];
          for (int cc = 1; cc < p.C; ++cc) mx = fmaxf(mx, row[cc]);
          float se = 0.f;
          for (int cc = 0; cc < p.C; ++cc) se += expf(row[cc] - mx);
          const long long tc = tgt[b];
          const float picked = (tc >= 0 && tc < p.C) ? row[tc] : 0.f;
          acc += (mx + logf(se)) - picked;
        }
      } else {
        const float* tgt = reinterpret_cast<const float*>(p.target);
        const long tot = (long)p.B * p.C;
        for (long i = threadIdx.x; i < tot; i += blockDim.x) {
          const float d = tgt[i] - lg[i];
          acc += (float)((double)(d * d) * cst.reg_inv2var + cst.reg_const);
        }
      }
      v[NV - 1] = acc;
    }
    const int nv = 3 * p.n_layers;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < nv || i == NV - 1) {               // block-uniform
        const float t = wave_sum(v[i]);
        if (lane == 0) part[wave * NV + i] = t;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int nwv = blockDim.x >> 6;
      auto red = [&](int i) {
        double t = 0;
        for (int w = 0; w < nwv; ++w) t += (double)part[w * NV + i];
        return t;
      };
      // per-layer fp32 rounding, then fp32 adds, as the reference sums l1 + l2 + l3
      // (networks.py:174-181)
      float a_tot = 0.f, b_tot = 0.f;
      for (int l = 0; l < p.n_layers; ++l) {
        const double r0 = red(3 * l), r1 = red(3 * l + 1), r2 = red(3 * l + 2);
        if (p.local_reparam) {
          a_tot += (float)(cst.kl_const[l] - r0 + (r1 + r2) * cst.inv2var);
        } else {
          const double lq = cst.cnt_c0[l] - r2 - 0.5 * r0;
          const double lp = (p.prior.kind == BNN_PRIOR_GAUSS) ? cst.lp_const[l] - r1 * cst.inv2var : r1;
          a_tot += (float)lp;
          b_tot += (float)lq;
        }
      }
      const float nll = (float)red(NV - 1);
      if (p.local_reparam) {
        if (p.kl) p.kl[s] = a_tot;
      } else {
        if (p.log_prior) p.log_prior[s] = a_tot;
        if (p.log_q) p.log_q[s] = b_tot;
      }
      if (p.nll && p.logits) p.nll[s] = nll;
      tot_a += a_tot;
      tot_b += b_tot;
      tot_n += nll;
    }
    if (s + 1 < s_end) __syncthreads();
  }
  if (single && sums && threadIdx.x == 0) {
    sums[0] = (float)tot_a;
    sums[1] = (float)tot_b;
    sums[2] = (float)tot_n;
    sums[3] = (float)p.S;
  }
  if (p.sample_counter && blockIdx.x == 0 && threadIdx.x == 0) *p.sample_counter += p.sample_counter_inc;
}

// Sum of the per-sample scalars over the local samples, in index order per thread and a
// fixed tree across threads: the 4-vector a sharded job all-reduces.
__global__ void sample_sums_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                   const float* __restrict__ c, int S, float* __restrict__ sums) {
  __shared__ double scratch[16];
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
    sums[0] = (float)x;
    sums[1] = (float)y;
    sums[2] = (float)z;
    sums[3] = (float)S;
  }
}

// fp32 -> bf16 (RNE) cast of the input batch, once per ELBO evaluation, so that every layer of
// the throughput path reads 2-byte activations (the x tile is then LDS-DMA'd as is).
__global__ void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n, int vec_ok) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
  if (vec_ok) {
    const long n8 = n >> 3;
    for (long i = tid; i < n8; i += nt) {
      const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
      bf16x8 v;
      v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
      v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
      reinterpret_cast<bf16x8*>(dst)[i] = v;
    }
    for (long i = (n8 << 3) + tid; i < n; i += nt) dst[i] = (__bf16)src[i];
  } else {
    for (long i = tid; i < n; i += nt) dst[i] = (__bf16)src[i];
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

extern "C" int bnn_elbo_finalize(const bnn_finalize_args* a, void* stream_) {
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_finalize_args)) return BNN_ERR_ABI;
  if (a->n_layers < 0 || a->n_layers > 8 || a->n_samples <= 0 || a->n_samples > 65535) return BNN_ERR_SHAPE;
  if ((unsigned)a->prior.kind > 1u || (unsigned)a->nll_mode > 1u) return BNN_ERR_ENUM;
  FinK k;
  for (int l = 0; l < 8; ++l) {
    k.ws[l] = l < a->n_layers ? reinterpret_cast<const float*>(a->layer_workspace[l]) : nullptr;
    k.lin[l] = l < a->n_layers ? a->layer_in[l] : 0;
    k.lout[l] = l < a->n_layers ? a->layer_out[l] : 0;
    if (l < a->n_layers) {
      if (!k.ws[l]) return BNN_ERR_NULL;
      if (reinterpret_cast<uintptr_t>(k.ws[l]) & 15) return BNN_ERR_ALIGN;
      if (k.lin[l] <= 0 || k.lout[l] <= 0) return BNN_ERR_SHAPE;
    }
  }
  if (a->n_layers > 0 && !(a->prior.kind == BNN_PRIOR_MIXTURE) && !(a->prior.sigma_p > 0.f)) return BNN_ERR_SHAPE;
  if (a->nll) {
    if (!a->logits || !a->target) return BNN_ERR_NULL;
    if (a->batch <= 0 || a->classes <= 0) return BNN_ERR_SHAPE;
    if (a->nll_mode == BNN_NLL_REGRESSION && !(a->nll_sigma > 0.f)) return BNN_ERR_SHAPE;
  }
  k.n_layers = a->n_layers; k.local_reparam = a->local_reparam; k.S = a->n_samples; k.B = a->batch; k.C = a->classes;
  k.prior = a->prior; k.logits = a->logits; k.target = a->target; k.nll_mode = a->nll_mode;
  k.nll_sigma = a->nll_sigma; k.log_prior = a->log_prior; k.log_q = a->log_q; k.kl = a->kl; k.nll = a->nll;
  k.sample_counter = a->sample_counter; k.sample_counter_inc = a->sample_counter_inc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FinC cst{};
  const double c0 = -0.91893853320467274178;
  const double sp = (a->prior.kind == BNN_PRIOR_MIXTURE) ? 1.0 : (double)a->prior.sigma_p;
  for (int l = 0; l < a->n_layers; ++l) {
    const double cnt = (double)a->layer_out[l] * a->layer_in[l] + a->layer_out[l];
    cst.cnt_c0[l] = cnt * c0;
    cst.lp_const[l] = cnt * (c0 - log(sp));
    cst.kl_const[l] = 0.5 * (2.0 * cnt * log(sp) - cnt);
  }
  cst.inv2var = 1.0 / (2.0 * sp * sp);
  const double ns = a->nll_sigma > 0.f ? (double)a->nll_sigma : 1.0;
  cst.reg_const = log(ns) - c0;
  cst.reg_inv2var = 1.0 / (2.0 * ns * ns);
  const int single = a->n_samples <= 16;
  hipLaunchKernelGGL(elbo_finalize_kernel, dim3(single ? 1 : a->n_samples), dim3(256), 0, stream, k, cst, single,
                     a->sums);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  if (a->sums && !single) {
    const float* first = a->local_reparam ? a->kl : a->log_prior;
    const float* second = a->local_reparam ? nullptr : a->log_q;
    hipLaunchKernelGGL(sample_sums_kernel, dim3(1), dim3(256), 0, stream, first, second, a->nll, a->n_samples, a->sums);
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

extern "C" int bnn_cast_bf16(const float* src, void* dst, int64_t n, void* stream_) {
  if (!src || !dst) return BNN_ERR_NULL;
  if (n <= 0) return BNN_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(src) & 3) || (reinterpret_cast<uintptr_t>(dst) & 1)) return BNN_ERR_ALIGN;
  const int vec_ok = !((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15);
  long nb = (n / 8 + 255) / 256;
  nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), src,
                     reinterpret_cast<__bf16*>(dst), (long)n, vec_ok);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_version(void) { return BNN_HIP_ABI_VERSION; }

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
