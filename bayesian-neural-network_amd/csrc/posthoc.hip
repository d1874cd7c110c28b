// F3 / F4: the two post-hoc analyses of the reference that walk every stochastic parameter or every predicted
// probability, as device kernels (HBM-bound streaming passes, no float atomics: bitwise reproducible).
//   bnn_ece          compute_ece.py:14-57   (ECELoss.forward: reliability bins over ALL N x C probabilities)
//   bnn_snr_db       weight_pruning.py:85-87 (compute_snr), :100, :109
//   bnn_snr_prune    weight_pruning.py:89-115 (prune_weights: mu, rho *= (snr > threshold))
#include "bnn_device.h"
#include "../../include/bnn_hip.h"

namespace bnn {

constexpr int kEceMaxBins = 64;
constexpr int kEceBlock = 256;
constexpr int kEceMaxBlocks = 256;

struct EceEdges {
  double e[kEceMaxBins + 1];   // np.arange(0, 1 + step, step): passed by value so the comparison is the reference's
  int n_edges;                 // float64 one
};

// Per block: counts / corrects / confidence sums of its rows, one row (= C probabilities) per thread at a time.
// Reference semantics (compute_ece.py:22-44): every one of the C probabilities of a row is binned
// (np.digitize(p, bins, right=True) - 1: bin k holds bins[k] < p <= bins[k+1]; p == 0 falls out), a probability
// counts as "correct" when it is the row's argmax AND that argmax is the label.
__global__ __launch_bounds__(kEceBlock) void ece_partial_kernel(const float* __restrict__ probs,
                                                                const long long* __restrict__ labels, long n, int C,
                                                                const EceEdges ed, double* __restrict__ part) {
  __shared__ unsigned int s_cnt[kEceMaxBins], s_cor[kEceMaxBins];
  __shared__ double s_conf[kEceBlock / 64][kEceMaxBins];
  const int nb = ed.n_edges - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kEceMaxBins; i += blockDim.x) s_cnt[i] = s_cor[i] = 0u;
  for (int i = threadIdx.x; i < (kEceBlock / 64) * kEceMaxBins; i += blockDim.x) (&s_conf[0][0])[i] = 0.0;
  __syncthreads();
  // rows are dealt to blocks in contiguous chunks, to a wave's lanes one by one in row order: the confidence sums are
  // accumulated per (wave, bin) by ONE lane at a time in lane order below, so the fp64 summation order is fixed
  const long per_block = (n + gridDim.x - 1) / gridDim.x;
  const long r0 = (long)blockIdx.x * per_block, r1 = min(n, r0 + per_block);
  for (long base = r0 + (long)wave * 64; base < r1; base += blockDim.x) {
    const long row = base + lane;
    const bool ok = row < r1;
    int arg = 0;
    float best = -1.f;
    if (ok) {
      for (int c = 0; c < C; ++c) {
        const float v = probs[row * C + c];
        if (v > best) { best = v; arg = c; }           // first maximum, as np.argmax
      }
    }
    const bool hit = ok && (long long)arg == labels[ok ? row : 0];
    for (int c = 0; c < C; ++c) {
      int bin = -1;
      double pv = 0.0;
      if (ok) {
        pv = (double)probs[row * C + c];
        int less = 0;
        for (int j = 0; j < ed.n_edges; ++j) less += ed.e[j] < pv ? 1 : 0;
        bin = less - 1;
        if (bin >= nb) bin = -1;
      }
      if (bin >= 0) {
        atomicAdd(&s_cnt[bin], 1u);                      // integers: exact in any order
        if (hit && c == arg) atomicAdd(&s_cor[bin], 1u);
      }
      // fp64 confidence sums: lane after lane (fixed order), no float atomics
      for (int l = 0; l < 64; ++l) {
        const int b_l = __shfl(bin, l, 64);
        const double p_l = __shfl(pv, l, 64);
        if (lane == 0 && b_l >= 0) s_conf[wave][b_l] += p_l;
      }
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
    double cs = 0.0;
    for (int w = 0; w < kEceBlock / 64; ++w) cs += s_conf[w][b];
    double* o = part + ((size_t)blockIdx.x * kEceMaxBins + b) * 3;
    o[0] = (double)s_cnt[b];
    o[1] = (double)s_cor[b];
    o[2] = cs;
  }
}

// out = {ece, then per bin: count, corrects, confidence sum}.  Bins without data contribute nothing (the reference's
// loop indexes a compressed accuracy array and a NaN mean there: it only runs when every bin has data).
__global__ void ece_final_kernel(const double* __restrict__ part, int nblocks, int nb, float* __restrict__ out) {
  __shared__ double cnt[kEceMaxBins], cor[kEceMaxBins], conf[kEceMaxBins];
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
    double a = 0, c = 0, s = 0;
    for (int k = 0; k < nblocks; ++k) {
      const double* o = part + ((size_t)k * kEceMaxBins + b) * 3;
      a += o[0]; c += o[1]; s += o[2];
    }
    cnt[b] = a; cor[b] = c; conf[b] = s;
    out[1 + 3 * b + 0] = (float)a;
    out[1 + 3 * b + 1] = (float)c;
    out[1 + 3 * b + 2] = (float)s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double total = 0, ece = 0;
    for (int b = 0; b < nb; ++b) total += cnt[b];
    for (int b = 0; b < nb; ++b)
      if (cnt[b] > 0) ece += fabs(conf[b] / cnt[b] - cor[b] / cnt[b]) * cnt[b] / total;
    out[0] = (float)ece;
  }
}

// 10 * log10(|mu| / softplus(rho)) in fp32, as torch evaluates weight_pruning.py:100 / :109 (softplus = log1p(exp(rho)))
__device__ __forceinline__ float snr_db(float mu, float rho) {
  return 10.0f * (__builtin_amdgcn_logf(fabsf(mu) / softplus(rho)) * 0.30102999566398120f);   // log2 -> log10
}

__global__ __launch_bounds__(256) void snr_db_kernel(const float* __restrict__ mu, const float* __restrict__ rho, long n,
                                                     int vec_ok, float* __restrict__ out) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
  if (vec_ok) {
    for (long i = tid; i < (n >> 2); i += nt) {
      const float4 m = reinterpret_cast<const float4*>(mu)[i], r = reinterpret_cast<const float4*>(rho)[i];
      reinterpret_cast<float4*>(out)[i] = make_float4(snr_db(m.x, r.x), snr_db(m.y, r.y), snr_db(m.z, r.z), snr_db(m.w, r.w));
    }
    for (long i = ((n >> 2) << 2) + tid; i < n; i += nt) out[i] = snr_db(mu[i], rho[i]);
  } else {
    for (long i = tid; i < n; i += nt) out[i] = snr_db(mu[i], rho[i]);
  }
}

// in place: mu *= mask, rho *= mask with mask = snr > threshold (a pruned weight keeps rho = 0, i.e. sigma = log 2,
// exactly as the reference's `weight_rhos*mask` leaves it)
__global__ __launch_bounds__(256) void snr_prune_kernel(float* __restrict__ mu, float* __restrict__ rho, long n, int vec_ok,
                                                        float threshold, unsigned long long* __restrict__ kept) {
  __shared__ unsigned long long s_kept[4];
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
  unsigned int mine = 0;
  auto one = [&](float& m, float& r) {
    const bool keep = snr_db(m, r) > threshold;
    m = keep ? m : m * 0.0f;                             // * 0 keeps the reference's signed zeros / NaN propagation
    r = keep ? r : r * 0.0f;
    mine += keep ? 1u : 0u;
  };
  if (vec_ok) {
    for (long i = tid; i < (n >> 2); i += nt) {
      float4 m = reinterpret_cast<float4*>(mu)[i], r = reinterpret_cast<float4*>(rho)[i];
      one(m.x, r.x); one(m.y, r.y); one(m.z, r.z); one(m.w, r.w);
      reinterpret_cast<float4*>(mu)[i] = m;
      reinterpret_cast<float4*>(rho)[i] = r;
    }
    for (long i = ((n >> 2) << 2) + tid; i < n; i += nt) one(mu[i], rho[i]);
  } else {
    for (long i = tid; i < n; i += nt) one(mu[i], rho[i]);
  }
  if (kept) {                                            // integer count: exact in any order
    unsigned long long w = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
    if ((threadIdx.x & 63) == 0) s_kept[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(kept, s_kept[0] + s_kept[1] + s_kept[2] + s_kept[3]);
  }
}

}  // namespace bnn

using namespace bnn;

static inline int stream_blocks(long n) {
  long b = (n / 4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" size_t bnn_ece_workspace_bytes(void) { return (size_t)kEceMaxBlocks * kEceMaxBins * 3 * sizeof(double); }

extern "C" int bnn_ece(const float* probs, const long long* labels, int64_t n, int32_t classes, const double* bin_edges,
                       int32_t n_edges, void* workspace, size_t workspace_bytes, float* out, void* stream_) {
  if (!probs || !labels || !bin_edges || !out) return BNN_ERR_NULL;
  if (n <= 0 || classes <= 0 || n_edges < 2 || n_edges > kEceMaxBins + 1) return BNN_ERR_SHAPE;
  if (!workspace || workspace_bytes < bnn_ece_workspace_bytes()) return BNN_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 7) return BNN_ERR_ALIGN;
  EceEdges ed;
  for (int i = 0; i <= kEceMaxBins; ++i) ed.e[i] = i < n_edges ? bin_edges[i] : 0.0;   // HOST array: copied by value
  ed.n_edges = n_edges;
  long nb = (n + 1023) / 1024;
  nb = nb < 1 ? 1 : (nb > kEceMaxBlocks ? kEceMaxBlocks : nb);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(ece_partial_kernel, dim3((unsigned)nb), dim3(kEceBlock), 0, stream, probs, labels, (long)n, classes, ed,
                     reinterpret_cast<double*>(workspace));
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(ece_final_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const double*>(workspace), (int)nb,
                     n_edges - 1, out);
  err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_snr_db(const float* mu, const float* rho, int64_t n, float* out, void* stream_) {
  if (!mu || !rho || !out) return BNN_ERR_NULL;
  if (n <= 0) return BNN_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(rho) | reinterpret_cast<uintptr_t>(out)) & 3) return BNN_ERR_ALIGN;
  const int vec_ok = !((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(rho) | reinterpret_cast<uintptr_t>(out)) & 15);
  hipLaunchKernelGGL(snr_db_kernel, dim3((unsigned)stream_blocks((long)n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                     mu, rho, (long)n, vec_ok, out);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_snr_prune(float* mu, float* rho, int64_t n, float threshold, unsigned long long* kept, void* stream_) {
  if (!mu || !rho) return BNN_ERR_NULL;
  if (n <= 0) return BNN_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(rho)) & 3) return BNN_ERR_ALIGN;
  if (kept && (reinterpret_cast<uintptr_t>(kept) & 7)) return BNN_ERR_ALIGN;
  const int vec_ok = !((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(rho)) & 15);
  hipLaunchKernelGGL(snr_prune_kernel, dim3((unsigned)stream_blocks((long)n)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream_), mu, rho, (long)n, vec_ok, threshold, kept);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}
