// F1  backward of BayesianLinearLR (what autograd derives from reference networks.py:116-138
// under classification/class_task.py:78 `loss.backward()`), closed forms of SURVEY Appendix A.5.
//
// Forward (per MC sample s):  m = x M,  v = x^2 sigma^2,  y = m + sqrt(v) eps_act + b_mu + sigma_b eps_b.
// With gz_s = gy_s * (y_s > 0) and h_s = gz_s * eps_act_s / (2 sqrt(v_s))  (eps REGENERATED):
//     g_M      = sum_s x_s^T gz_s            + c_w M / sigma_p^2
//     g_sigma  = 2 sigma * sum_s (x_s^2)^T h_s + c_w (sigma / sigma_p^2 - 1 / sigma)
//     g_rho    = g_sigma * sigmoid(rho)          (bias: column sums of gz, times eps_b for sigma_b)
//     g_x[s]   = gz_s M^T + 2 x_s * (h_s (sigma^2)^T)
// c_w = g_kl[0] + g_kl[1], c_b = g_kl[0] + g_kl[2] are the upstream gradients of the layer's
// {kl, weight_kl, bias_kl} (the KL is the same for every sample, so it enters once).
//
// Three kernels, all on the exact-fp32 matrix core (v_mfma_f32_16x16x4_f32):
//  * lr_bwd_prep_kernel     elementwise: ReLU mask, eps_act regenerated per Philox group -> gz, h.
//  * lr_bwd_weights_kernel  both [in,out] GEMMs (reduction over samples x batch rows) in one pass
//                           over x / gz / h with the parameter update terms fused in the epilogue.
//                           A wave owns a 32 x 32 block as 2 x 2 tiles whose rows/cols interleave
//                           (tile i holds k = i mod 2, tile j holds n = j mod 2), so every operand
//                           is one 8-byte load along its contiguous dimension and x^2 is formed in
//                           registers; two waves split the batch rows of each block and swap
//                           accumulators through LDS (one finalises g_mu, the other g_rho).
//  * lr_bwd_input_kernel    both [batch,in] GEMMs (reduction over out features, the contiguous
//                           dimension of gz, h, M and sigma): a 16-byte load feeds 4 consecutive
//                           MFMA k-steps; 8 waves split the reduction and fold through LDS.
#include "bnn_device.h"
#include "../../include/bnn_hip.h"

namespace bnn {

struct LrBwdK {
  const float* x;        // [Sx, B, K]
  long x_sstride;
  const float* gz;       // [S, B, N]
  const float* h;        // [S, B, N]
  const float* w_mu;     // [K, N]
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  const float* eps_b;    // BNN_EPS_MEMORY: [S, N]
  const float* gkl;      // device float[3] or nullptr
  float* g_wmu;
  float* g_wrho;
  float* g_bmu;
  float* g_brho;
  float* g_x;            // [S, B, K]
  int S, B, K, N;
  int eps_mode;
  int gx_mask;           // g_x *= (x > 0): the ReLU of the layer below
  int h_factor;          // h holds the forward's hfac (eps_act / (2 sqrt(v))), gz the upstream gradient itself: h = gz * hfac on load
  uint32_t k0, k1, layer_id, sample_offset;
  const uint32_t* sample_counter;
  float inv_var_p;
};

__device__ __forceinline__ float lr_sigmoid(float r) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-r)); }

// gz = gy * (y > 0);  h = gz * eps_act / (2 sqrt(v))  (0 where v == 0, as the tensor-op form)
__global__ void lr_bwd_prep_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ v,
                                   const float* __restrict__ eps_act, float* __restrict__ gz, float* __restrict__ h, int S,
                                   int B, int N, int relu, int eps_mode, uint32_t k0, uint32_t k1, uint32_t layer_id,
                                   uint32_t sample_offset, const uint32_t* sample_counter) {
  if (sample_counter) sample_offset += *sample_counter;
  const int gpr = (N + 3) >> 2;
  const bool vec = (N & 3) == 0;
  const long total = (long)S * B * gpr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % gpr);
    const long sb = idx / gpr;
    const int b = (int)(sb % B), s = (int)(sb / B);
    const int nb = g * 4;
    const size_t row = ((size_t)s * B + b) * N;
    const size_t off = row + nb;
    // batch every load first, none under per-lane control flow (N % 4 == 0 is launch-uniform)
    float g4[4], y4[4] = {1.f, 1.f, 1.f, 1.f}, v4[4], e4[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      const float4 a = *reinterpret_cast<const float4*>(gy + off), c = *reinterpret_cast<const float4*>(v + off);
      g4[0] = a.x; g4[1] = a.y; g4[2] = a.z; g4[3] = a.w;
      v4[0] = c.x; v4[1] = c.y; v4[2] = c.z; v4[3] = c.w;
      if (relu) {
        const float4 d = *reinterpret_cast<const float4*>(y + off);
        y4[0] = d.x; y4[1] = d.y; y4[2] = d.z; y4[3] = d.w;
      }
      if (eps_mode == BNN_EPS_MEMORY) {
        const float4 e = *reinterpret_cast<const float4*>(eps_act + off);
        e4[0] = e.x; e4[1] = e.y; e4[2] = e.z; e4[3] = e.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t o = row + min(nb + i, N - 1);
        g4[i] = gy[o];
        v4[i] = v[o];
        if (relu) y4[i] = y[o];
        if (eps_mode == BNN_EPS_MEMORY) e4[i] = eps_act[o];
      }
    }
    if (eps_mode == BNN_EPS_PHILOX)
      philox_normal4((uint32_t)b * (uint32_t)gpr + (uint32_t)g, sample_offset + (uint32_t)s, layer_id * 4u + 2u, k0, k1, e4);
    float gz4[4], h4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g_ = y4[i] > 0.f ? g4[i] : 0.f;
      const float sd = __builtin_sqrtf(v4[i]);
      gz4[i] = g_;
      h4[i] = sd > 0.f ? g_ * e4[i] / (2.f * sd) : 0.f;
    }
    if (vec) {
      *reinterpret_cast<float4*>(gz + off) = make_float4(gz4[0], gz4[1], gz4[2], gz4[3]);
      *reinterpret_cast<float4*>(h + off) = make_float4(h4[0], h4[1], h4[2], h4[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nb + i < N) {
          gz[off + i] = gz4[i];
          h[off + i] = h4[i];
        }
    }
  }
}

// Two consecutive floats of a row starting at (even) index i, with NO per-lane control flow: a
// load under a divergent branch gets its own basic block and the waits between blocks serialise
// the whole batch (one round trip per load instead of one per batch).  Out-of-range indices are
// clamped, so the values are finite garbage that only reaches accumulator rows / columns which are
// never stored.  VEC: len is even, rows are 8-byte aligned.
template <bool VEC>
__device__ __forceinline__ float2 load_pair(const float* row, int i, int len) {
  if (VEC) return *reinterpret_cast<const float2*>(row + min(i, len - 2));
  float2 r;
  r.x = row[min(i, len - 1)];
  r.y = row[min(i + 1, len - 1)];
  return r;
}

// Each wave owns a 32 x 32 sub-tile (2 x 2 interleaved MFMA tiles) for HALF of the batch rows and accumulates both
// GEMMs over them; the two halves of a sub-tile then swap one accumulator set through LDS so that half 0 finalises
// g_w_mu (needs only M) and half 1 g_w_rho (needs only rho): twice the waves to hide the load latency of this
// short-reduction, wide-output GEMM, and a balanced epilogue.
// Block = 64 k x 32 n: (k half, batch-row half) waves.  (64 x 64 blocks of eight waves were 361 equal blocks on
// 256 CUs at the 1200 x 1200 layer -- some CUs two, most one; 722 half-size blocks land 3 : 2.)
template <bool VEC>
__global__ __launch_bounds__(256, 3) void lr_bwd_weights_kernel(const LrBwdK p) {
  __shared__ f32x4 xch[4][4][64];                        // 16 KiB: the accumulators a wave hands to its partner
  __shared__ float4 bxch[2][64];                         // bias partial sums of half 1
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int sub = wave & 1, half = wave >> 1;
  const int K = p.K, N = p.N, B = p.B;
  // XCD-aware order: an XCD walks a few out-feature blocks (its slice of gz, h) across all k
  // blocks, so x and that slice stay in its own L2
  const int nkb = (K + 63) >> 6;
  int item;
  const bool in_range = xcd_work_item(nkb * ((N + 31) >> 5), item);     // block-uniform
  const int kblk = in_range ? item % nkb : 0, nblk = in_range ? item / nkb : 0;
  const int kb = kblk * 64 + sub * 32;
  const int nb = nblk * 32;
  const bool active = in_range && kb < K && nb < N;      // wave-uniform; inactive waves only keep the barriers
  const int ka = kb + 2 * c;                             // A operand: k pair of this lane (tile i <-> ka + i)
  const int na = nb + 2 * c;                             // B operand: n pair of this lane (tile j <-> na + j)
  const int Bh = min(B, (((B + 1) >> 1) + 3) & ~3);      // rows [0, Bh) -> half 0, [Bh, B) -> half 1
  const int r_lo = half ? Bh : 0, r_hi = half ? B : Bh;

  f32x4 gM[2][2], gS[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      gM[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      gS[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const bool do_bias = active && kblk == 0 && sub == 0;
  const uint32_t sample_base = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u);
  float Gb[2] = {0.f, 0.f}, Hb[2] = {0.f, 0.f};

  if (active) {
    for (int s = 0; s < p.S; ++s) {
      const float* xs = p.x + (size_t)s * (size_t)p.x_sstride;
      const float* gzs = p.gz + (size_t)s * B * N;
      const float* hs = p.h + (size_t)s * B * N;
      float cs0 = 0.f, cs1 = 0.f;
      constexpr int U = 8;                                // batch-row quads in flight
      for (int b0 = r_lo; b0 < r_hi; b0 += 4 * U) {
        float2 av[U], gv[U], hv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int brow = b0 + 4 * u + q;
          const int rc = min(brow, B - 1);
          av[u] = load_pair<VEC>(xs + (size_t)rc * K, ka, K);
          gv[u] = load_pair<VEC>(gzs + (size_t)rc * N, na, N);
          hv[u] = load_pair<VEC>(hs + (size_t)rc * N, na, N);
          const bool mine = brow < r_hi;                  // rows of the other half / beyond the batch: zero
          gv[u].x = mine ? gv[u].x : 0.f;
          gv[u].y = mine ? gv[u].y : 0.f;
          hv[u].x = mine ? hv[u].x : 0.f;
          hv[u].y = mine ? hv[u].y : 0.f;
          if (p.h_factor) {
            hv[u].x *= gv[u].x;
            hv[u].y *= gv[u].y;
          }
        }
        __builtin_amdgcn_sched_barrier(0);                // the loads stay one batch ahead of the MFMAs
#pragma unroll
        for (int u = 0; u < U; ++u) {
          cs0 += gv[u].x;
          cs1 += gv[u].y;
          const float a0 = av[u].x, a1 = av[u].y;
          const float a0s = a0 * a0, a1s = a1 * a1;
          gM[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, gv[u].x, gM[0][0], 0, 0, 0);
          gM[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, gv[u].y, gM[0][1], 0, 0, 0);
          gM[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, gv[u].x, gM[1][0], 0, 0, 0);
          gM[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, gv[u].y, gM[1][1], 0, 0, 0);
          gS[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0s, hv[u].x, gS[0][0], 0, 0, 0);
          gS[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0s, hv[u].y, gS[0][1], 0, 0, 0);
          gS[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1s, hv[u].x, gS[1][0], 0, 0, 0);
          gS[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1s, hv[u].y, gS[1][1], 0, 0, 0);
        }
      }
      if (do_bias) {                                      // per-sample column sums of gz over this half's rows
        cs0 += __shfl_xor(cs0, 16, kWave);
        cs0 += __shfl_xor(cs0, 32, kWave);
        cs1 += __shfl_xor(cs1, 16, kWave);
        cs1 += __shfl_xor(cs1, 32, kWave);
        float e0 = 0.f, e1 = 0.f;
        if (p.eps_mode == BNN_EPS_PHILOX) {
          float e4[4];
          philox_normal4((uint32_t)(na >> 2), sample_base + (uint32_t)s, p.layer_id * 4u + 1u, p.k0, p.k1, e4);
          e0 = (na & 2) ? e4[2] : e4[0];                  // na is even: (na, na + 1) sit in one group of 4
          e1 = (na & 2) ? e4[3] : e4[1];
        } else if (p.eps_mode == BNN_EPS_MEMORY) {
          e0 = p.eps_b[(size_t)s * N + min(na, N - 1)];
          e1 = p.eps_b[(size_t)s * N + min(na + 1, N - 1)];
        }
        Gb[0] += cs0;
        Gb[1] += cs1;
        Hb[0] = __builtin_fmaf(cs0, e0, Hb[0]);
        Hb[1] = __builtin_fmaf(cs1, e1, Hb[1]);
      }
    }
  }

  // ---- swap: half 0 keeps gM and takes its partner's gM; half 1 keeps gS and takes its partner's gS
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) xch[wave][i * 2 + j][lane] = half ? gM[i][j] : gS[i][j];
  if (half) bxch[sub][lane] = make_float4(Gb[0], Gb[1], Hb[0], Hb[1]);
  __syncthreads();
  if (!active) return;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (half ? gS[i][j] : gM[i][j]) + xch[wave ^ 2][i * 2 + j][lane];

  // ---- epilogue: D row 4q + reg of tile i is k = kb + 2 (4q + reg) + i; D col c of tile j is n = na + j
  const float cw = p.gkl ? p.gkl[0] + p.gkl[1] : 0.f;
  const float* wsrc = half ? p.w_rho : p.w_mu;
  float* wdst = half ? p.g_wrho : p.g_wmu;
  float2 w[2][4];                                        // all parameter loads first: the stores below may
#pragma unroll                                           // alias them as far as the compiler knows
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int k = min(kb + 8 * q + 2 * reg + i, K - 1);
      w[i][reg] = load_pair<VEC>(wsrc + (size_t)k * N, na, N);
    }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int k = kb + 8 * q + 2 * reg + i;
      if (k >= K) continue;
      const size_t off = (size_t)k * N + na;
      float o[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float wv = j ? w[i][reg].y : w[i][reg].x;
        if (half) {
          const float sg = softplus(wv);
          const float gsig = 2.f * sg * acc[i][j][reg] + cw * (sg * p.inv_var_p - __builtin_amdgcn_rcpf(sg));
          o[j] = gsig * lr_sigmoid(wv);
        } else {
          o[j] = acc[i][j][reg] + cw * wv * p.inv_var_p;
        }
      }
      if (VEC && na < N) {
        *reinterpret_cast<float2*>(wdst + off) = make_float2(o[0], o[1]);
      } else {
        if (na < N) wdst[off] = o[0];
        if (na + 1 < N) wdst[off + 1] = o[1];
      }
    }
  }
  if (do_bias && half == 0 && q == 0) {
    const float cb = p.gkl ? p.gkl[0] + p.gkl[2] : 0.f;
    const float4 o = bxch[sub][lane];
    Gb[0] += o.x; Gb[1] += o.y; Hb[0] += o.z; Hb[1] += o.w;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = na + j;
      if (n < N) {
        const float bm = p.b_mu[n], br = p.b_rho[n];
        const float sb = softplus(br);
        p.g_bmu[n] = Gb[j] + cb * bm * p.inv_var_p;
        p.g_brho[n] = (Hb[j] + cb * (sb * p.inv_var_p - __builtin_amdgcn_rcpf(sb))) * lr_sigmoid(br);
      }
    }
  }
}

// Four consecutive floats of a row starting at index i (a multiple of 4), zero beyond `len`,
// without per-lane control flow (see load_pair).  VEC: len % 4 == 0, rows 16-byte aligned.
template <bool VEC>
__device__ __forceinline__ float4 load_quad(const float* row, int i, int len) {
  float4 r;
  if (VEC) {
    r = *reinterpret_cast<const float4*>(row + min(i, len - 4));
    const bool ok = i < len;
    r.x = ok ? r.x : 0.f; r.y = ok ? r.y : 0.f; r.z = ok ? r.z : 0.f; r.w = ok ? r.w : 0.f;
  } else {
    const float a = row[min(i, len - 1)], b2 = row[min(i + 1, len - 1)], c2 = row[min(i + 2, len - 1)],
                d = row[min(i + 3, len - 1)];
    r.x = i < len ? a : 0.f; r.y = i + 1 < len ? b2 : 0.f; r.z = i + 2 < len ? c2 : 0.f; r.w = i + 3 < len ? d : 0.f;
  }
  return r;
}

// One block per (32-wide k tile, 32-row batch block, sample), 8 waves.  Block tile 32 batch rows x 32 input features as
// 2 x 2 MFMA tiles for each of P = gz M^T and Q = h (sigma^2)^T; wave w takes the 16-wide
// slices w, w + 8, ... of the out-feature range.  MFMA k-step t of a slice uses reduction
// index n = 16 slice + 4 q + t on both operands, i.e. component t of one 16-byte load.
// BF16 (bf16 math, N % 8 == 0, 16-byte aligned rows): the operands are rounded to bf16 in registers and the slices
// are 32 wide (v_mfma_f32_16x16x32_bf16: lane (c, q) holds the 8 consecutive n = 32 slice + 8 q of its row), fp32
// accumulation -- the arithmetic of the forward's bf16 mode.  The exact-fp32 form spends 64 matrix-core
// instructions of 32 cycles per 32 n, this one 8 of 16: at 2 x 128 x 1200 x 1200 the kernel goes from the fp32 matrix
// core's bound to that of its loads.
template <bool VEC, bool BF16>
__global__ __launch_bounds__(512) void lr_bwd_input_kernel(const LrBwdK p) {
  __shared__ f32x4 red[4][8][64];                       // 32 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N, B = p.B;
  // XCD-aware order: the (batch block, sample) blocks of one 32-wide k tile are consecutive items, and an XCD owns a
  // contiguous range of items, so each XCD pulls ITS k tiles' (mu, rho) rows through its L2 once (a 3-D grid dealt the
  // k tiles round-robin: every XCD streamed all of [K, N] mu and rho, 8 x 11.5 MB over the fabric at 1200 x 1200)
  const int nbb = (B + 31) >> 5;
  int item;
  if (!xcd_work_item(((K + 31) >> 5) * nbb * p.S, item)) return;   // block-uniform, before any barrier
  const int kt = item / (nbb * p.S), rest = item - kt * (nbb * p.S);
  const int k0 = kt * 32, b0 = (rest % nbb) * 32, s = rest / nbb;
  const float* gzs = p.gz + (size_t)s * B * N;
  const float* hs = p.h + (size_t)s * B * N;
  const int nslices = (N + 15) >> 4;

  f32x4 P[2][2], Q[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      P[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      Q[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const int ra0 = min(b0 + c, B - 1), ra1 = min(b0 + 16 + c, B - 1);        // A rows (clamped: those D rows are not stored)
  const int kc0 = min(k0 + c, K - 1), kc1 = min(k0 + 16 + c, K - 1);        // B columns (likewise)
  struct Frag { float4 ga0, ga1, ha0, ha1, m0, m1, s0, s1; };
  auto load_slice = [&](int sl, Frag& f) {
    const int n = sl * 16 + 4 * q;                      // beyond N (past the last slice): clamped and zeroed
    f.ga0 = load_quad<VEC>(gzs + (size_t)ra0 * N, n, N);
    f.ga1 = load_quad<VEC>(gzs + (size_t)ra1 * N, n, N);
    f.ha0 = load_quad<VEC>(hs + (size_t)ra0 * N, n, N);
    f.ha1 = load_quad<VEC>(hs + (size_t)ra1 * N, n, N);
    f.m0 = load_quad<VEC>(p.w_mu + (size_t)kc0 * N, n, N);
    f.m1 = load_quad<VEC>(p.w_mu + (size_t)kc1 * N, n, N);
    f.s0 = load_quad<VEC>(p.w_rho + (size_t)kc0 * N, n, N);
    f.s1 = load_quad<VEC>(p.w_rho + (size_t)kc1 * N, n, N);
    if (p.h_factor) {
      f.ha0.x *= f.ga0.x; f.ha0.y *= f.ga0.y; f.ha0.z *= f.ga0.z; f.ha0.w *= f.ga0.w;
      f.ha1.x *= f.ga1.x; f.ha1.y *= f.ga1.y; f.ha1.z *= f.ga1.z; f.ha1.w *= f.ga1.w;
    }
  };
  auto sq_softplus = [](float r) { const float sg = softplus(r); return sg * sg; };
  auto mfma_slice = [&](Frag& f) {
    // sigma^2 from rho in place (a separate softplus pass over [K, N] was one more launch per layer)
    f.s0.x = sq_softplus(f.s0.x); f.s0.y = sq_softplus(f.s0.y); f.s0.z = sq_softplus(f.s0.z); f.s0.w = sq_softplus(f.s0.w);
    f.s1.x = sq_softplus(f.s1.x); f.s1.y = sq_softplus(f.s1.y); f.s1.z = sq_softplus(f.s1.z); f.s1.w = sq_softplus(f.s1.w);
#define LR_STEP(F)                                                                     \
    P[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ga0.F, f.m0.F, P[0][0], 0, 0, 0); \
    P[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ga0.F, f.m1.F, P[0][1], 0, 0, 0); \
    P[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ga1.F, f.m0.F, P[1][0], 0, 0, 0); \
    P[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ga1.F, f.m1.F, P[1][1], 0, 0, 0); \
    Q[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ha0.F, f.s0.F, Q[0][0], 0, 0, 0); \
    Q[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ha0.F, f.s1.F, Q[0][1], 0, 0, 0); \
    Q[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ha1.F, f.s0.F, Q[1][0], 0, 0, 0); \
    Q[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ha1.F, f.s1.F, Q[1][1], 0, 0, 0);
    LR_STEP(x)
    LR_STEP(y)
    LR_STEP(z)
    LR_STEP(w)
#undef LR_STEP
  };
  if constexpr (BF16) {
    struct Frag8 { float4 lo, hi; };
    auto load8 = [&](const float* row, int n) {
      Frag8 f;
      f.lo = *reinterpret_cast<const float4*>(row + min(n, N - 8));
      f.hi = *reinterpret_cast<const float4*>(row + min(n, N - 8) + 4);
      return f;
    };
    auto to_bf16 = [](const Frag8& f, bool ok) {
      const float v[8] = {f.lo.x, f.lo.y, f.lo.z, f.lo.w, f.hi.x, f.hi.y, f.hi.z, f.hi.w};
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)(ok ? v[j] : 0.f);
      return o;
    };
    auto to_var_bf16 = [&](const Frag8& f, bool ok) {
      const float v[8] = {f.lo.x, f.lo.y, f.lo.z, f.lo.w, f.hi.x, f.hi.y, f.hi.z, f.hi.w};
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)(ok ? sq_softplus(v[j]) : 0.f);
      return o;
    };
    const int nsteps = (N + 31) >> 5;
    struct Step { Frag8 g0, g1, h0, h1, m0, m1, r0, r1; };
    auto load_step = [&](int st, Step& f) {
      const int n = st * 32 + 8 * q;
      f.g0 = load8(gzs + (size_t)ra0 * N, n); f.g1 = load8(gzs + (size_t)ra1 * N, n);
      f.h0 = load8(hs + (size_t)ra0 * N, n); f.h1 = load8(hs + (size_t)ra1 * N, n);
      f.m0 = load8(p.w_mu + (size_t)kc0 * N, n); f.m1 = load8(p.w_mu + (size_t)kc1 * N, n);
      f.r0 = load8(p.w_rho + (size_t)kc0 * N, n); f.r1 = load8(p.w_rho + (size_t)kc1 * N, n);
    };
    auto times = [](const Frag8& a, const Frag8& b) {
      Frag8 r;
      r.lo = make_float4(a.lo.x * b.lo.x, a.lo.y * b.lo.y, a.lo.z * b.lo.z, a.lo.w * b.lo.w);
      r.hi = make_float4(a.hi.x * b.hi.x, a.hi.y * b.hi.y, a.hi.z * b.hi.z, a.hi.w * b.hi.w);
      return r;
    };
    auto mfma_step = [&](int st, const Step& f) {
      const bool ok = st * 32 + 8 * q < N;                // N % 8 == 0: an 8-group is whole or absent (also a step past the end)
      const bf16x8 ga0 = to_bf16(f.g0, ok), ga1 = to_bf16(f.g1, ok);
      const bf16x8 ha0 = to_bf16(p.h_factor ? times(f.h0, f.g0) : f.h0, ok), ha1 = to_bf16(p.h_factor ? times(f.h1, f.g1) : f.h1, ok);
      const bf16x8 mb0 = to_bf16(f.m0, ok), mb1 = to_bf16(f.m1, ok), sb0 = to_var_bf16(f.r0, ok), sb1 = to_var_bf16(f.r1, ok);
      P[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga0, mb0, P[0][0], 0, 0, 0);
      P[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga0, mb1, P[0][1], 0, 0, 0);
      P[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga1, mb0, P[1][0], 0, 0, 0);
      P[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga1, mb1, P[1][1], 0, 0, 0);
      Q[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha0, sb0, Q[0][0], 0, 0, 0);
      Q[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha0, sb1, Q[0][1], 0, 0, 0);
      Q[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha1, sb0, Q[1][0], 0, 0, 0);
      Q[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha1, sb1, Q[1][1], 0, 0, 0);
    };
    // (two steps' loads in flight per round measured slower: 36 against 30 us at 2 x 128 x 1200 x 1200)
    for (int st = wave; st < nsteps; st += 8) {
      Step f;
      load_step(st, f);
      mfma_step(st, f);
    }
  } else {
    // (issuing slice t + 1's loads ahead of slice t's MFMAs, or fencing the batch with a scheduling
    // barrier, both measured slower here: 45 us and 34 us against 31 us at 2 x 128 x 1200 x 1200)
    for (int sl = wave; sl < nslices; sl += 8) {
      Frag f;
      load_slice(sl, f);
      mfma_slice(f);
    }
  }

  // ---- fold the 8 partial sums: waves 4..7 -> LDS -> waves 0..3 add; then 0..3 -> LDS -> wave `combo` sums
  if (wave >= 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        red[wave - 4][i * 2 + j][lane] = P[i][j];
        red[wave - 4][4 + i * 2 + j][lane] = Q[i][j];
      }
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        P[i][j] += red[wave][i * 2 + j][lane];
        Q[i][j] += red[wave][4 + i * 2 + j][lane];
      }
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        red[wave][i * 2 + j][lane] = P[i][j];
        red[wave][4 + i * 2 + j][lane] = Q[i][j];
      }
  }
  __syncthreads();
  if (wave < 4) {
    const int combo = wave, ti = combo >> 1, tj = combo & 1;
    f32x4 ps = f32x4{0.f, 0.f, 0.f, 0.f}, qs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      ps += red[w][combo][lane];
      qs += red[w][4 + combo][lane];
    }
    const int k = k0 + 16 * tj + c;
    if (k < K) {
      const float* xs = p.x + (size_t)s * (size_t)p.x_sstride;
      float xv[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) xv[reg] = xs[(size_t)min(b0 + 16 * ti + 4 * q + reg, B - 1) * K + k];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int b = b0 + 16 * ti + 4 * q + reg;
        const float gx = __builtin_fmaf(2.f * xv[reg], qs[reg], ps[reg]);
        if (b < B) p.g_x[((size_t)s * B + b) * K + k] = (p.gx_mask && !(xv[reg] > 0.f)) ? 0.f : gx;
      }
    }
  }
}


// Backward of a NARROW LR output layer (N <= 16: the 10 classes / the 1 regression output, batch <= 128) in ONE launch,
// plain fp32 FMAs: the three general launches (prep, weights, input) are built for wide layers and cost 4.7 + 12 + 5.3 us
// at 1200 -> 10, mostly their launch and latency chains.  Every block forms the gz / h rows it needs itself (eps_act
// regenerated, as lr_bwd_prep_kernel does), so nothing passes through memory between the roles:
//   weight role (blocks < wblocks): 16 k rows of [K, N].  Thread (kc, bg) accumulates the batch rows b = bg (mod 16) of
//     g_M[k][n] = sum_{s,b} x gz and g_S[k][n] = sum_{s,b} x^2 h over all samples (the KL terms enter once, so no
//     per-sample epilogue), one LDS round adds the 16 row classes, thread (n, kc) applies the parameter terms; block 0
//     also takes the bias (column sums of gz, times eps_b for rho).
//   input role: a block owns 64 k x 32 (sample, row) pairs: M and sigma^2 of its k rows parked in LDS, thread (kc, rg)
//     walks 8 rows: g_x = (x > 0) * (sum_n gz[n] M[k][n] + 2 x sum_n h[n] sigma^2[k][n]).
struct LrOutBwd {
  const float* gy;      // [S, B, N] upstream gradient
  const float* y;       // forward output (ReLU mask) or nullptr
  const float* v;       // [S, B, N] the forward's variance
  int wblocks, kchunks, rchunks;
};

__device__ __forceinline__ void lr_out_rows(const LrBwdK& p, const LrOutBwd& o, int s, int b0, int nrows, uint32_t gs,
                                            float* gzs, float* hs) {
  // gz / h of rows b0 .. b0 + nrows - 1 of sample s into LDS as [row][16] (features padded with zeros); a thread takes
  // one (row, group of 4 features)
  const int N = p.N, B = p.B;
  const int gpr = (N + 3) >> 2;
  for (int i = threadIdx.x; i < nrows * 4; i += blockDim.x) {
    const int r = i >> 2, g = i & 3;
    const int b = b0 + r;
    float gz4[4] = {0.f, 0.f, 0.f, 0.f}, h4[4] = {0.f, 0.f, 0.f, 0.f};
    if (b < B && g < gpr) {
      const size_t row = ((size_t)s * B + b) * N;
      float e4[4];
      philox_normal4((uint32_t)b * (uint32_t)gpr + (uint32_t)g, gs, p.layer_id * 4u + 2u, p.k0, p.k1, e4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = g * 4 + j;
        if (n < N) {
          const float gyv = o.gy[row + n], vv = o.v[row + n];
          const float g_ = (!o.y || o.y[row + n] > 0.f) ? gyv : 0.f;
          const float sd = __builtin_sqrtf(vv);
          gz4[j] = g_;
          h4[j] = sd > 0.f ? g_ * e4[j] / (2.f * sd) : 0.f;
        }
      }
    }
    *reinterpret_cast<float4*>(gzs + r * 16 + g * 4) = make_float4(gz4[0], gz4[1], gz4[2], gz4[3]);
    *reinterpret_cast<float4*>(hs + r * 16 + g * 4) = make_float4(h4[0], h4[1], h4[2], h4[3]);
  }
}

// weight (and bias) gradients of 16 k rows: the weight role of lr_out_layer_bwd_kernel
template <bool PAIR>
__device__ __forceinline__ void lr_out_weight_role(const LrBwdK& p, const LrOutBwd& o, float* lds, uint32_t sample_base) {
  float* gzs = lds;
  float* hs = lds + 2 * 128 * 16;
  float (*red)[16 * 16 * 16] = reinterpret_cast<float (*)[16 * 16 * 16]>(lds);
  const int K = p.K, N = p.N, B = p.B, S = p.S;
  const int tid = threadIdx.x;
  const int k0 = (int)blockIdx.x * 16;
  const int kc = tid & 15, bg = tid >> 4;
  const int kx = min(k0 + kc, K - 1);
  const bool k_ok = k0 + kc < K;
  const bool b_ok = blockIdx.x == 0 && tid < N;            // bias: thread n
  float accM[16], accS[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) accM[n] = accS[n] = 0.f;
  float Gb = 0.f, Hb = 0.f;
  // PAIR (112 < batch <= 128: J = 8 rows per thread and sample, known at compile time): samples go through in rounds of
  // two -- their x loads and gz / h rows are one memory round trip
  const int J = PAIR ? 8 : (B + 15) >> 4;                   // rows per thread and sample (<= 8)
  constexpr int SP = PAIR ? 2 : 1;
  for (int s0 = 0; s0 < S; s0 += SP) {
    float xv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int sp = PAIR ? (j >> 3) : 0;
      const int b = bg + 16 * (j - sp * J);
      const int sm = min(s0 + sp, S - 1);
      const float v = p.x[(size_t)sm * (size_t)p.x_sstride + (size_t)min(b, B - 1) * K + kx];
      xv[j] = (j < SP * J && s0 + sp < S && b < B && k_ok) ? v : 0.f;
    }
    __syncthreads();                                        // the previous round's readers of gzs / hs are done
    lr_out_rows(p, o, s0, 0, B, sample_base + (uint32_t)s0, gzs, hs);
    if (PAIR && s0 + 1 < S) lr_out_rows(p, o, s0 + 1, 0, B, sample_base + (uint32_t)(s0 + 1), gzs + 128 * 16, hs + 128 * 16);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < SP * J) {
        const int sp = PAIR ? (j >> 3) : 0;
        const int b = min(bg + 16 * (j - sp * J), B - 1);
        const float x1 = xv[j], x2 = xv[j] * xv[j];         // (zero for the absent second sample of the last round)
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) {
          const float4 g = *reinterpret_cast<const float4*>(gzs + (sp * 128 + b) * 16 + n4 * 4);
          const float4 h = *reinterpret_cast<const float4*>(hs + (sp * 128 + b) * 16 + n4 * 4);
          accM[n4 * 4 + 0] = __builtin_fmaf(x1, g.x, accM[n4 * 4 + 0]); accM[n4 * 4 + 1] = __builtin_fmaf(x1, g.y, accM[n4 * 4 + 1]);
          accM[n4 * 4 + 2] = __builtin_fmaf(x1, g.z, accM[n4 * 4 + 2]); accM[n4 * 4 + 3] = __builtin_fmaf(x1, g.w, accM[n4 * 4 + 3]);
          accS[n4 * 4 + 0] = __builtin_fmaf(x2, h.x, accS[n4 * 4 + 0]); accS[n4 * 4 + 1] = __builtin_fmaf(x2, h.y, accS[n4 * 4 + 1]);
          accS[n4 * 4 + 2] = __builtin_fmaf(x2, h.z, accS[n4 * 4 + 2]); accS[n4 * 4 + 3] = __builtin_fmaf(x2, h.w, accS[n4 * 4 + 3]);
        }
      }
    }
    if (b_ok) {
      for (int sp = 0; sp < SP && s0 + sp < S; ++sp) {
        float cs = 0.f;
        for (int b = 0; b < B; ++b) cs += gzs[(sp * 128 + b) * 16 + tid];
        float e4[4];
        philox_normal4((uint32_t)(tid >> 2), sample_base + (uint32_t)(s0 + sp), p.layer_id * 4u + 1u, p.k0, p.k1, e4);
        const float e = (tid & 3) == 0 ? e4[0] : (tid & 3) == 1 ? e4[1] : (tid & 3) == 2 ? e4[2] : e4[3];
        Gb += cs;
        Hb = __builtin_fmaf(cs, e, Hb);
      }
    }
  }
  __syncthreads();                                          // gzs / hs are dead: their memory takes the partial sums
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    red[0][(bg * 16 + n) * 16 + kc] = accM[n];
    red[1][(bg * 16 + n) * 16 + kc] = accS[n];
  }
  __syncthreads();
  {
    const int n = tid >> 4, c = tid & 15;
    const int k = k0 + c;
    if (n < N && k < K) {
      float gm = 0.f, gsv = 0.f;
      for (int g = 0; g < 16; ++g) {
        gm += red[0][(g * 16 + n) * 16 + c];
        gsv += red[1][(g * 16 + n) * 16 + c];
      }
      const float cw = p.gkl ? p.gkl[0] + p.gkl[1] : 0.f;
      const size_t off = (size_t)k * N + n;
      const float m = p.w_mu[off], r = p.w_rho[off];
      const float sg = softplus(r);
      p.g_wmu[off] = gm + cw * m * p.inv_var_p;
      p.g_wrho[off] = (2.f * sg * gsv + cw * (sg * p.inv_var_p - __builtin_amdgcn_rcpf(sg))) * lr_sigmoid(r);
    }
  }
  if (b_ok) {
    const float cb = p.gkl ? p.gkl[0] + p.gkl[2] : 0.f;
    const float bm = p.b_mu[tid], br = p.b_rho[tid];
    const float sb = softplus(br);
    p.g_bmu[tid] = Gb + cb * bm * p.inv_var_p;
    p.g_brho[tid] = (Hb + cb * (sb * p.inv_var_p - __builtin_amdgcn_rcpf(sb))) * lr_sigmoid(br);
  }
}

template <bool PAIR>
__global__ __launch_bounds__(256) void lr_out_layer_bwd_kernel(const LrBwdK p, const LrOutBwd o) {
  // weight role: gz / h rows of the round's (up to two) samples, then -- the same memory -- the partial sums
  // [M | S][bg][n][kc]; input role: gz / h of the block's rows and the M, sigma^2 rows of its k chunk
  __shared__ __attribute__((aligned(16))) float lds[2 * 16 * 16 * 16];     // 32 KiB
  float* gzs = lds;                                         // [2 x 128 rows][16]
  float* hs = lds + 2 * 128 * 16;
  float (*red)[16 * 16 * 16] = reinterpret_cast<float (*)[16 * 16 * 16]>(lds);
  const int K = p.K, N = p.N, B = p.B, S = p.S;
  const int tid = threadIdx.x;
  const uint32_t sample_base = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u);
  if ((int)blockIdx.x >= o.wblocks) {
    // ---- input gradient: 64 k x 32 rows of the flattened (sample, row) list
    const int bi = (int)blockIdx.x - o.wblocks;
    const int kch = bi % o.kchunks, rch = bi / o.kchunks;
    const int k0 = kch * 64;
    float* Ms = lds + 1024;                                 // [64][16] each, between the 32-row gz area (512 floats at 0)
    float* Vs = lds + 2048;                                 // and the h area (at 4096)
    for (int i = tid; i < 64 * 16; i += 256) {
      const int kk = i >> 4, n = i & 15;
      const int k = k0 + kk;
      const bool ok = k < K && n < N;
      const float m = p.w_mu[(size_t)min(k, K - 1) * N + min(n, N - 1)], r = p.w_rho[(size_t)min(k, K - 1) * N + min(n, N - 1)];
      const float sg = softplus(r);
      Ms[i] = ok ? m : 0.f;
      Vs[i] = ok ? sg * sg : 0.f;
    }
    const int row0 = rch * 32;                              // rows of one sample (B is split into whole 32-row chunks per sample)
    const int cps = (B + 31) >> 5;                          // chunks per sample
    const int s = rch / cps, b0 = (rch - s * cps) * 32;
    lr_out_rows(p, o, s, b0, 32, sample_base + (uint32_t)s, gzs, hs);
    (void)row0;
    __syncthreads();
    const int kc = tid & 63, rg = tid >> 6;
    const int k = k0 + kc;
    if (k >= K) return;
    float mrow[16], vrow[16];
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) {
      const float4 a = *reinterpret_cast<const float4*>(Ms + kc * 16 + n4 * 4), c = *reinterpret_cast<const float4*>(Vs + kc * 16 + n4 * 4);
      mrow[n4 * 4 + 0] = a.x; mrow[n4 * 4 + 1] = a.y; mrow[n4 * 4 + 2] = a.z; mrow[n4 * 4 + 3] = a.w;
      vrow[n4 * 4 + 0] = c.x; vrow[n4 * 4 + 1] = c.y; vrow[n4 * 4 + 2] = c.z; vrow[n4 * 4 + 3] = c.w;
    }
    const float* xs = p.x + (size_t)s * (size_t)p.x_sstride;
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = xs[(size_t)min(b0 + rg + 4 * j, B - 1) * K + k];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = rg + 4 * j, b = b0 + r;
      float pm = 0.f, qv = 0.f;
#pragma unroll
      for (int n4 = 0; n4 < 4; ++n4) {
        const float4 g = *reinterpret_cast<const float4*>(gzs + r * 16 + n4 * 4), h = *reinterpret_cast<const float4*>(hs + r * 16 + n4 * 4);
        pm = __builtin_fmaf(g.x, mrow[n4 * 4 + 0], pm); pm = __builtin_fmaf(g.y, mrow[n4 * 4 + 1], pm);
        pm = __builtin_fmaf(g.z, mrow[n4 * 4 + 2], pm); pm = __builtin_fmaf(g.w, mrow[n4 * 4 + 3], pm);
        qv = __builtin_fmaf(h.x, vrow[n4 * 4 + 0], qv); qv = __builtin_fmaf(h.y, vrow[n4 * 4 + 1], qv);
        qv = __builtin_fmaf(h.z, vrow[n4 * 4 + 2], qv); qv = __builtin_fmaf(h.w, vrow[n4 * 4 + 3], qv);
      }
      const float gx = __builtin_fmaf(2.f * xv[j], qv, pm);
      if (b < B) p.g_x[((size_t)s * B + b) * K + k] = (p.gx_mask && !(xv[j] > 0.f)) ? 0.f : gx;
    }
    return;
  }
  // ---- weight (and bias) gradients of 16 k rows
  lr_out_weight_role<PAIR>(p, o, lds, sample_base);
}

}  // namespace bnn

using namespace bnn;

static size_t lr_bwd_align(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t bnn_lr_linear_bwd_workspace_bytes(int32_t n_samples, int32_t batch, int32_t in_features,
                                                    int32_t out_features, int32_t want_gx) {
  if (n_samples <= 0 || batch <= 0 || in_features <= 0 || out_features <= 0) return 0;
  const size_t act = lr_bwd_align((size_t)n_samples * batch * out_features * sizeof(float));
  (void)in_features; (void)want_gx;          // gz and h; sigma is recomputed from rho where it is used
  return 2 * act;
}

extern "C" int bnn_lr_linear_bwd(const bnn_lr_bwd_args* a, void* stream_) {
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_lr_bwd_args)) return BNN_ERR_ABI;
  if (a->n_samples <= 0 || a->batch <= 0 || a->in_features <= 0 || a->out_features <= 0) return BNN_ERR_SHAPE;
  if (a->n_samples > 65535) return BNN_ERR_SHAPE;
  if (!a->x || !a->gy || (!a->v && !a->hfac) || !a->w_mu || !a->w_rho || !a->b_mu || !a->b_rho || !a->g_w_mu || !a->g_w_rho ||
      !a->g_b_mu || !a->g_b_rho)
    return BNN_ERR_NULL;
  if ((unsigned)a->eps_mode > 2u || (unsigned)a->math > 1u) return BNN_ERR_ENUM;
  if (a->eps_mode == BNN_EPS_MEMORY && (!a->eps_act || !a->eps_b)) return BNN_ERR_NULL;
  if (a->relu && !a->y) return BNN_ERR_NULL;
  if (!(a->sigma_p > 0.f)) return BNN_ERR_SHAPE;
  const int S = a->n_samples, B = a->batch, K = a->in_features, N = a->out_features;
  const bool factor = a->hfac && !a->relu;               // no preparation launch, no workspace
  if (a->hfac && a->relu && !a->v) return BNN_ERR_NULL;  // a fused ReLU needs the mask pass, and that pass v
  if (!factor) {
    if (!a->workspace || a->workspace_bytes < bnn_lr_linear_bwd_workspace_bytes(S, B, K, N, a->g_x ? 1 : 0))
      return BNN_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(a->workspace) & 15) return BNN_ERR_ALIGN;
  }
  const uintptr_t al = reinterpret_cast<uintptr_t>(a->x) | reinterpret_cast<uintptr_t>(a->w_mu) |
                       reinterpret_cast<uintptr_t>(a->w_rho) | reinterpret_cast<uintptr_t>(a->g_w_mu) |
                       reinterpret_cast<uintptr_t>(a->g_w_rho) | reinterpret_cast<uintptr_t>(a->gy) |
                       reinterpret_cast<uintptr_t>(a->v) | reinterpret_cast<uintptr_t>(a->y) | reinterpret_cast<uintptr_t>(a->hfac) |
                       reinterpret_cast<uintptr_t>(a->eps_act);
  if (al & 15) return BNN_ERR_ALIGN;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const size_t act = lr_bwd_align((size_t)S * B * N * sizeof(float));
  char* base = reinterpret_cast<char*>(a->workspace);
  float* gz = reinterpret_cast<float*>(base);
  float* h = reinterpret_cast<float*>(base + act);

  const uint32_t k0 = (uint32_t)a->seed, k1 = (uint32_t)(a->seed >> 32);
  if (N <= 16 && B <= 128 && a->g_x && a->eps_mode == BNN_EPS_PHILOX && a->v) {
    // narrow output layer: prep, both weight gradients and the input gradient in one launch
    LrBwdK kk;
    kk.x = a->x;
    kk.x_sstride = a->x_per_sample ? (long)B * K : 0;
    kk.gz = nullptr; kk.h = nullptr;
    kk.w_mu = a->w_mu; kk.w_rho = a->w_rho; kk.b_mu = a->b_mu; kk.b_rho = a->b_rho;
    kk.eps_b = nullptr; kk.gkl = a->g_kl;
    kk.g_wmu = a->g_w_mu; kk.g_wrho = a->g_w_rho; kk.g_bmu = a->g_b_mu; kk.g_brho = a->g_b_rho; kk.g_x = a->g_x;
    kk.S = S; kk.B = B; kk.K = K; kk.N = N;
    kk.eps_mode = a->eps_mode; kk.k0 = k0; kk.k1 = k1; kk.layer_id = a->layer_id; kk.sample_offset = a->sample_offset;
    kk.sample_counter = a->sample_counter;
    kk.gx_mask = a->gx_relu_mask ? 1 : 0;
    kk.h_factor = 0;
    kk.inv_var_p = (float)(1.0 / ((double)a->sigma_p * a->sigma_p));
    LrOutBwd o;
    o.gy = a->gy; o.y = a->relu ? a->y : nullptr; o.v = a->v;
    o.wblocks = (K + 15) / 16;
    o.kchunks = (K + 63) / 64;
    o.rchunks = S * ((B + 31) / 32);
    const long nblk = (long)o.wblocks + (long)o.kchunks * o.rchunks;
    if (nblk < (1L << 30)) {
      // one sample per round (PAIR = false).  The two-samples-per-round instantiation is not used: with both samples'
      // accumulation unrolled the compiler keeps 512 VGPRs and spills.  (The same loop written for one sample only --
      // 8 instead of 16 row slots per thread -- measured 3 us SLOWER, 15.3 against 12.1 us; kept as measured.)
      hipLaunchKernelGGL(lr_out_layer_bwd_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, stream, kk, o);
      const hipError_t e1 = hipGetLastError();
      return e1 == hipSuccess ? BNN_OK : (int)e1;
    }
  }
  const long groups = (long)S * B * ((N + 3) / 4);
  long nb = (groups + 255) / 256;
  nb = nb > 4096 ? 4096 : nb;
  hipError_t err = hipSuccess;
  if (!factor) {
    if (!a->v) return BNN_ERR_NULL;
    hipLaunchKernelGGL(lr_bwd_prep_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a->gy, a->y, a->v, a->eps_act, gz, h, S, B,
                       N, a->relu ? 1 : 0, a->eps_mode, k0, k1, a->layer_id, a->sample_offset, a->sample_counter);
    err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }

  LrBwdK k;
  k.x = a->x;
  k.x_sstride = a->x_per_sample ? (long)B * K : 0;
  k.gz = factor ? a->gy : gz;
  k.h = factor ? a->hfac : h;
  k.h_factor = factor ? 1 : 0;
  k.w_mu = a->w_mu; k.w_rho = a->w_rho; k.b_mu = a->b_mu; k.b_rho = a->b_rho;
  k.eps_b = a->eps_b; k.gkl = a->g_kl;
  k.g_wmu = a->g_w_mu; k.g_wrho = a->g_w_rho; k.g_bmu = a->g_b_mu; k.g_brho = a->g_b_rho; k.g_x = a->g_x;
  k.S = S; k.B = B; k.K = K; k.N = N;
  k.eps_mode = a->eps_mode; k.k0 = k0; k.k1 = k1; k.layer_id = a->layer_id; k.sample_offset = a->sample_offset;
  k.sample_counter = a->sample_counter;
  k.gx_mask = a->gx_relu_mask ? 1 : 0;
  k.inv_var_p = (float)(1.0 / ((double)a->sigma_p * a->sigma_p));
  const int wblocks = ((K + 63) / 64) * ((N + 31) / 32);
  const dim3 wgrid((unsigned)(((wblocks + 7) / 8) * 8));
  if (((K | N) & 1) == 0) hipLaunchKernelGGL(lr_bwd_weights_kernel<true>, wgrid, dim3(256), 0, stream, k);
  else hipLaunchKernelGGL(lr_bwd_weights_kernel<false>, wgrid, dim3(256), 0, stream, k);
  err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  if (a->g_x) {
    const long items = (long)((K + 31) / 32) * ((B + 31) / 32) * S;
    if (items > (1L << 30)) return BNN_ERR_SHAPE;
    const dim3 igrid((unsigned)(((items + 7) / 8) * 8));
    if (a->math == BNN_MATH_BF16 && (N & 7) == 0) hipLaunchKernelGGL((lr_bwd_input_kernel<true, true>), igrid, dim3(512), 0, stream, k);
    else if ((N & 3) == 0) hipLaunchKernelGGL((lr_bwd_input_kernel<true, false>), igrid, dim3(512), 0, stream, k);
    else hipLaunchKernelGGL((lr_bwd_input_kernel<false, false>), igrid, dim3(512), 0, stream, k);
    err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  return BNN_OK;
}
