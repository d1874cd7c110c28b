// K1  bbb_linear_fwd — BayesianLinear.forward (reference networks.py:73-88) for all locally
// owned MC samples of one layer in one launch.
//
// Lane geometry (both kernels).  A wave owns a 16-row MFMA A-operand tile of sampled
// weights.  Lane (r = lane&15, q = lane>>4) holds, per 32-deep k-step, the 8 consecutive
// weights W[n][k0 + 8q .. +7] of ONE output feature n — exactly the A fragment of
// v_mfma_f32_16x16x32_bf16 — so mu/rho arrive as two 16-byte loads each, eps is generated
// in that layout (2 Philox calls = 8 normals), w = mu + sigma*eps is formed in registers,
// folded into the fp32 log-prob partial sums, rounded to bf16 and fed to the matrix core.
// w never exists in memory.  B operand = x[16m + r][k0 + 8q .. +7] for the 8 batch tiles m
// of the block's 128 rows.  D[row][col = batch row r]: a lane ends with 4 consecutive output
// features of one batch row -> one 16-byte (fp32) / 8-byte (bf16) store.
//
// `bbb_fwd_kernel<MATH, XDT, R, ALIGNED>`: grid (tiles, samples, ceil(batch/128)).
//   * The 16 A rows are split into R k-range classes of F = 16/R features: row r carries
//     feature f = r % F and k-range class c = r / F, i.e. the tile covers F features x R*32
//     k per "super-step".  Per batch tile the wave issues R MFMAs, the c-th with the A rows
//     of the other classes zeroed and the x fragment of k-range c.  The D rows of one
//     feature are summed in the epilogue.  R > 1 makes tiles narrower (150 / 300 tiles for
//     1200 features) so a single-sample launch still covers the chip WITHOUT any cross-block
//     reduction (no atomics, no split-K slabs: bitwise reproducible for a given shape).
//   * A block's NW waves split the super-steps (wave w owns w, w+NW, ...); NW is chosen so
//     they divide evenly.  Per step a wave first issues its x-fragment loads (unconditional
//     16-byte loads at clamped addresses, one batch), prefetches the NEXT step's (mu, rho),
//     then runs softplus + Philox + Box-Muller + fma + stats, then the MFMAs.
//   * The NW partial accumulators meet in LDS (8 KiB per wave), then bias + ReLU +
//     down-conversion run as the epilogue.
//   ALIGNED = false is the any-shape variant (odd K, K = 1, unaligned views): guarded loads.
//
// Stats workspace (opaque to callers): float4 ws[1 + S*T]; ws[0] = {T as int bits,0,0,0}
// written by block 0, then entry (s, t) = {sum eps^2, sum w^2 | sum log p_mix(w),
// sum log sigma (s = 0 only), 0} for the features of tile t.
#include "bnn_device.h"
#include "bnn_fin.h"
#include "../../include/bnn_hip.h"
#include <string.h>

namespace bnn {

struct BbbK {
  const void* x;
  long x_sstride;   // elements between samples of x (0: shared)
  const float* w_mu;
  const float* w_rho;
  const float* w_sigma;   // optional: softplus(w_rho) precomputed once per evaluation (bnn_softplus)
  const float* b_mu;
  const float* b_rho;
  const float* eps_w;
  const float* eps_b;
  float* eps_w_dump;
  float* eps_b_dump;
  void* y;
  float4* ws;       // stats workspace (see above) or nullptr
  int S, B, K, N;
  int eps_mode, prior_kind, want_stats, relu, y_bf16, spb;
  int ksl;          // GEMM form: K-range slices per (tile group, sample); 1 = none
  float* ks_part;   // GEMM form, ksl > 1: fp32 partial outputs [ksl][S][B][N] (bias in slice 0)
  int ldw;          // TRANS only: leading dimension of the [out,in] weight matrix (= original in_features)
  const __bf16* w_pre;  // PRE only: sampled weights bf16 [S, N, K] (bnn_bbb_sample_weights); no sampling in the launch
  const float* b_pre;   // PRE only: sampled biases [S, N]
  const float* mask;  // TRANS only, optional: [S|1, B, N] the layer's input; the stored gx is multiplied by (mask > 0)
  long mask_sstride;
  uint32_t k0, k1, layer_id, sample_offset;
  const uint32_t* sample_counter;
  float inv2var1, c1, inv2var2, c2, pi;   // mixture: log N(w;0,s_i) = c_i - w^2 * inv2var_i
#ifdef BNN_STAMPS
  unsigned long long* dbg;   // diagnostic build only: [block][16] shader-clock stamps of wave 0
#endif
};

#ifdef BNN_STAMPS
#define BNN_STAMP(i)                                                                              \
  do {                                                                                            \
    if (p.dbg && threadIdx.x == 0)                                                                \
      p.dbg[((size_t)blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define BNN_STAMP_RT(i)                                                                           \
  do {                                                                                            \
    if (p.dbg && threadIdx.x == 0)                                                                \
      p.dbg[((size_t)blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define BNN_STAMP(i)
#define BNN_STAMP_RT(i)
#endif

template <bool ALIGNED>
__device__ __forceinline__ void load8(const float* __restrict__ p, int valid, float v[8]) {
  if (ALIGNED) {
    if (valid > 0) {
      const float4 a = *reinterpret_cast<const float4*>(p);
      const float4 b = *reinterpret_cast<const float4*>(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
      v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (j < valid) ? p[j] : 0.0f;
  }
}

template <bool ALIGNED>
__device__ __forceinline__ void store8(float* __restrict__ p, int valid, const float v[8]) {
  if (ALIGNED) {
    if (valid > 0) {
      *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < valid) p[j] = v[j];
  }
}

// x fragment for one batch tile: 8 consecutive k of one row, as bf16x8 (bf16 math) or 8 floats.
template <int MATH, int XDT, bool ALIGNED>
__device__ __forceinline__ void load_xfrag(const char* __restrict__ xs, long off, int valid, float xv[8], bf16x8& xb) {
  if (XDT == BNN_F32) {
    load8<ALIGNED>(reinterpret_cast<const float*>(xs) + off, valid, xv);
    if (MATH == BNN_MATH_BF16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[j] = (__bf16)xv[j];
    }
  } else {
    const __bf16* p = reinterpret_cast<const __bf16*>(xs) + off;
    if (ALIGNED) {
      if (valid > 0) {
        xb = *reinterpret_cast<const bf16x8*>(p);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xb[j] = (__bf16)0.0f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[j] = (j < valid) ? p[j] : (__bf16)0.0f;
    }
    if (MATH == BNN_MATH_F32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = (float)xb[j];
    }
  }
}

// the mixture density itself (argument of the log), for add_log
__device__ __forceinline__ float mix_p(const BbbK& p, float w) {
  const float w2 = w * w;
  const float p1 = fast_exp(__builtin_fmaf(-w2, p.inv2var1, p.c1));
  const float p2 = fast_exp(__builtin_fmaf(-w2, p.inv2var2, p.c2));
  return __builtin_fmaf(p.pi, p1, (1.0f - p.pi) * p2);
}

// Sampled bias of feature n for global sample gs (+ its stats); returns b.
// eps of the bias of feature n for global sample gs.
__device__ __forceinline__ float bias_eps(const BbbK& p, int n, int s, uint32_t gs, bool dump) {
  float e = 0.f;
  if (p.eps_mode == BNN_EPS_PHILOX) {
    float e4[4];
    philox_normal4((uint32_t)(n >> 2), gs, p.layer_id * 4u + 1u, p.k0, p.k1, e4);
    e = (n & 3) == 0 ? e4[0] : (n & 3) == 1 ? e4[1] : (n & 3) == 2 ? e4[2] : e4[3];
  } else if (p.eps_mode == BNN_EPS_MEMORY) {
    e = p.eps_b[(size_t)s * p.N + n];
  }
  if (p.eps_b_dump && dump) p.eps_b_dump[(size_t)s * p.N + n] = e;
  return e;
}

__device__ __forceinline__ float sample_bias(const BbbK& p, float bmu, float brho, float e, bool do_stats, bool do_ls,
                                             float& s_e2, float& s_a, float& s_ls) {
  const float sig = softplus(brho);
  const float b = __builtin_fmaf(sig, e, bmu);
  if (do_stats) {
    s_e2 = __builtin_fmaf(e, e, s_e2);
    s_a = (p.prior_kind == BNN_PRIOR_GAUSS) ? __builtin_fmaf(b, b, s_a) : add_log(s_a, mix_p(p, b));
    if (do_ls) s_ls = add_log(s_ls, sig);
  }
  return b;
}

// Epilogue: sum the NW slabs (and the R k-range classes of a feature), + bias, ReLU,
// convert, store 4 consecutive features per item.
template <int R, int MT = 8>
__device__ __forceinline__ void epilogue_store(const BbbK& p, const f32x4* __restrict__ slab,
                                               const float* __restrict__ lds_bias, int nw, int mtiles, int nt, int s,
                                               int m0, float* __restrict__ lds_out = nullptr,
                                               float* __restrict__ raw_out = nullptr) {
  constexpr int F = 16 / R, FG = F / 4;
  const int N = p.N, B = p.B;
  const bool vec_ok = (N & 3) == 0;
  for (int item = threadIdx.x; item < mtiles * 16 * FG; item += blockDim.x) {
    const int m = item / (16 * FG);
    const int rem = item - m * (16 * FG);
    const int fg = rem >> 4, b = rem & 15;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int wv = 0; wv < nw; ++wv) {          // fixed order: wave-major, class-minor
#pragma unroll
      for (int c = 0; c < R; ++c) v += slab[(wv * MT + m) * 64 + (c * FG + fg) * 16 + b];
    }
    const int brow = m0 + m * 16 + b;
    const int nb = nt * F + fg * 4;
    if (raw_out) {                                 // K-slice partial of the fused last layer: no ReLU yet
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += lds_bias[fg * 4 + i];
      *reinterpret_cast<f32x4*>(raw_out + (m * 16 + b) * 16 + fg * 4) = v;
      continue;
    }
    if (brow >= B || nb >= N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o = v[i] + lds_bias[fg * 4 + i];
      if (p.relu) o = fmaxf(o, 0.f);
      v[i] = o;
    }
    if (lds_out) *reinterpret_cast<f32x4*>(lds_out + (m * 16 + b) * 16 + fg * 4) = v;
    if (p.mask) {                                  // input gradient through the ReLU of the layer below
      const float* mp = p.mask + (size_t)s * (size_t)p.mask_sstride + (size_t)brow * N + nb;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nb + i < N && !(mp[i] > 0.f)) v[i] = 0.f;
    }
    const size_t yoff = ((size_t)s * B + brow) * N + nb;
    if (p.y_bf16) {
      __bf16* yp = reinterpret_cast<__bf16*>(p.y) + yoff;
      if (vec_ok) {
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
        *reinterpret_cast<bf16x4*>(yp) = o;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (nb + i < N) yp[i] = (__bf16)v[i];
      }
    } else {
      float* yp = reinterpret_cast<float*>(p.y) + yoff;
      if (vec_ok) {
        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (nb + i < N) yp[i] = v[i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// ALIGNED = true : K % 8 == 0 and 16-byte aligned bases; every load is an unconditional
//                  16-byte access at a clamped address, issued in batches.
// ALIGNED = false: any K / alignment (guarded scalar loads); R must be 1.
struct FinPack {
  FinK k;
  FinC c;
  float* sums;          // float[4] or nullptr
  uint32_t* ticket;     // zero-initialised word: arrival counter of the sample blocks
  int ks;               // K-range slices per sample (blocks per sample); 1: no cross-block stage
  uint32_t* ks_ticket;  // [S] zero-initialised arrival counters of a sample's slice blocks
  float4* ks_stats;     // [S*ks] per-slice {sum eps^2, sum w^2|log p_mix, sum log sigma, 0}
  float* ks_tiles;      // [S*ks][128*16] per-slice partial logits (fp32, bias in slice 0)
};

// FINAL: this launch is the last layer of an ELBO evaluation (one 16-feature tile, batch <= 128):
// the block of sample s finishes with that sample's NLL and log p / log q, i.e. the work of
// bnn_elbo_finalize, without another launch.
// TRANS: the same kernel computes the input gradient gx = gz . w (reduction over the ORIGINAL
// output features): kernel "feature" n = original input index, kernel reduction index k =
// original output index, weight element (k, n) lives at w[k * ldw + n]; its eps is slot n & 3 of
// Philox group (k, n >> 2).  No bias, no statistics.
// Returns false for the padding blocks of the XCD-aware grid (no work done).  `forced_item` >= 0 runs that
// work item whatever the block index is (the fused-tail kernel hands the last layer to whichever block
// finished the layer before it last).
// MT: 16-row batch tiles per block (8 = 128 rows).  The matmul-only forms use 2: without generator work a block's
// cost is the x it pulls through its CU's L1 (all of K for its rows), so four times the blocks each ingest a quarter;
// with sampling fused every batch block would redo the tile's sampling, hence 8 there.
template <int MATH, int XDT, int R, bool ALIGNED, bool FINAL, bool TRANS = false, bool PRE = false, int MT = 8>
__device__ __forceinline__ bool bbb_fwd_body(const BbbK& p, const FinPack* fp, int forced_item = -1) {
  constexpr int F = 16 / R;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int c = r / F, f = r % F;
  const int K = p.K, N = p.N, B = p.B;
  const int ntiles = (N + F - 1) / F, mbs = (B + 16 * MT - 1) / (16 * MT);
  int item;
  const int KS = FINAL ? fp->ks : 1;                          // K-range slices per sample (FINAL only)
  if (forced_item >= 0) item = forced_item;
  else if (!xcd_work_item(ntiles * p.S * mbs * KS, item)) return false;   // block-uniform
  const int ks = item % KS;
  item /= KS;
  const int nt = item / (p.S * mbs), s = (item / mbs) % p.S, mb = item % mbs;
  const int n = nt * F + f;
  const bool n_ok = n < N;
  const int nc = min(n, N - 1);                     // clamped feature for unconditional loads
  const int m0 = mb * (16 * MT);
  const int mtiles = min(MT, (B - m0 + 15) >> 4);
  const int ssteps = (K + 32 * R - 1) / (32 * R);
  const int spb = (ssteps + KS - 1) / KS;                      // super-steps per K-range slice
  const int t_lo = ks * spb, t_hi = min(ssteps, t_lo + spb);
  const uint32_t gs = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u) + (uint32_t)s;
  const bool do_stats = p.want_stats && mb == 0;
  const bool do_ls = do_stats && (s == 0 || FINAL);
  const bool do_dump = mb == 0;
  const int gpr = (K + 3) >> 2;
  const uint32_t wid = p.layer_id * 4u;
  const char* xs = reinterpret_cast<const char*>(p.x) + (size_t)s * (size_t)p.x_sstride * (XDT == BNN_F32 ? 4 : 2);

  float* lds_bias = lds + (size_t)nw * MT * 64 * 4;   // 16 floats
  float* lds_red = lds_bias + 16;                    // 3 * nw floats
  f32x4* slab = reinterpret_cast<f32x4*>(lds);

  BNN_STAMP(0);
  BNN_STAMP_RT(8);
  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s_e2 = 0.f, s_a = 0.f, s_ls = 0.f;
  if (do_stats && item == 0 && ks == 0 && threadIdx.x == 0)
    p.ws[0] = make_float4(__int_as_float(ntiles), 0.f, 0.f, 0.f);

  // (mu, rho) of this lane for super-step t: prefetched one step ahead.
  float mu_n[8], rho_n[8];
  float4 wraw_n = make_float4(0.f, 0.f, 0.f, 0.f);    // PRE: the lane's 8 sampled bf16 weights of the step
  auto load_params = [&](int t) {
    const int k = (t * R + c) * 32 + q * 8;
    if (PRE && TRANS) {                             // input gradient over the forward's sampled weights [S, K, ldw]
      bf16x8 tw;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 v = p.w_pre[((size_t)s * K + min(k + j, K - 1)) * p.ldw + nc];
        tw[j] = (k + j < K && n_ok) ? v : (__bf16)0.f;
      }
      wraw_n = __builtin_bit_cast(float4, tw);
    } else if (PRE) {
      wraw_n = *reinterpret_cast<const float4*>(p.w_pre + ((size_t)s * N + nc) * K + min(k, K - 8));
    } else if (TRANS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t woff = (size_t)min(k + j, K - 1) * p.ldw + nc;
        mu_n[j] = p.w_mu[woff];
        rho_n[j] = p.w_rho[woff];
      }
    } else if (ALIGNED) {
      const size_t woff = (size_t)nc * K + min(k, K - 8);
      load8<true>(p.w_mu + woff, 8, mu_n);
      load8<true>(p.w_rho + woff, 8, rho_n);
    } else {
      const int valid = n_ok ? min(8, K - k) : 0;
      const size_t woff = (size_t)n * K + k;
      load8<false>(p.w_mu + woff, valid, mu_n);
      load8<false>(p.w_rho + woff, valid, rho_n);
    }
  };
  if (t_lo + wave < t_hi) load_params(t_lo + wave);
  // bias parameters of the tile (used after the k-loop): fetched now, off the critical path
  float bmu_pre = 0.f, brho_pre = 0.f, beps_pre = 0.f;
  if (PRE) {
    if (!TRANS && wave == nw - 1 && lane < F && n_ok && ks == 0) bmu_pre = p.b_pre[(size_t)s * N + n];
  } else if (!TRANS && wave == nw - 1 && lane < F && n_ok && ks == 0) {   // the last wave owns the fewest k-steps
    bmu_pre = p.b_mu[n];
    brho_pre = p.b_rho[n];
    beps_pre = bias_eps(p, n, s, gs, do_dump);
  }

#pragma nounroll
  for (int t = t_lo + wave; t < t_hi; t += nw) {
    const int k = (t * R + c) * 32 + q * 8;
    const int valid = n_ok ? min(8, K - k) : 0;     // ALIGNED: 8 or <= 0
    // ---- x fragments of the first batch-tile chunk, issued ahead of the generator work.
    constexpr int FR = (XDT == BNN_F32) ? 2 : 1;                  // 16-byte loads per fragment
    constexpr int MC0 = (8 / (R * FR)) < 1 ? 1 : (8 / (R * FR));
    constexpr int MC = MC0 > MT ? MT : MC0;                       // batch tiles staged at once
    float4 xraw[MC * R * FR];
    auto stage = [&](int ch) {
#pragma unroll
      for (int mm = 0; mm < MC; ++mm) {
        const int row = m0 + (ch * MC + mm) * 16 + r;
#pragma unroll
        for (int cc = 0; cc < R; ++cc) {
          const int xk = (t * R + cc) * 32 + q * 8;
          if (ALIGNED) {
            const size_t off = (size_t)min(row, B - 1) * K + min(xk, K - 8);
            if (XDT == BNN_F32) {
              const float4* px = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xs) + off);
              xraw[(mm * R + cc) * 2 + 0] = px[0];
              xraw[(mm * R + cc) * 2 + 1] = px[1];
            } else {
              xraw[mm * R + cc] = *reinterpret_cast<const float4*>(reinterpret_cast<const __bf16*>(xs) + off);
            }
          } else {
            const int xvalid = (row < B) ? min(8, K - xk) : 0;
            const size_t off = (size_t)row * K + xk;
            if (XDT == BNN_F32) {
              float v[8];
              load8<false>(reinterpret_cast<const float*>(xs) + off, xvalid, v);
              xraw[(mm * R + cc) * 2 + 0] = make_float4(v[0], v[1], v[2], v[3]);
              xraw[(mm * R + cc) * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
            } else {
              bf16x8 vb;
#pragma unroll
              for (int j = 0; j < 8; ++j)
                vb[j] = (j < xvalid) ? (reinterpret_cast<const __bf16*>(xs) + off)[j] : (__bf16)0.0f;
              xraw[mm * R + cc] = __builtin_bit_cast(float4, vb);
            }
          }
        }
      }
    };
    stage(0);

    // ---- consume the prefetched (mu, rho), prefetch the next step's
    float mu[8], sg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = mu_n[j];
      sg[j] = rho_n[j];
    }
    const float4 wraw = wraw_n;
    if (t + nw < t_hi) load_params(t + nw);
    if (t == t_lo + wave) { asm volatile("" :: "v"(mu[0]), "v"(sg[0])); BNN_STAMP(1); }

    float e[8], w[8];
    if (PRE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = w[j] = 0.f;
    } else if (TRANS) {
      const uint32_t gprw = (uint32_t)((p.ldw + 3) >> 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e[j] = 0.f;
        if (p.eps_mode == BNN_EPS_PHILOX) {
          float e4[4];
          philox_normal4((uint32_t)(k + j) * gprw + (uint32_t)(n >> 2), gs, wid, p.k0, p.k1, e4);
          e[j] = (n & 3) == 0 ? e4[0] : (n & 3) == 1 ? e4[1] : (n & 3) == 2 ? e4[2] : e4[3];
        } else if (p.eps_mode == BNN_EPS_MEMORY) {
          if (j < valid) e[j] = p.eps_w[((size_t)s * K + k + j) * p.ldw + n];
        }
      }
    } else if (p.eps_mode == BNN_EPS_PHILOX) {
      const uint32_t g = (uint32_t)n * (uint32_t)gpr + (uint32_t)(k >> 2);
      philox_normal4(g, gs, wid, p.k0, p.k1, e);
      philox_normal4(g + 1u, gs, wid, p.k0, p.k1, e + 4);
    } else if (p.eps_mode == BNN_EPS_MEMORY) {
      load8<ALIGNED>(p.eps_w + ((size_t)s * N + n) * K + k, valid, e);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = 0.f;
    }
    if (!TRANS && p.eps_w_dump && do_dump) store8<ALIGNED>(p.eps_w_dump + ((size_t)s * N + n) * K + k, valid, e);

    // ALIGNED: a lane's 8 weights are all real or all padding (valid is 8 or <= 0), so one mask
    // per step suffices: padded lanes compute on clamped (finite) data and are zeroed at the end.
    float e2 = 0.f, a = 0.f, ls = 0.f;
    if (!PRE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = ALIGNED ? true : (j < valid);
        sg[j] = softplus(sg[j]);
        w[j] = ok ? __builtin_fmaf(sg[j], e[j], mu[j]) : 0.f;
        e2 = ok ? __builtin_fmaf(e[j], e[j], e2) : e2;
      }
    }
    if (do_stats && !PRE) {
      if (p.prior_kind == BNN_PRIOR_GAUSS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a = __builtin_fmaf(w[j], w[j], a);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) a = (ALIGNED || j < valid) ? add_log(a, mix_p(p, w[j])) : a;
      }
      if (do_ls) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ls = (ALIGNED || j < valid) ? add_log(ls, sg[j]) : ls;
      }
      const bool lane_ok = !ALIGNED || valid > 0;
      s_e2 += lane_ok ? e2 : 0.f;
      s_a += lane_ok ? a : 0.f;
      s_ls += lane_ok ? ls : 0.f;
    }
    if (ALIGNED && valid <= 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = 0.f;
    }
    if (t == t_lo + wave) { asm volatile("" :: "v"(w[0]), "v"(w[7])); BNN_STAMP(2); }
    bf16x8 wa, wz;
    if (MATH == BNN_MATH_BF16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        wa[j] = (__bf16)w[j];
        wz[j] = (__bf16)0.f;
      }
      if (PRE && (TRANS || valid > 0)) wa = __builtin_bit_cast(bf16x8, wraw);   // padding (k >= K, n >= N) is zero
    }
#pragma unroll
    for (int ch = 0; ch < MT / MC; ++ch) {
      if (ch * MC < mtiles) {                     // block-uniform
#pragma unroll
        for (int mm = 0; mm < MC; ++mm) {
#pragma unroll
          for (int cc = 0; cc < R; ++cc) {
            const int m = ch * MC + mm;
            if (MATH == BNN_MATH_BF16) {
              bf16x8 xb;
              if (XDT == BNN_F32) {
                const float4 lo = xraw[(mm * R + cc) * 2], hi = xraw[(mm * R + cc) * 2 + 1];
                xb[0] = (__bf16)lo.x; xb[1] = (__bf16)lo.y; xb[2] = (__bf16)lo.z; xb[3] = (__bf16)lo.w;
                xb[4] = (__bf16)hi.x; xb[5] = (__bf16)hi.y; xb[6] = (__bf16)hi.z; xb[7] = (__bf16)hi.w;
              } else {
                xb = __builtin_bit_cast(bf16x8, xraw[mm * R + cc]);
              }
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((R == 1 || c == cc) ? wa : wz, xb, acc[m], 0, 0, 0);
            } else {
              float xv[8];
              if (XDT == BNN_F32) {
                const float4 lo = xraw[(mm * R + cc) * 2], hi = xraw[(mm * R + cc) * 2 + 1];
                xv[0] = lo.x; xv[1] = lo.y; xv[2] = lo.z; xv[3] = lo.w;
                xv[4] = hi.x; xv[5] = hi.y; xv[6] = hi.z; xv[7] = hi.w;
              } else {
                const bf16x8 xb = __builtin_bit_cast(bf16x8, xraw[mm * R + cc]);
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = (float)xb[j];
              }
#pragma unroll
              for (int j = 0; j < 8; ++j)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32((R == 1 || c == cc) ? w[j] : 0.f, xv[j], acc[m], 0, 0, 0);
            }
          }
        }
        if ((ch + 1) * MC < MT && (ch + 1) * MC < mtiles) stage(ch + 1);
      }
    }
  }

  asm volatile("" :: "v"(acc[0][0]), "v"(acc[MT - 1][3]));
  BNN_STAMP(3);
  // ---- bias of the tile's F features: wave 0, lanes 0..F-1
  if (wave == nw - 1 && lane < 16) {
    float b = 0.f;
    if (PRE) b = (!TRANS && lane < F && n_ok && ks == 0) ? bmu_pre : 0.f;
    else if (!TRANS && lane < F && n_ok && ks == 0) b = sample_bias(p, bmu_pre, brho_pre, beps_pre, do_stats, do_ls, s_e2, s_a, s_ls);
    lds_bias[lane] = b;
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
    if (m < mtiles) slab[(wave * MT + m) * 64 + lane] = acc[m];
  if (do_stats) {
    const float a = wave_sum(s_e2), b = wave_sum(s_a), cc = wave_sum(s_ls);
    if (lane == 0) {
      lds_red[wave * 3 + 0] = a;
      lds_red[wave * 3 + 1] = b;
      lds_red[wave * 3 + 2] = cc;
    }
  }
  BNN_STAMP(4);
  __syncthreads();
  BNN_STAMP(5);
  float own0 = 0.f, own1 = 0.f, own2 = 0.f;
  if (do_stats && threadIdx.x == 0) {
    for (int wv = 0; wv < nw; ++wv) {
      own0 += lds_red[wv * 3 + 0];
      own1 += lds_red[wv * 3 + 1];
      own2 += lds_red[wv * 3 + 2];
    }
    if (KS == 1) p.ws[1 + (size_t)s * ntiles + nt] = make_float4(own0, own1, own2, 0.f);
  }
  float* fin_lg = lds_red + 3 * nw;               // FINAL only: [128][16] final logits + reduce scratch
  if (!FINAL || KS == 1) {
    epilogue_store<R, MT>(p, slab, lds_bias, nw, mtiles, nt, s, m0, FINAL ? fin_lg : nullptr);
  }
  BNN_STAMP(6);
  BNN_STAMP_RT(9);
  if (FINAL) {
    const FinK& fk = fp->k;
    float* part = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(fin_lg + 128 * 16) + 7) & ~(uintptr_t)7);   // holds doubles too
    int T[8];
#pragma unroll
    for (int l = 0; l < 8; ++l)
      T[l] = (l < fk.n_layers - 1) ? __float_as_int(reinterpret_cast<const float4*>(fk.ws[l])[0].x) : 0;
    if (KS > 1) {
      // ---- K-range slices of one sample meet here (placement-independent hand-off: every storing
      // wave drains its stores, one lane releases at agent scope, takes a ticket; the last
      // arriver acquires, then sums the slices' partial tiles in slice order).
      float* mine = fp->ks_tiles + ((size_t)s * KS + ks) * (128 * 16);
      epilogue_store<R>(p, slab, lds_bias, nw, 8, nt, s, m0, nullptr, mine);
      if (threadIdx.x == 0) fp->ks_stats[(size_t)s * KS + ks] = make_float4(own0, own1, own2, 0.f);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      uint32_t* flag = reinterpret_cast<uint32_t*>(part);          // LDS broadcast of "I am last"
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t tk = __hip_atomic_fetch_add(fp->ks_ticket + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = (tk == (uint32_t)KS - 1u) ? 1u : 0u;
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = last;
      }
      __syncthreads();
      BNN_STAMP(7);
      if (*flag == 0u) return true;                               // block-uniform
      __syncthreads();
      const float* tiles = fp->ks_tiles + (size_t)s * KS * (128 * 16);
      float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);                 // the slices' statistics: requested together with
      if ((int)threadIdx.x < KS) q4 = fp->ks_stats[(size_t)s * KS + threadIdx.x];   // the tiles (one round trip, not two)
      for (int it = threadIdx.x; it < 128 * 4; it += blockDim.x) {   // (batch row, 4 consecutive features)
        const int brow = it >> 2, fg = it & 3;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < KS; ++j) v += *reinterpret_cast<const f32x4*>(tiles + (size_t)j * (128 * 16) + brow * 16 + fg * 4);
        if (p.relu) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        *reinterpret_cast<f32x4*>(fin_lg + brow * 16 + fg * 4) = v;
        if (brow < B) {
          float* yp = reinterpret_cast<float*>(p.y) + ((size_t)s * B + brow) * N + fg * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (fg * 4 + i < N) yp[i] = v[i];
        }
      }
      if (threadIdx.x < 64) {                                      // wave 0: one slice per lane, folded by DPP sums
        own0 = wave_sum(q4.x);
        own1 = wave_sum(q4.y);
        own2 = wave_sum(q4.z);
      }
      if (threadIdx.x == 0) {
        p.ws[1 + (size_t)s * ntiles + nt] = make_float4(own0, own1, own2, 0.f);
        fp->ks_ticket[s] = 0u;                                    // ready for the next launch
      }
    }
    __syncthreads();                               // logits tile complete in LDS
    BNN_STAMP(10);
    float a = 0.f, b = 0.f, nll = 0.f;
    // thread 0 supplies this layer's own partial sums directly (every FINAL block computes its
    // own sum log sigma, so nothing is read from another block of this launch)
    const int own_layer = fk.n_layers - 1;
#ifdef BNN_STAMPS
    fin_sample(fk, fp->c, s, T, fin_lg, 16, own_layer, own0, own1, own2, part, a, b, nll, p.dbg ? p.dbg + (size_t)blockIdx.x * 16 : nullptr);
#else
    fin_sample(fk, fp->c, s, T, fin_lg, 16, own_layer, own0, own1, own2, part, a, b, nll);
#endif
    BNN_STAMP(11);
    if (threadIdx.x == 0) {
      fin_store(fk, s, a, b, nll);
      if (fk.S == 1) {
        if (fp->sums) {
          float* so = fin_sums_slot(fk, fp->sums);
          so[0] = a; so[1] = b; so[2] = nll; so[3] = 1.f;
        }
        if (fk.sample_counter) *fk.sample_counter += fk.sample_counter_inc;
      } else if (fp->ticket) {
        // last-arriving sample block folds the per-sample scalars (sample order) and advances
        // the Philox sample counter: every block has read it by the time it takes a ticket.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t tk = __hip_atomic_fetch_add(fp->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == (uint32_t)fk.S - 1u) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (fp->sums) {
            double ta = 0, tb = 0, tn = 0;
            const float* pa = fk.local_reparam ? fk.kl : fk.log_prior;
            for (int i = 0; i < fk.S; ++i) {
              if (pa) ta += __hip_atomic_load(pa + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (!fk.local_reparam && fk.log_q) tb += __hip_atomic_load(fk.log_q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (fk.nll) tn += __hip_atomic_load(fk.nll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            float* so = fin_sums_slot(fk, fp->sums);
            so[0] = (float)ta; so[1] = (float)tb; so[2] = (float)tn; so[3] = (float)fk.S;
          }
          *fp->ticket = 0u;
          if (fk.sample_counter) *fk.sample_counter += fk.sample_counter_inc;
        }
      }
    }
  }
  return true;
}

template <int MATH, int XDT, int R, bool ALIGNED>
__global__ __launch_bounds__(768) void bbb_fwd_kernel(const BbbK p) {
  bbb_fwd_body<MATH, XDT, R, ALIGNED, false>(p, nullptr);
}

template <int MATH, int XDT>
__global__ __launch_bounds__(768) void bbb_fwd_final_kernel(const BbbK p, const FinPack fp) {
  bbb_fwd_body<MATH, XDT, 1, true, true>(p, &fp);
}

// K1e  the output layer + finalize of ONE evaluation and the FIRST layer of the next one in one launch: two independent
// pieces of work (the caller keeps their buffers apart).  The output layer is a handful of latency-bound blocks
// (five K-slice blocks and a hand-off at the MNIST shape, ~10 us); as a launch of its own it holds its stream's chain
// for that long, next to the next evaluation's first layer it costs the chain nothing.
// One step further (three-layer nets): the hidden layer of evaluation j+1 joins as a third independent piece
// (blocks [nf, nf + nm), bf16 x, the first layer's tile plan), so that in steady state an evaluation is ONE launch.
template <int XDT1, int R1>
__global__ __launch_bounds__(768) void bbb_fwd_final_next_kernel(const BbbK p3, const FinPack fp, const BbbK pm, const BbbK p1,
                                                                 int nf, int nm, int n1, int plain) {
  // the two plain layers' block ranges start on, and are padded to, a multiple of 8 blocks, so the XCD-aware work order
  // of the stand-alone launch can hold inside each (`plain` = 0).  Measured for this kernel it does not pay: the stage
  // alone 14.8 against 15.4 us, but four evaluators side by side 105.8 k against 106.8 k samples/s on the same box
  // ([out,in] weights are read along their rows: no line is shared between tiles) -- block order is the default here;
  // the LR stage, whose tiles share the lines of gathered [in,out] weights, keeps the XCD order.
  const int b = (int)blockIdx.x;
  const int ef = (nf + 7) & ~7, em = ef + ((nm + 7) & ~7);
  int item;
  if (b < ef) {
    if (b < nf) bbb_fwd_body<BNN_MATH_BF16, BNN_BF16, 1, true, true>(p3, &fp, b);
  } else if (b < em) {
    if (xcd_piece_item(b - ef, nm, item, plain)) bbb_fwd_body<BNN_MATH_BF16, BNN_BF16, R1, true, false>(pm, nullptr, item);
  } else {
    if (xcd_piece_item(b - em, n1, item, plain)) bbb_fwd_body<BNN_MATH_BF16, XDT1, R1, true, false>(p1, nullptr, item);
  }
}

// K1d  one-sample tail of an evaluation in ONE launch: the last hidden layer (K1a, any tile plan) and, run by
// whichever of its blocks finishes last, the output layer + finalize (K1c without K-slices).  Every block
// drains its stores, one lane releases at agent scope and takes a ticket; the last arriver acquires and
// carries on (no block ever waits for another: placement-independent, guide G16).  Saves a dependent launch
// and the K-slice hand-off of the separate last-layer kernel on the latency chain of a one-sample evaluation.
template <int MATH, int R2>
__global__ __launch_bounds__(768) void bbb_fwd_tail2_kernel(const BbbK p2, const BbbK p3, const FinPack fp,
                                                            uint32_t* ticket, uint32_t n_real_blocks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // word 0 doubles as the "I am last" flag between the two
  uint32_t& last_flag = *reinterpret_cast<uint32_t*>(lds);      // bodies (static LDS would eat into the 160 KiB the slabs need)
  const bool real = bbb_fwd_body<MATH, BNN_BF16, R2, true, false>(p2, nullptr);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores of the layer's outputs and statistics
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t last = 0u;
    if (real) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tk == n_real_blocks - 1u) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *ticket = 0u;                                    // ready for the next launch
        last = 1u;
      }
    }
    last_flag = last;
  }
  __syncthreads();
  if (last_flag == 0u) return;                           // block-uniform
  __syncthreads();
  bbb_fwd_body<MATH, BNN_BF16, 1, true, true>(p3, &fp, 0);
}

// matmul half over pre-sampled bf16 weights (bnn_bbb_sample_weights): no generator work in the launch
template <int XDT, int R, int MT>
__global__ __launch_bounds__(768) void bbb_fwd_pre_kernel(const BbbK p) {
  bbb_fwd_body<BNN_MATH_BF16, XDT, R, true, false, false, true, MT>(p, nullptr);
}

// input gradient over the bf16 weights the forward sampled (no generator work: the transposed generator draws a
// whole Philox group per weight it needs, four times the forward's cost)
template <int R, bool ALIGNED, int MT>
__global__ __launch_bounds__(768) void bbb_input_grad_pre_kernel(const BbbK p) {
  bbb_fwd_body<BNN_MATH_BF16, BNN_F32, R, ALIGNED, false, true, true, MT>(p, nullptr);
}

template <int MATH, int R, bool ALIGNED>
__global__ __launch_bounds__(768) void bbb_input_grad_kernel(const BbbK p) {
  bbb_fwd_body<MATH, BNN_F32, R, ALIGNED, false, true>(p, nullptr);
}

// ------------------------------------------------------------------------------------------
// Throughput kernel (many MC samples in flight, bf16 math, bf16 x): the block GEMM form.
//   1-D grid over (feature-tile group, sample, batch block) work items (XCD-aware order),
//   block = NW waves.  Wave j owns the 16 output features of tile tb*NW + j for ALL of K (no
//   cross-wave reduction, no LDS slabs); the 128 x 32 x-tile of each k-step is brought ONCE
//   per block into double-buffered LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs are held
//   across the generator work).  One DMA wave-instruction fills exactly one batch tile's
//   lane-linear 1 KiB fragment block: lane l fetches x[16m + (l&15)][k0 + 8(l>>4) .. +7], so
//   every wave's B-fragment read is one conflict-free ds_read_b128.  The DMA for tile t+1 is
//   issued at the top of step t and drained (vmcnt(0)) just before the step's single barrier:
//   a whole k-step of Philox/softplus work covers its latency.  ~115 VGPRs -> 4 waves/SIMD.
template <int NW, bool SIG>
__global__ __launch_bounds__(NW * 64, 4) void bbb_fwd_gemm_kernel(const BbbK p) {
  __shared__ __attribute__((aligned(16))) float4 xt[2][8 * 64];      // 2 x 8 KiB
  __shared__ float bias_s[NW][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N, B = p.B;
  const int tbs = (N + 16 * NW - 1) / (16 * NW), mbs = (B + 127) >> 7;
  int item;
  const int KS = p.ksl;                                       // K-range slices (deterministic split-K)
  if (!xcd_work_item(tbs * p.S * mbs * KS, item)) return;      // block-uniform
  const int ks = item % KS;
  item /= KS;
  const int tb = item / (p.S * mbs), s = (item / mbs) % p.S, mb = item % mbs;
  const int tile = tb * NW + wave;
  const int n = tile * 16 + r;
  const bool n_ok = n < N;
  const int nc = min(n, N - 1);
  const int m0 = mb * 128;
  const int ksteps = (K + 31) >> 5;
  const int spb = (ksteps + KS - 1) / KS, t_lo = ks * spb, t_hi = min(ksteps, t_lo + spb);
  const uint32_t gs = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u) + (uint32_t)s;
  const bool do_stats = p.want_stats && mb == 0;
  const bool do_ls = do_stats && s == 0;
  const bool do_dump = mb == 0;
  const int gpr = (K + 3) >> 2;
  const uint32_t wid = p.layer_id * 4u;
  const __bf16* xs = reinterpret_cast<const __bf16*>(p.x) + (size_t)s * (size_t)p.x_sstride;
  const int T = (N + 15) >> 4;

  if (do_stats && item == 0 && ks == 0 && threadIdx.x == 0) p.ws[0] = make_float4(__int_as_float(T * KS), 0.f, 0.f, 0.f);

  // LDS-DMA staging: wave w brings batch tiles m = w, w + NW, ... of the k-step's x tile.
  size_t xrow[8 / NW];
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i) xrow[i] = (size_t)min(m0 + (wave + i * NW) * 16 + r, B - 1) * K;
  auto stage_dma = [&](int t, int buf) {
    const int kk = min(t * 32 + q * 8, K - 8);
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xs + xrow[i] + kk),
                                       (__attribute__((address_space(3))) void*)&xt[buf][(wave + i * NW) * 64], 16, 0, 0);
    }
  };

  float mu_n[8], rho_n[8];
  auto load_params = [&](int t) {
    const size_t woff = (size_t)nc * K + min(t * 32 + q * 8, K - 8);
    load8<true>(p.w_mu + woff, 8, mu_n);
    load8<true>((SIG ? p.w_sigma : p.w_rho) + woff, 8, rho_n);
  };

  // bias of this wave's tile: eps now, applied in the epilogue
  float bmu_pre = 0.f, brho_pre = 0.f, beps_pre = 0.f;
  if (q == 0 && n_ok && ks == 0) {
    bmu_pre = p.b_mu[n];
    brho_pre = p.b_rho[n];
    beps_pre = bias_eps(p, n, s, gs, do_dump);
  }

  if (t_lo < t_hi) {
    load_params(t_lo);
    stage_dma(t_lo, t_lo & 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s_e2 = 0.f, s_a = 0.f, s_ls = 0.f;

#pragma nounroll
  for (int t = t_lo; t < t_hi; ++t) {
    const int k = t * 32 + q * 8;
    const bool lane_ok = n_ok && k < K;
    float mu[8], sg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = mu_n[j];
      sg[j] = rho_n[j];
    }
    const bool more = t + 1 < t_hi;
    if (more) {
      stage_dma(t + 1, (t + 1) & 1);     // buffer (t+1)&1 was last read in step t-1 (barrier since)
      load_params(t + 1);
    }
    float e[8], w[8];
    if (p.eps_mode == BNN_EPS_PHILOX) {
      const uint32_t g = (uint32_t)n * (uint32_t)gpr + (uint32_t)(k >> 2);
      philox_normal4(g, gs, wid, p.k0, p.k1, e);
      philox_normal4(g + 1u, gs, wid, p.k0, p.k1, e + 4);
    } else if (p.eps_mode == BNN_EPS_MEMORY) {
      load8<true>(p.eps_w + ((size_t)s * N + n) * K + k, lane_ok ? 8 : 0, e);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = 0.f;
    }
    if (p.eps_w_dump && do_dump) store8<true>(p.eps_w_dump + ((size_t)s * N + n) * K + k, lane_ok ? 8 : 0, e);
    float e2 = 0.f, a = 0.f, ls = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (!SIG) sg[j] = softplus(sg[j]);               // SIG: sigma arrives precomputed
      w[j] = __builtin_fmaf(sg[j], e[j], mu[j]);
      e2 = __builtin_fmaf(e[j], e[j], e2);
    }
    if (do_stats) {
      if (p.prior_kind == BNN_PRIOR_GAUSS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a = __builtin_fmaf(w[j], w[j], a);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) a = add_log(a, mix_p(p, w[j]));
      }
      if (do_ls) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ls = add_log(ls, sg[j]);
      }
      s_e2 += lane_ok ? e2 : 0.f;
      s_a += lane_ok ? a : 0.f;
      s_ls += lane_ok ? ls : 0.f;
    }
    bf16x8 wa;
#pragma unroll
    for (int j = 0; j < 8; ++j) wa[j] = lane_ok ? (__bf16)w[j] : (__bf16)0.f;
    const float4* xb = xt[t & 1];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const bf16x8 xf = __builtin_bit_cast(bf16x8, xb[(m * 4 + q) * 16 + r]);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xf, acc[m], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
    __syncthreads();
  }

  // ---- epilogue: bias, stats, store (no cross-wave reduction)
  if (q == 0) {
    float b = 0.f;
    if (n_ok && ks == 0) b = sample_bias(p, bmu_pre, brho_pre, beps_pre, do_stats, do_ls, s_e2, s_a, s_ls);
    bias_s[wave][r] = b;
  }
  if (do_stats) {
    const float a = wave_sum(s_e2), b = wave_sum(s_a), cc = wave_sum(s_ls);
    if (lane == 0 && tile < T) p.ws[1 + ((size_t)s * T + tile) * KS + ks] = make_float4(a, b, cc, 0.f);
  }
  __syncthreads();
  const int nb = tile * 16 + q * 4;
  const bool vec_ok = (N & 3) == 0;
  if (nb < N) {
    float bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = bias_s[wave][q * 4 + i];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int brow = m0 + m * 16 + r;
      if (brow < B && KS > 1) {                          // raw fp32 partial of this K slice
        f32x4 v = acc[m];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += bq[i];
        float* pp = p.ks_part + (((size_t)ks * p.S + s) * B + brow) * N + nb;
        if (vec_ok) {
          *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < N) pp[i] = v[i];
        }
      } else if (brow < B) {
        f32x4 v = acc[m];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float o = v[i] + bq[i];
          if (p.relu) o = fmaxf(o, 0.f);
          v[i] = o;
        }
        const size_t yoff = ((size_t)s * B + brow) * N + nb;
        if (p.y_bf16) {
          __bf16* yp = reinterpret_cast<__bf16*>(p.y) + yoff;
          if (vec_ok) {
            bf16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
            *reinterpret_cast<bf16x4*>(yp) = o;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (nb + i < N) yp[i] = (__bf16)v[i];
          }
        } else {
          float* yp = reinterpret_cast<float*>(p.y) + yoff;
          if (vec_ok) {
            *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (nb + i < N) yp[i] = v[i];
          }
        }
      }
    }
  }
}

// Sum of the K-slice partials of the GEMM form (slice order), ReLU, down-conversion.
__global__ void ks_reduce_kernel(const float* __restrict__ part, int KS, long cnt, int relu, void* __restrict__ y,
                                 int y_bf16, int vec_ok) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
  if (vec_ok) {
    for (long i = tid; i < (cnt >> 2); i += nt) {
      float4 v = reinterpret_cast<const float4*>(part)[i];
      for (int j = 1; j < KS; ++j) {
        const float4 u = reinterpret_cast<const float4*>(part + (size_t)j * cnt)[i];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (y_bf16) {
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        reinterpret_cast<bf16x4*>(y)[i] = o;
      } else {
        reinterpret_cast<float4*>(y)[i] = v;
      }
    }
  } else {
    for (long i = tid; i < cnt; i += nt) {
      float v = part[i];
      for (int j = 1; j < KS; ++j) v += part[(size_t)j * cnt + i];
      if (relu) v = fmaxf(v, 0.f);
      if (y_bf16) reinterpret_cast<__bf16*>(y)[i] = (__bf16)v;
      else reinterpret_cast<float*>(y)[i] = v;
    }
  }
}

// Per-layer reduction of the stats partials into the scalars BayesianLinear stores
// (networks.py:82-83).  One block per sample.
__global__ void bbb_layer_scalars_kernel(const float4* __restrict__ ws, int K, int N, bnn_prior prior,
                                         float* __restrict__ log_prior, float* __restrict__ log_q) {
  __shared__ double scratch[16];
  const int s = blockIdx.x;
  const int T = __float_as_int(ws[0].x);
  double e2 = 0, a = 0, ls = 0;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float4 v = ws[1 + (size_t)s * T + t];
    const float4 v0 = ws[1 + t];
    e2 += v.x;
    a += v.y;
    ls += v0.z;
  }
  e2 = block_sum(e2, scratch);
  a = block_sum(a, scratch);
  ls = block_sum(ls, scratch);
  if (threadIdx.x == 0) {
    const double cnt = (double)N * K + N;
    const double c0 = -0.91893853320467274178;
    if (log_q) log_q[s] = (float)(cnt * c0 - ls - 0.5 * e2);
    if (log_prior) {
      if (prior.kind == BNN_PRIOR_GAUSS)
        log_prior[s] = (float)(cnt * (c0 - log((double)prior.sigma_p)) -
                               a / (2.0 * (double)prior.sigma_p * (double)prior.sigma_p));
      else
        log_prior[s] = (float)a;
    }
  }
}

}  // namespace bnn

using namespace bnn;

extern "C" size_t bnn_bbb_linear_fwd_workspace_bytes(int32_t n_samples, int32_t out_features) {
  if (n_samples <= 0 || out_features <= 0) return 0;
  return (1 + (size_t)n_samples * (size_t)((out_features + 3) / 4)) * 4 * sizeof(float);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

namespace {
struct Plan {
  int R, nw, tiles;
};

// Launch geometry.  Depends only on the problem shape (and optional env overrides), never on
// pointers: the same shape always runs the same summation order.
Plan make_plan(int S, int B, int K, int N, bool aligned, int concurrency = 1) {
  Plan pl{};
  const int mbs = (B + 127) / 128;
  const int forceR = env_int("BNN_HIP_BBB_R", 0);
  const int forceNw = env_int("BNN_HIP_BBB_WAVES", 0);
  int R = 1;
  if (aligned) {
    // narrower tiles (more k-range classes per MFMA tile) until the launch covers the chip
    // ... or its share of the chip when `concurrency` launches like this one run side by side: with four
    // one-sample evaluations in flight 75 tiles of 16 features each (300 blocks in all) beat 150 of 8 (measured
    // 14.7 -> 12.5 us per evaluation) although one launch alone is slower that way (10.4 -> 13.5 us)
    const long want = 120 / (concurrency > 1 ? concurrency : 1);
    while (R < 4 && (long)((N + 16 / R - 1) / (16 / R)) * S * mbs < want) R *= 2;
    if (forceR == 1 || forceR == 2 || forceR == 4) R = forceR;
  }
  const int F = 16 / R;
  const int ssteps = (K + 32 * R - 1) / (32 * R);
  int spw = 1;                                       // super-steps per wave; waves divide them evenly
  while ((ssteps + spw - 1) / spw > 12) ++spw;
  int nw = (ssteps + spw - 1) / spw;
  // Several hundred blocks in flight: smaller blocks (4 waves, 32 KiB of slabs) so that three of
  // them share a CU and cover each other's load and barrier stalls; a lone sample instead wants
  // the shortest per-wave chain (one or two k-steps per wave).
  if ((long)((N + F - 1) / F) * S * mbs >= 400 && ssteps >= 8) nw = 4;
  // a CU has 4 SIMDs: 10 waves sit 3,3,2,2 and the block waits for the crowded pair (stamps: 3.9 k cycles
  // of barrier wait); a multiple of 4 keeps the generator work per SIMD even (measured 11.2 -> 10.4 us per
  // one-sample launch of the 1200x1200 layer, 16.2 -> 14.8 us per evaluation with four in flight)
  else if (ssteps >= 4) nw = (ssteps >= 12 ? 12 : ssteps) & ~3;   // as many waves as there are k-steps, up to 12, in fours
  if (forceNw > 0) nw = forceNw > 12 ? 12 : forceNw;
  if (nw > ssteps) nw = ssteps;
  pl.R = R;
  pl.nw = nw < 1 ? 1 : nw;
  pl.tiles = (N + F - 1) / F;
  return pl;
}

template <typename KernelT>
hipError_t allow_big_lds(KernelT kernel, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             160 * 1024);
}
}  // namespace

// Validate the arguments and fill the kernel parameter block.  `al` = the 16-byte vector path
// applies (K % 8 == 0, aligned bases).
static int prepare(const bnn_bbb_fwd_args* a, BbbK& k, bool& al) {
  k.mask = nullptr; k.mask_sstride = 0;      // input-gradient form only
  k.w_pre = nullptr; k.b_pre = nullptr;      // set below for the matmul-only form
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_bbb_fwd_args)) return BNN_ERR_ABI;
  if (a->n_samples <= 0 || a->batch <= 0 || a->in_features <= 0 || a->out_features <= 0) return BNN_ERR_SHAPE;
  if ((double)a->n_samples * ((a->batch + 127) / 128) * ((a->out_features + 3) / 4) > 2.0e9) return BNN_ERR_SHAPE;
  const bool pre = a->w_sampled != nullptr;
  if (!a->x || !a->y) return BNN_ERR_NULL;
  if (!pre && (!a->w_mu || !a->w_rho || !a->b_mu || !a->b_rho)) return BNN_ERR_NULL;
  if (pre) {                                       // matmul half over bnn_bbb_sample_weights' output
    if (!a->b_sampled) return BNN_ERR_NULL;
    if (a->math != BNN_MATH_BF16 || a->want_stats || a->log_prior || a->log_q || a->eps_w_dump || a->eps_b_dump ||
        (a->in_features & 7))
      return BNN_ERR_SHAPE;
    if (!aligned16(a->x) || !aligned16(a->w_sampled)) return BNN_ERR_ALIGN;
  }
  if ((unsigned)a->x_dtype > 1u || (unsigned)a->y_dtype > 1u || (unsigned)a->math > 1u || (unsigned)a->eps_mode > 2u ||
      (unsigned)a->prior.kind > 1u)
    return BNN_ERR_ENUM;
  if (!pre && a->eps_mode == BNN_EPS_MEMORY && (!a->eps_w || !a->eps_b)) return BNN_ERR_NULL;
  if (a->want_stats) {
    if (!a->workspace || a->workspace_bytes < bnn_bbb_linear_fwd_workspace_bytes(a->n_samples, a->out_features))
      return BNN_ERR_WORKSPACE;
    if (!aligned16(a->workspace)) return BNN_ERR_ALIGN;
  }
  if ((a->log_prior || a->log_q) && !a->want_stats) return BNN_ERR_WORKSPACE;
  k.x = a->x;
  k.x_sstride = a->x_per_sample ? (long)a->batch * a->in_features : 0;
  k.w_mu = a->w_mu; k.w_rho = a->w_rho; k.b_mu = a->b_mu; k.b_rho = a->b_rho;
  k.w_sigma = a->w_sigma;
  if (a->w_sigma && (reinterpret_cast<uintptr_t>(a->w_sigma) & 15)) return BNN_ERR_ALIGN;
  k.eps_w = a->eps_w; k.eps_b = a->eps_b; k.eps_w_dump = a->eps_w_dump; k.eps_b_dump = a->eps_b_dump;
  k.y = a->y;
  k.ws = a->want_stats ? reinterpret_cast<float4*>(a->workspace) : nullptr;
  k.S = a->n_samples; k.B = a->batch; k.K = a->in_features; k.N = a->out_features;
  k.eps_mode = a->eps_mode; k.prior_kind = a->prior.kind; k.want_stats = a->want_stats ? 1 : 0;
  k.relu = a->relu ? 1 : 0; k.y_bf16 = a->y_dtype == BNN_BF16; k.spb = 1; k.ksl = 1; k.ks_part = nullptr; k.ldw = 0;
  k.k0 = (uint32_t)a->seed; k.k1 = (uint32_t)(a->seed >> 32);
  k.layer_id = a->layer_id; k.sample_offset = a->sample_offset; k.sample_counter = a->sample_counter;
#ifdef BNN_STAMPS
  {
    const char* v = getenv("BNN_HIP_DBG_PTR");
    k.dbg = v ? reinterpret_cast<unsigned long long*>(strtoull(v, nullptr, 0)) : nullptr;
  }
#endif
  const double c0 = -0.91893853320467274178;
  k.pi = a->prior.pi;
  if (a->prior.kind == BNN_PRIOR_MIXTURE) {
    if (!(a->prior.sigma1 > 0.f) || !(a->prior.sigma2 > 0.f)) return BNN_ERR_SHAPE;
    k.inv2var1 = (float)(1.0 / (2.0 * (double)a->prior.sigma1 * a->prior.sigma1));
    k.inv2var2 = (float)(1.0 / (2.0 * (double)a->prior.sigma2 * a->prior.sigma2));
    k.c1 = (float)(c0 - log((double)a->prior.sigma1));
    k.c2 = (float)(c0 - log((double)a->prior.sigma2));
  } else {
    if (a->want_stats && !(a->prior.sigma_p > 0.f)) return BNN_ERR_SHAPE;
    k.inv2var1 = k.inv2var2 = k.c1 = k.c2 = 0.f;
  }
  const int K = a->in_features;
  al = (K % 8 == 0) && aligned16(a->x) && aligned16(a->w_mu) && aligned16(a->w_rho);
  if (pre) {
    k.w_pre = reinterpret_cast<const __bf16*>(a->w_sampled);
    k.b_pre = a->b_sampled;
    k.eps_mode = BNN_EPS_ZERO;
    al = true;
  } else if (a->eps_mode == BNN_EPS_MEMORY) al = al && aligned16(a->eps_w);
  if (a->eps_w_dump) al = al && aligned16(a->eps_w_dump);
  const bool ybf = a->y_dtype == BNN_BF16;
  if ((a->out_features % 4 == 0) && (reinterpret_cast<uintptr_t>(a->y) & (ybf ? 7 : 15))) return BNN_ERR_ALIGN;
  return BNN_OK;
}

extern "C" int bnn_bbb_linear_fwd(const bnn_bbb_fwd_args* a, void* stream_) {
  BbbK k;
  bool al = false;
  const int rc = prepare(a, k, al);
  if (rc != BNN_OK) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int K = a->in_features;
  const int xdt = a->x_dtype, math = a->math;
  hipError_t err = hipSuccess;
  const int mbs = (a->batch + 127) / 128;
  // Throughput form (block GEMM, x tile shared through LDS) once the launch has enough
  // (feature-tile-group x sample) blocks to fill the chip several waves deep.
  const long gemm_blocks = (long)((a->out_features + 63) / 64) * a->n_samples * mbs;
  const int force = env_int("BNN_HIP_BBB_GEMM", -1);
  // K-range slices for the GEMM form when a modest number of samples would leave the chip
  // under-filled: slices write fp32 partial tiles into the caller's split scratch and a tiny
  // reduce kernel sums them in slice order (deterministic), applies ReLU and the down-conversion.
  int ksl = 1;
  const int ksteps_g = (K + 31) / 32;
  const size_t part_bytes = (size_t)a->n_samples * a->batch * a->out_features * sizeof(float);
  if (a->split_scratch && gemm_blocks < 450) {
    ksl = (int)((600 + gemm_blocks - 1) / gemm_blocks);
    if (ksl > 8) ksl = 8;
    if (ksl > ksteps_g / 4) ksl = ksteps_g / 4;
    // the stats workspace has ceil(N/4) entries per sample; the sliced form uses T*ksl of them
    const int max_ks = ((a->out_features + 3) / 4) / ((a->out_features + 15) / 16);
    if (ksl > max_ks) ksl = max_ks;
    const int fk = env_int("BNN_HIP_BBB_GEMM_KS", 0);
    if (fk >= 1 && fk <= max_ks) ksl = fk;
    if (ksl < 1) ksl = 1;
    if (a->split_scratch_bytes < (size_t)ksl * part_bytes || (reinterpret_cast<uintptr_t>(a->split_scratch) & 15)) ksl = 1;
  }
  if (a->w_sampled) {
    // ---- matmul half only: same tile machinery, no generator work.  Two rebuilds of it measured slower and were
    // dropped: every load of a wave issued in one round (10 waves x 2 steps of 64 k, 136 landing registers: 11.8
    // against 8.7 us at 128 x 1200 x 1200), and a dedicated kernel whose blocks own 2 feature tiles x 4 batch tiles so
    // that each fragment feeds more MFMAs (half the L2 -> L1 bytes: 6.1 against 5.2 us per two layers with four
    // evaluations side by side).  What did help is below: 32-row batch blocks.
    // Batch tiles per block: 2 (32 rows).  Without generator work a block's cost is the x it pulls through its CU's L1
    // -- all of K for its rows -- so four times the blocks each ingest a quarter of it (BNN_HIP_PRE_MT=8: 128 rows).
    const int mt = env_int("BNN_HIP_PRE_MT", 2) == 8 ? 8 : 2;
    const int mbs_p = (a->batch + 16 * mt - 1) / (16 * mt);
    Plan pl = make_plan(a->n_samples, a->batch, K, a->out_features, true, a->concurrency);
    if (mt == 2 || pl.R > 2) {                           // enough blocks already: whole 16-feature tiles
      pl.R = 1;
      pl.tiles = (a->out_features + 15) / 16;
    }
    const int fr = env_int("BNN_HIP_PRE_R", 0), fw = env_int("BNN_HIP_PRE_WAVES", 0);
    if (fr == 1 || fr == 2) {
      pl.R = fr;
      pl.tiles = (a->out_features + 16 / fr - 1) / (16 / fr);
    }
    if (fw >= 1 && fw <= 12) pl.nw = fw;
    const long total = (long)pl.tiles * a->n_samples * mbs_p;
    const dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(pl.nw * 64);
    const size_t lds = ((size_t)pl.nw * mt * 64 * 4 + 16 + 3 * pl.nw) * sizeof(float);
#define BNN_PRE(XDT, RR, MTT)                                                                     \
  do {                                                                                            \
    err = allow_big_lds(bbb_fwd_pre_kernel<XDT, RR, MTT>, lds);                                   \
    if (err == hipSuccess)                                                                        \
      hipLaunchKernelGGL((bbb_fwd_pre_kernel<XDT, RR, MTT>), grid, block, lds, stream, k);        \
  } while (0)
#define BNN_PRE_R(XDT)                                               \
  do {                                                               \
    if (mt == 2) { if (pl.R == 1) BNN_PRE(XDT, 1, 2); else BNN_PRE(XDT, 2, 2); }  \
    else { if (pl.R == 1) BNN_PRE(XDT, 1, 8); else BNN_PRE(XDT, 2, 8); }          \
  } while (0)
    if (xdt == BNN_F32) BNN_PRE_R(BNN_F32); else BNN_PRE_R(BNN_BF16);
#undef BNN_PRE
#undef BNN_PRE_R
    if (err != hipSuccess) return (int)err;
    err = hipGetLastError();
    return err == hipSuccess ? BNN_OK : (int)err;
  }
  const bool gemm_ok = al && math == BNN_MATH_BF16 && xdt == BNN_BF16 && K >= 8;
  const bool use_gemm = gemm_ok && (force == 1 || (force != 0 && (gemm_blocks >= 450 || (ksl > 1 && gemm_blocks * ksl >= 300))));
  if (use_gemm) {
    if (gemm_blocks >= 450 && !env_int("BNN_HIP_BBB_GEMM_KS", 0)) ksl = 1;
    k.ksl = ksl;
    k.ks_part = reinterpret_cast<float*>(a->split_scratch);
    const long blocks = gemm_blocks * ksl;
    const dim3 grid((unsigned)(((blocks + 7) / 8) * 8)), block(256);
    if (a->w_sigma)
      hipLaunchKernelGGL((bbb_fwd_gemm_kernel<4, true>), grid, block, 0, stream, k);
    else
      hipLaunchKernelGGL((bbb_fwd_gemm_kernel<4, false>), grid, block, 0, stream, k);
    if (ksl > 1) {
      err = hipGetLastError();
      if (err != hipSuccess) return (int)err;
      const long cnt = (long)a->n_samples * a->batch * a->out_features;
      const int vec_ok = (cnt % 4 == 0) && !(reinterpret_cast<uintptr_t>(a->y) & 15);
      long nb = (cnt / 4 + 255) / 256;
      nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
      hipLaunchKernelGGL(ks_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, k.ks_part, ksl, cnt, k.relu, a->y,
                         k.y_bf16, vec_ok);
    }
  } else {
    const Plan pl = make_plan(a->n_samples, a->batch, K, a->out_features, al, a->concurrency);
    const long total = (long)pl.tiles * a->n_samples * mbs;
    const dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(pl.nw * 64);
    const size_t lds = ((size_t)pl.nw * 8 * 64 * 4 + 16 + 3 * pl.nw) * sizeof(float);
#define BNN_GO(MATH, XDT, RR, AL)                                                              \
  do {                                                                                         \
    err = allow_big_lds(bbb_fwd_kernel<MATH, XDT, RR, AL>, lds);                               \
    if (err == hipSuccess)                                                                     \
      hipLaunchKernelGGL((bbb_fwd_kernel<MATH, XDT, RR, AL>), grid, block, lds, stream, k);    \
  } while (0)
#define BNN_GO_R(MATH, XDT)                                   \
  do {                                                        \
    if (!al) BNN_GO(MATH, XDT, 1, false);                     \
    else if (pl.R == 1) BNN_GO(MATH, XDT, 1, true);           \
    else if (pl.R == 2) BNN_GO(MATH, XDT, 2, true);           \
    else BNN_GO(MATH, XDT, 4, true);                          \
  } while (0)
    if (math == BNN_MATH_BF16) {
      if (xdt == BNN_F32) BNN_GO_R(BNN_MATH_BF16, BNN_F32); else BNN_GO_R(BNN_MATH_BF16, BNN_BF16);
    } else {
      if (xdt == BNN_F32) BNN_GO_R(BNN_MATH_F32, BNN_F32); else BNN_GO_R(BNN_MATH_F32, BNN_BF16);
    }
#undef BNN_GO
#undef BNN_GO_R
    if (err != hipSuccess) return (int)err;
  }
  err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  if (a->log_prior || a->log_q) {
    hipLaunchKernelGGL(bbb_layer_scalars_kernel, dim3(a->n_samples), dim3(256), 0, stream, k.ws, K, a->out_features,
                       a->prior, a->log_prior, a->log_q);
    err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  return BNN_OK;
}

static constexpr int kFinalMaxSlices = 8;

// reduce.hip: sums over the per-sample outputs + sample-counter advance (one block)
extern "C" int bnn_elbo_sums_(const bnn_finalize_args* f, void* stream);

// Scratch of the fused last layer: per-sample tickets, per-slice stats and partial logits tiles.
// Zero-initialised ONCE by the caller (the kernel leaves the tickets at zero again).
extern "C" size_t bnn_bbb_final_scratch_bytes(int32_t n_samples) {
  if (n_samples <= 0) return 0;
  const size_t S = (size_t)n_samples;
  return ((S * 4 + 255) / 256) * 256 + S * kFinalMaxSlices * 16 + S * kFinalMaxSlices * (128 * 16 * 4);
}

// Last layer + ELBO finalize.  Fused into ONE launch when the layer is a single 16-feature
// tile over a single 128-row batch block (MNIST: 10 classes; regression: 1 output); otherwise
// the two launches of bnn_bbb_linear_fwd + bnn_elbo_finalize.
// a plain layer that may ride in a combined launch: K1a tile form, bf16 math, vector path, nothing on the side
static bool stage_ok(const bnn_bbb_fwd_args* l, bool al) {
  const int mbs = (l->batch + 127) / 128;
  const long gemm_blocks = (long)((l->out_features + 63) / 64) * l->n_samples * mbs;
  return al && l->math == BNN_MATH_BF16 && !l->w_sampled && !l->split_scratch && !l->log_prior && !l->log_q &&
         gemm_blocks < 450 && env_int("BNN_HIP_BBB_GEMM", -1) != 1;
}

#define BNN_STAGE_LAUNCH(NEXT, PLR, GRID, BLOCK, LDS, ...)                                                     \
  do {                                                                                                         \
    if ((NEXT)->x_dtype == BNN_F32) {                                                                          \
      if ((PLR) == 1) {                                                                                        \
        err = allow_big_lds(bbb_fwd_final_next_kernel<BNN_F32, 1>, LDS);                                       \
        if (err == hipSuccess) hipLaunchKernelGGL((bbb_fwd_final_next_kernel<BNN_F32, 1>), GRID, BLOCK, LDS, stream, __VA_ARGS__); \
      } else {                                                                                                 \
        err = allow_big_lds(bbb_fwd_final_next_kernel<BNN_F32, 2>, LDS);                                       \
        if (err == hipSuccess) hipLaunchKernelGGL((bbb_fwd_final_next_kernel<BNN_F32, 2>), GRID, BLOCK, LDS, stream, __VA_ARGS__); \
      }                                                                                                        \
    } else {                                                                                                   \
      if ((PLR) == 1) {                                                                                        \
        err = allow_big_lds(bbb_fwd_final_next_kernel<BNN_BF16, 1>, LDS);                                      \
        if (err == hipSuccess) hipLaunchKernelGGL((bbb_fwd_final_next_kernel<BNN_BF16, 1>), GRID, BLOCK, LDS, stream, __VA_ARGS__); \
      } else {                                                                                                 \
        err = allow_big_lds(bbb_fwd_final_next_kernel<BNN_BF16, 2>, LDS);                                      \
        if (err == hipSuccess) hipLaunchKernelGGL((bbb_fwd_final_next_kernel<BNN_BF16, 2>), GRID, BLOCK, LDS, stream, __VA_ARGS__); \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)

// two independent plain layers (the hidden layer of one evaluation, the first layer of the next) in one launch
static int stage_pair(const bnn_bbb_fwd_args* mid, const bnn_bbb_fwd_args* next, void* stream_) {
  BbbK km, k1;
  bool alm = false, al1 = false;
  int rc = prepare(mid, km, alm);
  if (rc != BNN_OK) return rc;
  rc = prepare(next, k1, al1);
  if (rc != BNN_OK) return rc;
  const Plan plm = make_plan(mid->n_samples, mid->batch, mid->in_features, mid->out_features, true, mid->concurrency);
  const Plan pl1 = make_plan(next->n_samples, next->batch, next->in_features, next->out_features, true, next->concurrency);
  if (!(stage_ok(mid, alm) && stage_ok(next, al1) && mid->x_dtype == BNN_BF16 && plm.R == pl1.R && pl1.R <= 2 &&
        env_int("BNN_HIP_FINAL_NEXT", 1) != 0)) {
    rc = bnn_bbb_linear_fwd(mid, stream_);
    return rc != BNN_OK ? rc : bnn_bbb_linear_fwd(next, stream_);
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int nwc = plm.nw > pl1.nw ? plm.nw : pl1.nw;
  const long totm = (long)plm.tiles * mid->n_samples * ((mid->batch + 127) / 128);
  const long tot1 = (long)pl1.tiles * next->n_samples * ((next->batch + 127) / 128);
  const dim3 grid((unsigned)(((totm + 7) & ~7L) + ((tot1 + 7) & ~7L))), block(nwc * 64);
  const size_t lds = ((size_t)nwc * 8 * 64 * 4 + 16 + 3 * nwc) * sizeof(float);
  FinPack fp;
  memset(&fp, 0, sizeof(fp));
  hipError_t err = hipSuccess;
  BNN_STAGE_LAUNCH(next, pl1.R, grid, block, lds, km, fp, km, k1, 0, (int)totm, (int)tot1, env_int("BNN_HIP_STAGE_PLAIN_ORDER", 1));
  if (err != hipSuccess) return (int)err;
  err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

static int final_fwd_impl(const bnn_bbb_fwd_args* a, const bnn_finalize_args* f, const bnn_bbb_fwd_args* mid,
                          const bnn_bbb_fwd_args* next, void* stream_) {
  BbbK k;
  bool al = false;
  int rc = prepare(a, k, al);
  if (rc != BNN_OK) return rc;
  FinPack fp;
  rc = make_fin(f, fp.k, fp.c);
  if (rc != BNN_OK) return rc;
  const int nl = f->n_layers;
  const bool fuse = al && a->out_features <= 16 && a->batch <= 128 && a->want_stats && !f->local_reparam && nl >= 1 &&
                    f->layer_workspace[nl - 1] == a->workspace && f->n_samples == a->n_samples &&
                    f->classes == a->out_features && f->batch == a->batch && a->y_dtype == BNN_F32 &&
                    (a->n_samples == 1 || a->n_samples > 16 || f->ticket != nullptr) && !(a->log_prior || a->log_q) &&
                    env_int("BNN_HIP_FUSE_FINAL", 1) != 0;
  if (!fuse) {
    rc = bnn_bbb_linear_fwd(a, stream_);
    if (rc != BNN_OK) return rc;
    rc = bnn_elbo_finalize(f, stream_);
    if (rc != BNN_OK) return rc;
    if (mid && next) return stage_pair(mid, next, stream_);
    if (mid) return bnn_bbb_linear_fwd(mid, stream_);
    return next ? bnn_bbb_linear_fwd(next, stream_) : BNN_OK;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  fp.sums = f->sums;
  // Few samples: the last-arriving sample block folds the sums and advances the counter (one
  // release/acquire per block).  Many samples: those fences (an L2 write-back each) cost more
  // than a launch, so a one-block follow-up kernel does it instead.
  const bool tail_kernel = a->n_samples > 16;
  fp.ticket = tail_kernel ? nullptr : f->ticket;
  const int K = a->in_features;
  const int ssteps = (K + 31) / 32;
  // K-range slices per sample: one k-step per wave where the scratch allows it (<= 12 waves per
  // block), so the last layer's latency chain is one step long instead of ceil(ssteps/12).
  int KS = (ssteps + 11) / 12;
  KS = KS > kFinalMaxSlices ? kFinalMaxSlices : KS;
  for (int ks = KS; ks <= kFinalMaxSlices; ++ks) {     // ... and a wave count in fours (even work per SIMD), if a
    const int per = (ssteps + ks - 1) / ks;             // slightly finer slicing gives one: 38 steps -> 5 slices of 8
    if (per <= 12 && ((per & 3) == 0 || per < 4)) {
      KS = ks;
      break;
    }
  }
  if ((long)a->n_samples * KS > 128) KS = 1;           // enough sample blocks already: skip the hand-off
  const int forceKs = env_int("BNN_HIP_FINAL_KS", 0);
  if (forceKs >= 1 && forceKs <= kFinalMaxSlices) KS = forceKs;
  if (KS > 1 && (!f->scratch || f->scratch_bytes < bnn_bbb_final_scratch_bytes(a->n_samples) ||
                 (reinterpret_cast<uintptr_t>(f->scratch) & 15)))
    KS = 1;
  const int spb = (ssteps + KS - 1) / KS;
  int spw = 1;
  while ((spb + spw - 1) / spw > 12) ++spw;
  int nw = (spb + spw - 1) / spw;
  nw = nw < 1 ? 1 : nw;
  fp.ks = KS;
  {
    char* base = reinterpret_cast<char*>(f->scratch);
    const size_t S = (size_t)a->n_samples;
    fp.ks_ticket = reinterpret_cast<uint32_t*>(base);
    fp.ks_stats = reinterpret_cast<float4*>(base + ((S * 4 + 255) / 256) * 256);
    fp.ks_tiles = reinterpret_cast<float*>(base + ((S * 4 + 255) / 256) * 256 + S * kFinalMaxSlices * 16);
  }
  const long total = (long)a->n_samples * KS;           // one tile, one batch block, KS slices
  hipError_t err = hipSuccess;
  if (!next && mid) {                                  // one plain layer beside the output layer: same kernel, it rides as `next`
    next = mid;
    mid = nullptr;
  }
  if (next) {
    // ---- combined launch with the next evaluation's first layer (and the hidden layer of the one between), when
    // all take the plain tile forms
    BbbK k1, km;
    bool al1 = false, alm = false;
    rc = prepare(next, k1, al1);
    if (rc != BNN_OK) return rc;
    if (mid) {
      rc = prepare(mid, km, alm);
      if (rc != BNN_OK) return rc;
    }
    const bool combine = !tail_kernel && a->math == BNN_MATH_BF16 && a->x_dtype == BNN_BF16 && stage_ok(next, al1) &&
                         env_int("BNN_HIP_FINAL_NEXT", 1) != 0;
    if (combine) {
      const Plan pl1 = make_plan(next->n_samples, next->batch, next->in_features, next->out_features, true, next->concurrency);
      Plan plm = pl1;
      bool mid_in = false;
      if (mid) {
        plm = make_plan(mid->n_samples, mid->batch, mid->in_features, mid->out_features, true, mid->concurrency);
        mid_in = stage_ok(mid, alm) && mid->x_dtype == BNN_BF16 && plm.R == pl1.R;
      }
      if (pl1.R <= 2) {
        int nwc = pl1.nw > nw ? pl1.nw : nw;
        if (mid_in && plm.nw > nwc) nwc = plm.nw;
        const long total1 = (long)pl1.tiles * next->n_samples * ((next->batch + 127) / 128);
        const long totalm = mid_in ? (long)plm.tiles * mid->n_samples * ((mid->batch + 127) / 128) : 0;
        const dim3 gridc((unsigned)(((total + 7) & ~7L) + ((totalm + 7) & ~7L) + ((total1 + 7) & ~7L))), blockc(nwc * 64);
        const size_t ldsc = ((size_t)nwc * 8 * 64 * 4 + 16 + 3 * nwc + 128 * 16 + kFinMaxWaves * kFinNV) * sizeof(float);
        if (!mid_in) km = k1;
        BNN_STAGE_LAUNCH(next, pl1.R, gridc, blockc, ldsc, k, fp, km, k1, (int)total, (int)totalm, (int)total1,
                         env_int("BNN_HIP_STAGE_PLAIN_ORDER", 1));
        if (err != hipSuccess) return (int)err;
        err = hipGetLastError();
        if (err != hipSuccess) return (int)err;
        return (mid && !mid_in) ? bnn_bbb_linear_fwd(mid, stream_) : BNN_OK;
      }
    }
  }
  const dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(nw * 64);
  const size_t lds = ((size_t)nw * 8 * 64 * 4 + 16 + 3 * nw + 128 * 16 + kFinMaxWaves * kFinNV) * sizeof(float);
#define BNN_FIN(MATH, XDT)                                                                     \
  do {                                                                                         \
    err = allow_big_lds(bbb_fwd_final_kernel<MATH, XDT>, lds);                                 \
    if (err == hipSuccess)                                                                     \
      hipLaunchKernelGGL((bbb_fwd_final_kernel<MATH, XDT>), grid, block, lds, stream, k, fp);  \
  } while (0)
  if (a->math == BNN_MATH_BF16) {
    if (a->x_dtype == BNN_F32) BNN_FIN(BNN_MATH_BF16, BNN_F32); else BNN_FIN(BNN_MATH_BF16, BNN_BF16);
  } else {
    if (a->x_dtype == BNN_F32) BNN_FIN(BNN_MATH_F32, BNN_F32); else BNN_FIN(BNN_MATH_F32, BNN_BF16);
  }
#undef BNN_FIN
  if (err != hipSuccess) return (int)err;
  err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  if (tail_kernel) {
    rc = bnn_elbo_sums_(f, stream_);
    if (rc != BNN_OK) return rc;
  }
  if (mid && next) return stage_pair(mid, next, stream_);
  if (mid) return bnn_bbb_linear_fwd(mid, stream_);
  return next ? bnn_bbb_linear_fwd(next, stream_) : BNN_OK;
}

extern "C" int bnn_bbb_final_fwd(const bnn_bbb_fwd_args* a, const bnn_finalize_args* f, void* stream_) {
  return final_fwd_impl(a, f, nullptr, nullptr, stream_);
}

// One stage of a software pipeline over independent evaluations: any of {output layer + finalize of evaluation j,
// hidden layer of evaluation j+1, first layer of evaluation j+2} in ONE launch (see include/bnn_hip.h).
extern "C" int bnn_bbb_stage_fwd(const bnn_bbb_fwd_args* last, const bnn_finalize_args* fin, const bnn_bbb_fwd_args* mid,
                                 const bnn_bbb_fwd_args* first, void* stream_) {
  if ((last == nullptr) != (fin == nullptr)) return BNN_ERR_NULL;
  if (last) return final_fwd_impl(last, fin, mid, first, stream_);
  if (mid && first) return stage_pair(mid, first, stream_);
  if (mid) return bnn_bbb_linear_fwd(mid, stream_);
  if (first) return bnn_bbb_linear_fwd(first, stream_);
  return BNN_ERR_NULL;
}

// bnn_bbb_final_fwd(last, fin) and bnn_bbb_linear_fwd(next_first) -- the first layer of the NEXT, independent
// evaluation -- in one launch when both take their plain tile forms; the two calls otherwise.
extern "C" int bnn_bbb_final_next_fwd(const bnn_bbb_fwd_args* a, const bnn_finalize_args* f, const bnn_bbb_fwd_args* next,
                                      void* stream_) {
  if (!next) return BNN_ERR_NULL;
  return final_fwd_impl(a, f, nullptr, next, stream_);
}

// The last hidden layer + the output layer + finalize of a ONE-sample evaluation in one launch (K1d) when the
// shapes allow it and BNN_HIP_FUSE_TAIL2=1; otherwise bnn_bbb_linear_fwd(hidden) followed by
// bnn_bbb_final_fwd(last, fin).  OFF by default: measured on MI355X the single block that inherits the output
// layer (38 k-steps over its 12 waves, all of the hidden activations through one CU's L1) takes longer than the
// dependent launch of five K-slice blocks it replaces: 47.0 against 40.3 us per evaluation alone, 13.8 against
// 11.7 us with four in flight.
extern "C" int bnn_bbb_tail2_fwd(const bnn_bbb_fwd_args* hidden, const bnn_bbb_fwd_args* last, const bnn_finalize_args* f,
                                 void* stream_) {
  BbbK k2, k3;
  bool al2 = false, al3 = false;
  int rc = prepare(hidden, k2, al2);
  if (rc != BNN_OK) return rc;
  rc = prepare(last, k3, al3);
  if (rc != BNN_OK) return rc;
  FinPack fp;
  rc = make_fin(f, fp.k, fp.c);
  if (rc != BNN_OK) return rc;
  const int nl = f->n_layers;
  const long gemm_blocks = (long)((hidden->out_features + 63) / 64) * hidden->n_samples * ((hidden->batch + 127) / 128);
  const bool fuse =
      env_int("BNN_HIP_FUSE_TAIL2", 0) != 0 && f->ticket != nullptr && hidden->n_samples == 1 && last->n_samples == 1 &&
      f->n_samples == 1 && nl >= 2 && al2 && al3 && hidden->math == BNN_MATH_BF16 && last->math == BNN_MATH_BF16 &&
      hidden->x_dtype == BNN_BF16 && hidden->y_dtype == BNN_BF16 && last->x_dtype == BNN_BF16 && last->x == hidden->y &&
      last->x_per_sample == 1 && last->in_features == hidden->out_features && last->batch == hidden->batch &&
      hidden->batch <= 128 && last->out_features <= 16 && hidden->want_stats && last->want_stats && !f->local_reparam &&
      f->layer_workspace[nl - 1] == last->workspace && f->layer_workspace[nl - 2] == hidden->workspace &&
      f->classes == last->out_features && f->batch == last->batch && last->y_dtype == BNN_F32 &&
      !(hidden->log_prior || hidden->log_q || last->log_prior || last->log_q) && !hidden->split_scratch &&
      gemm_blocks < 450 && env_int("BNN_HIP_BBB_GEMM", -1) != 1;
  if (!fuse) {
    rc = bnn_bbb_linear_fwd(hidden, stream_);
    if (rc != BNN_OK) return rc;
    return bnn_bbb_final_fwd(last, f, stream_);
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  fp.sums = f->sums;
  fp.ticket = nullptr;                                  // one sample: the finalising block writes the sums itself
  fp.ks = 1;
  fp.ks_ticket = nullptr;
  fp.ks_stats = nullptr;
  fp.ks_tiles = nullptr;
  const Plan pl = make_plan(1, hidden->batch, hidden->in_features, hidden->out_features, true, hidden->concurrency);
  const long total = (long)pl.tiles;                    // one sample, one batch block
  const dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(pl.nw * 64);
  const size_t lds = ((size_t)pl.nw * 8 * 64 * 4 + 16 + 3 * pl.nw + 128 * 16 + kFinMaxWaves * kFinNV + 2) * sizeof(float);
  hipError_t err = hipSuccess;
#define BNN_T2(RR)                                                                                                   \
  do {                                                                                                               \
    err = allow_big_lds(bbb_fwd_tail2_kernel<BNN_MATH_BF16, RR>, lds);                                               \
    if (err == hipSuccess)                                                                                           \
      hipLaunchKernelGGL((bbb_fwd_tail2_kernel<BNN_MATH_BF16, RR>), grid, block, lds, stream, k2, k3, fp, f->ticket, \
                         (uint32_t)total);                                                                           \
  } while (0)
  if (pl.R == 1) BNN_T2(1);
  else if (pl.R == 2) BNN_T2(2);
  else BNN_T2(4);
#undef BNN_T2
  if (err != hipSuccess) return (int)err;
  err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

// gx[S,B,K] = gz[S,B,N] . w_s with w regenerated (TRANS form of the K-split kernel).  Called by
// bnn_bbb_linear_bwd (bbb_bwd.hip).
extern "C" int bnn_bbb_input_grad_(const bnn_bbb_bwd_args* a, const float* gz, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  BbbK k{};
  k.x = gz;
  k.x_sstride = (long)a->batch * a->out_features;
  k.w_mu = a->w_mu; k.w_rho = a->w_rho; k.b_mu = a->b_mu; k.b_rho = a->b_rho;
  k.w_sigma = nullptr;
  k.eps_w = a->eps_w; k.eps_b = a->eps_b; k.eps_w_dump = nullptr; k.eps_b_dump = nullptr;
  k.y = a->g_x;
  k.ws = nullptr;
  k.S = a->n_samples; k.B = a->batch;
  k.K = a->out_features;            // reduction length
  k.N = a->in_features;             // produced features
  k.ldw = a->in_features;
  k.mask = a->gx_relu_mask ? a->x : nullptr;
  k.mask_sstride = a->x_per_sample ? (long)a->batch * a->in_features : 0;
  k.w_pre = reinterpret_cast<const __bf16*>(a->w_sampled);
  k.b_pre = nullptr;
  k.eps_mode = a->eps_mode; k.prior_kind = a->prior.kind; k.want_stats = 0; k.relu = 0; k.y_bf16 = 0; k.spb = 1;
  k.k0 = (uint32_t)a->seed; k.k1 = (uint32_t)(a->seed >> 32);
  k.layer_id = a->layer_id; k.sample_offset = a->sample_offset; k.sample_counter = a->sample_counter;
  k.inv2var1 = k.inv2var2 = k.c1 = k.c2 = k.pi = 0.f;
#ifdef BNN_STAMPS
  k.dbg = nullptr;
#endif
  if ((a->in_features % 4 == 0) && (reinterpret_cast<uintptr_t>(a->g_x) & 15)) return BNN_ERR_ALIGN;
  const bool al = (k.K % 8 == 0) && aligned16(gz);
  const Plan pl = make_plan(k.S, k.B, k.K, k.N, al);
  const long total = (long)pl.tiles * k.S * ((k.B + 127) / 128);
  const dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(pl.nw * 64);
  const size_t lds = ((size_t)pl.nw * 8 * 64 * 4 + 16 + 3 * pl.nw) * sizeof(float);
  hipError_t err = hipSuccess;
#define BNN_IG(MATH, RR, AL)                                                                    \
  do {                                                                                          \
    err = allow_big_lds(bbb_input_grad_kernel<MATH, RR, AL>, lds);                              \
    if (err == hipSuccess)                                                                      \
      hipLaunchKernelGGL((bbb_input_grad_kernel<MATH, RR, AL>), grid, block, lds, stream, k);   \
  } while (0)
#define BNN_IG_R(MATH)                                 \
  do {                                                 \
    if (!al) BNN_IG(MATH, 1, false);                   \
    else if (pl.R == 1) BNN_IG(MATH, 1, true);         \
    else if (pl.R == 2) BNN_IG(MATH, 2, true);         \
    else BNN_IG(MATH, 4, true);                        \
  } while (0)
#define BNN_IGP(RR, AL, MTT)                                                                         \
  do {                                                                                               \
    err = allow_big_lds(bbb_input_grad_pre_kernel<RR, AL, MTT>, ldsp);                               \
    if (err == hipSuccess)                                                                           \
      hipLaunchKernelGGL((bbb_input_grad_pre_kernel<RR, AL, MTT>), gridp, block, ldsp, stream, k);   \
  } while (0)
  if (a->w_sampled) {
    if (a->math != BNN_MATH_BF16) return BNN_ERR_ENUM;   // the sampled weights are the bf16 operands of that mode
    // 128-row batch blocks here: a lane's 8 reduction-consecutive weights are 8 strided 2-byte loads in this
    // direction, and 32-row blocks would repeat them four times (28.8 against 21.4 us at 2 x 128 x 1200 x 1200)
    const int mt = env_int("BNN_HIP_PRE_MT_BWD", 8) == 2 ? 2 : 8;
    const long totalp = (long)((k.N + 15) / 16) * k.S * ((k.B + 16 * mt - 1) / (16 * mt));
    const dim3 gridp((unsigned)(((totalp + 7) / 8) * 8));
    const size_t ldsp = ((size_t)pl.nw * mt * 64 * 4 + 16 + 3 * pl.nw) * sizeof(float);
    if (mt == 2) { if (!al) BNN_IGP(1, false, 2); else BNN_IGP(1, true, 2); }
    else { if (!al) BNN_IGP(1, false, 8); else BNN_IGP(1, true, 8); }
  } else if (a->math == BNN_MATH_BF16) BNN_IG_R(BNN_MATH_BF16); else BNN_IG_R(BNN_MATH_F32);
#undef BNN_IGP
#undef BNN_IG
#undef BNN_IG_R
  if (err != hipSuccess) return (int)err;
  err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}
