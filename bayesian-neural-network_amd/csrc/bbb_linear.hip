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
#include "bbb_sample_body.h"
#include "bbb_block_gemm.h"
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
  __bf16* y16;      // optional bf16 copy of an fp32 y (tile forms)
  int eps_mode, prior_kind, want_stats, relu, y_bf16, spb;
  int ksl;          // GEMM form: K-range slices per (tile group, sample, batch block) unit; 1 = none
  float4* ks_part;  // GEMM form, ksl > 1: fp32 partial tiles [unit][ksl][wave][batch tile][lane] (bias in slice 0)
  uint32_t* ks_ticket;  // GEMM form, ksl > 1: [unit] arrival counters of a unit's slice blocks, zero between launches
#ifdef BNN_TUNE
  int tune;             // tuning build only
#endif
  Xcd2D xc;             // GEMM form: work order (feature group x (sample, batch block) x K slice), see bnn_device.h
  int ldw;          // TRANS only: leading dimension of the [out,in] weight matrix (= original in_features)
  const __bf16* w_pre;  // PRE only: sampled weights bf16 [S, N, K] (bnn_bbb_sample_weights); no sampling in the launch
  const float* b_pre;   // PRE only: sampled biases [S, N] (nullptr: none -- the input-gradient use of the matmul)
  __bf16* wt_out;       // PRE only, optional: the sampled weights written once more TRANSPOSED, [S, K, N] (bnn_bbb_fwd_args.w_sampled_t_out)
  const float* mask;  // TRANS only, optional: [S|1, B, N] the layer's input; the stored gx is multiplied by (mask > 0)
  long mask_sstride;
  uint32_t k0, k1, layer_id, sample_offset;
  const uint32_t* sample_counter;
  int xg;                                 // samples sharing one x row block (x index = s / xg); x_sstride = 0: one x for all
  uint32_t sgrp, sgrp_stride;             // sample groups (bnn_bbb_fwd_args.sample_group): 0 = none
  float inv2var1, c1, inv2var2, c2, pi;   // mixture: log N(w;0,s_i) = c_i - w^2 * inv2var_i
  const void* x_lo;                       // BNN_MATH_BF16X3, bf16 x: the low plane of x (strided like x)
  void* y_lo;                             // BNN_MATH_BF16X3, bf16 y: the low plane of y
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

// global MC sample index (Philox subsequence) of local sample s
__device__ __forceinline__ uint32_t global_sample(const BbbK& p, int s) {
  const uint32_t base = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u);
  if (p.sgrp == 0u) return base + (uint32_t)s;
  const uint32_t g = (uint32_t)s / p.sgrp;
  return base + g * p.sgrp_stride + ((uint32_t)s - g * p.sgrp);
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
      if (p.y_lo) {                                // split-bf16 math: the low plane, bf16(v - hi)
        __bf16* lp = reinterpret_cast<__bf16*>(p.y_lo) + yoff;
        bf16x4 l;
#pragma unroll
        for (int i = 0; i < 4; ++i) l[i] = (__bf16)(v[i] - (float)(__bf16)v[i]);
        if (vec_ok) {
          *reinterpret_cast<bf16x4*>(lp) = l;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < N) lp[i] = l[i];
        }
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
      if (p.y16) {                                 // fp32 y for the backward, bf16 y for the next layer's forward
        __bf16* cp = p.y16 + yoff;
        if (vec_ok) {
          bf16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
          *reinterpret_cast<bf16x4*>(cp) = o;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < N) cp[i] = (__bf16)v[i];
        }
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
// Returns false for the padding blocks of the XCD-aware grid (no work done).
// MT: 16-row batch tiles per block (8 = 128 rows).  The matmul-only forms use 2: without generator work a block's
// cost is the x it pulls through its CU's L1 (all of K for its rows), so four times the blocks each ingest a quarter;
// with sampling fused every batch block would redo the tile's sampling, hence 8 there.
template <int MATH, int XDT, int R, bool ALIGNED, bool FINAL, bool TRANS = false, bool PRE = false, int MT = 8, bool WT = false>
__device__ __forceinline__ bool bbb_fwd_body(const BbbK& p, const FinPack* fp) {
  constexpr int F = 16 / R;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int c = r / F, f = r % F;
  const int K = p.K, N = p.N, B = p.B;
  const int ntiles = (N + F - 1) / F, mbs = (B + 16 * MT - 1) / (16 * MT);
  int item;
  const int KS = FINAL ? fp->ks : 1;                          // K-range slices per sample (FINAL only)
  if (!xcd_work_item(ntiles * p.S * mbs * KS, item)) return false;   // block-uniform
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
  const uint32_t gs = global_sample(p, s);
  const bool do_stats = p.want_stats && mb == 0;
  const bool do_ls = do_stats && (s == 0 || FINAL);
  const bool do_dump = mb == 0;
  const int gpr = (K + 3) >> 2;
  const uint32_t wid = p.layer_id * 4u;
  const char* xs = reinterpret_cast<const char*>(p.x) + (size_t)(s / p.xg) * (size_t)p.x_sstride * (XDT == BNN_F32 ? 4 : 2);
  // split-bf16 math (BNN_MATH_BF16X3): bf16 activations are a pair of planes, fp32 activations are split in registers
  constexpr bool X3 = MATH == BNN_MATH_BF16X3;
  static_assert(!X3 || (!TRANS && !PRE && !WT), "the split-bf16 mode is a forward sampling form");
  const char* xs_lo = (X3 && XDT == BNN_BF16) ? reinterpret_cast<const char*>(p.x_lo) + (size_t)(s / p.xg) * (size_t)p.x_sstride * 2 : nullptr;

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
    if (!TRANS && wave == nw - 1 && lane < F && n_ok && ks == 0 && p.b_pre) bmu_pre = p.b_pre[(size_t)s * N + n];
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
    constexpr int FR = (XDT == BNN_F32 || X3) ? 2 : 1;            // 16-byte loads per fragment (X3 on bf16 x: hi, lo)
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
            } else if (X3) {
              xraw[(mm * R + cc) * 2 + 0] = *reinterpret_cast<const float4*>(reinterpret_cast<const __bf16*>(xs) + off);
              xraw[(mm * R + cc) * 2 + 1] = *reinterpret_cast<const float4*>(reinterpret_cast<const __bf16*>(xs_lo) + off);
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
              if (X3) {
                bf16x8 vl;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                  vl[j] = (j < xvalid) ? (reinterpret_cast<const __bf16*>(xs_lo) + off)[j] : (__bf16)0.0f;
                xraw[(mm * R + cc) * 2 + 0] = __builtin_bit_cast(float4, vb);
                xraw[(mm * R + cc) * 2 + 1] = __builtin_bit_cast(float4, vl);
              } else {
                xraw[mm * R + cc] = __builtin_bit_cast(float4, vb);
              }
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
      philox_normal8(g, gs, wid, p.k0, p.k1, e);
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
    bf16x8 wa, wz, wl;
    if (X3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        wa[j] = (__bf16)w[j];
        wl[j] = split_lo(w[j], wa[j]);
        wz[j] = (__bf16)0.f;
      }
    }
    if (MATH == BNN_MATH_BF16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        wa[j] = (__bf16)w[j];
        wz[j] = (__bf16)0.f;
      }
      if (PRE && (TRANS || valid > 0)) wa = __builtin_bit_cast(bf16x8, wraw);   // padding (k >= K, n >= N) is zero
      if (WT && valid > 0 && n_ok) {
        // the row blocks of a (tile, sample) leave the tile's weights transposed between them (block mb takes the k rows
        // j = mb mod mbs of every lane's eight): 2-byte stores, 32 contiguous bytes per k row and 16-lane group -- the
        // input gradient of a training step then runs as THIS matmul over [S, K, N] instead of gathering 2-byte elements
        // along the reduction (all eight rows on the first row block made it the launch's long pole: +3.6 us)
        __bf16* dst = p.wt_out + ((size_t)s * K + k) * N + n;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j % mbs == mb) dst[(size_t)j * N] = wa[j];
      }
    }
#pragma unroll
    for (int ch = 0; ch < MT / MC; ++ch) {
      if (ch * MC < mtiles) {                     // block-uniform
#pragma unroll
        for (int mm = 0; mm < MC; ++mm) {
#pragma unroll
          for (int cc = 0; cc < R; ++cc) {
            const int m = ch * MC + mm;
            if (X3) {
              bf16x8 xh, xl;
              if (XDT == BNN_F32) {
                const float4 lo = xraw[(mm * R + cc) * 2], hi = xraw[(mm * R + cc) * 2 + 1];
                const float xv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  xh[j] = (__bf16)xv[j];
                  xl[j] = split_lo(xv[j], xh[j]);
                }
              } else {
                xh = __builtin_bit_cast(bf16x8, xraw[(mm * R + cc) * 2]);
                xl = __builtin_bit_cast(bf16x8, xraw[(mm * R + cc) * 2 + 1]);
              }
              const bool mine = R == 1 || c == cc;
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine ? wa : wz, xh, acc[m], 0, 0, 0);
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine ? wl : wz, xh, acc[m], 0, 0, 0);
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine ? wa : wz, xl, acc[m], 0, 0, 0);
            } else if (MATH == BNN_MATH_BF16) {
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
          float* so = fp->sums;
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
          if (fp->sums) fin_fold_sums(fk, fp->sums);
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

// K1a carrying an independent sampling job (bnn_bbb_fwd_args.rider) as extra blocks behind its own (whose count the
// launcher pads to a multiple of 8, so the XCD-aware work order of the layer's blocks is unchanged).
template <int XDT, int R>
__global__ __launch_bounds__(768) void bbb_fwd_rider_kernel(const BbbK p, const SampleK sk, int n_main) {
  if ((int)blockIdx.x >= n_main) {                         // block-uniform
    extern __shared__ __attribute__((aligned(16))) float lds[];
    sample_block(sk, (int)blockIdx.x - n_main, lds);
    return;
  }
  bbb_fwd_body<BNN_MATH_BF16, XDT, R, true, false>(p, nullptr);
}

// K1r  the output layer of a few-sample evaluation over PRE-SAMPLED weights, split by batch rows, with the finalize.
// Grid: per sample RB = ceil(B / 16) row blocks + 1 statistics block, 256 threads each.
//   row block : logits of its 16 batch rows = bf16 x . w^T (4 waves split the k-steps, every load of a wave issued
//               before its MFMAs; one LDS round to add the waves up), + bias, stored; the rows' NLL (networks.py:183-190).
//   stats     : the layers' log p / log q partial sums (networks.py:174-178), as K4 forms them.
// Hand-off (placement-independent, nobody waits): each block's thread 0 stores its scalar(s) write-through (agent-scope
// relaxed atomic store = sc1), drains (vmcnt(0)) and takes a ticket; the block whose ticket is last reads the RB + 2
// scalars back with agent-scope loads, in a fixed order, and writes the sample's outputs.  Samples meet the same way.
struct FinRows {
  const void* x;        // [S | shared, B, K], bf16 or (XF32: the training step's saved activations) fp32 rounded here
  long x_sstride;
  int xg;
  const __bf16* w;      // [S, N, K]
  const float* b;       // [S, N]
  float* y;             // [S, B, N]
  int S, B, K, N, relu;
  uint32_t* tickets;    // [S], zero between launches
  float* parts;         // [S][16]: 0..7 the row blocks' NLL, 8 / 9 log p (KL) / log q
};

template <bool XF32>
__global__ __launch_bounds__(256) void bbb_final_rows_kernel(const FinRows p, const FinPack fp, const FinLoss tr) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4][64];
  __shared__ float lg[16][17];
  __shared__ __attribute__((aligned(8))) float part[4 * kFinNV];
  const FinK& fk = fp.k;
  const int RB = (p.B + 15) >> 4;
  const int s = (int)blockIdx.x / (RB + 1), rb = (int)blockIdx.x - s * (RB + 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pub0 = 0.f, pub1 = 0.f;
  int slot = rb;                                            // where thread 0 publishes
  if (rb == RB) {
    // ---- statistics block: every layer's partial sums of sample s (no logits: nll stays 0)
    int T[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) T[l] = (l < fk.n_layers) ? __float_as_int(reinterpret_cast<const float4*>(fk.ws[l])[0].x) : 0;
    float a = 0.f, b = 0.f, nll = 0.f;
    fin_sample(fk, fp.c, s, T, nullptr, 0, -1, 0.f, 0.f, 0.f, part, a, b, nll);
    pub0 = a;
    pub1 = b;
    slot = 8;
  } else {
    const int r = lane & 15, q = lane >> 4;
    const int K = p.K, N = p.N, B = p.B;
    const int row = min(rb * 16 + r, B - 1);
    const size_t xoff = (size_t)(s / p.xg) * (size_t)p.x_sstride + (size_t)row * K;
    const __bf16* xr = reinterpret_cast<const __bf16*>(p.x) + xoff;
    const float* xr32 = reinterpret_cast<const float*>(p.x) + xoff;
    const __bf16* wr = p.w + ((size_t)s * N + min(r, N - 1)) * K;
    const int ksteps = (K + 31) >> 5;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 10;                                   // k-steps in flight per wave: 38 steps of the 1200-wide layer = one round
    for (int base = wave; base < ksteps; base += 4 * U) {
      float4 xa[U], xh[XF32 ? U : 1], wa[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = (base + 4 * u) * 32 + q * 8;
        const int kk = min(k, K - 8);
        if (XF32) {
          xa[u] = *reinterpret_cast<const float4*>(xr32 + kk);
          xh[u] = *reinterpret_cast<const float4*>(xr32 + kk + 4);
        } else {
          xa[u] = *reinterpret_cast<const float4*>(xr + kk);
        }
        wa[u] = *reinterpret_cast<const float4*>(wr + kk);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = (base + 4 * u) * 32 + q * 8;
        const bool ok = (base + 4 * u) < ksteps && k < K && r < N;    // past the reduction / a padding feature: zero weights
        const float4 wz = ok ? wa[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        bf16x8 xb;
        if (XF32) {
          const float xv[8] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w, xh[u].x, xh[u].y, xh[u].z, xh[u].w};
#pragma unroll
          for (int j = 0; j < 8; ++j) xb[j] = (__bf16)xv[j];
        } else {
          xb = __builtin_bit_cast(bf16x8, xa[u]);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wz), xb, acc, 0, 0, 0);
      }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
      // lane (r = batch row of the block, q): features 4q .. 4q+3
      f32x4 v = red[0][lane];
#pragma unroll
      for (int wv = 1; wv < 4; ++wv) v += red[wv][lane];
      const int brow = rb * 16 + r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = q * 4 + i;
        float o = v[i] + (n < N ? p.b[(size_t)s * N + n] : 0.f);
        if (p.relu) o = fmaxf(o, 0.f);
        v[i] = o;
        lg[r][n] = o;
      }
      if (brow < B) {
        float* yp = p.y + ((size_t)s * B + brow) * N + q * 4;
        if ((N & 3) == 0 && q * 4 < N) {
          *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (q * 4 + i < N) yp[i] = v[i];
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
      // ---- NLL of the block's rows: lane r < 16 takes row r (the arithmetic of fin_sample's thread-per-row forms)
      float acc_n = 0.f;
      const int brow = rb * 16 + lane;
      if (lane < 16 && brow < B && fk.nll) {
        const int C = fk.C;
        if (fk.nll_mode == BNN_NLL_CLASSIFICATION) {
          const long long* tgt = reinterpret_cast<const long long*>(fk.target) + (fk.group > 0 ? (s / fk.group) * fk.tgt_stride : 0);
          const long long tc = tgt[brow];
          float mx = -3.0e38f, se = 0.f;
          for (int c = 0; c < C; ++c) mx = fmaxf(mx, lg[lane][c]);
          for (int c = 0; c < C; ++c) se += __expf(lg[lane][c] - mx);
          const float picked = (tc >= 0 && tc < C) ? lg[lane][(int)tc] : __builtin_nanf("");
          acc_n = (mx + __logf(se)) - picked;
        } else {
          const float* tgt = reinterpret_cast<const float*>(fk.target) + (fk.group > 0 ? (s / fk.group) * fk.tgt_stride : 0);
          for (int c = 0; c < C; ++c) {
            const float d = tgt[(size_t)brow * C + c] - lg[lane][c];
            acc_n += (float)((double)(d * d) * fp.c.reg_inv2var + fp.c.reg_const);
          }
        }
        if (tr.out4) fin_loss_row_grad(fk, tr, s, p.B, brow, lg[lane]);
      }
      pub0 = wave_sum(acc_n);
    }
  }
  if (threadIdx.x != 0) return;
  float* mine = p.parts + (size_t)s * 16;
  __hip_atomic_store(mine + slot, pub0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (rb == RB) __hip_atomic_store(mine + 9, pub1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t tk = __hip_atomic_fetch_add(p.tickets + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tk != (uint32_t)RB) return;                           // RB + 1 blocks per sample
  // ---- last block of the sample: fold (row-block order), store the sample's scalars
  // (all ten loads in flight before the first is consumed -- one round trip, not RB + 1: a loop over `tn += load` waits per
  // element; slots past RB hold no block's value and are not added)
  float pv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pv[i] = __hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float a = __hip_atomic_load(mine + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float b = __hip_atomic_load(mine + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double tn = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < RB) tn += pv[i];
  const float nll = (float)tn;
  __hip_atomic_store(p.tickets + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
  if (fk.log_prior) __hip_atomic_store(fk.log_prior + s, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fk.log_q) __hip_atomic_store(fk.log_q + s, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fk.nll) __hip_atomic_store(fk.nll + s, nll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fk.S == 1) {
    if (fp.sums) {
      fp.sums[0] = a; fp.sums[1] = b; fp.sums[2] = nll; fp.sums[3] = 1.f;
    }
    if (tr.out4) fin_loss_assemble(fk, tr);
    if (fk.sample_counter) *fk.sample_counter += fk.sample_counter_inc;
    return;
  }
  // ---- samples meet: the sample whose ticket is last folds the 4-vector(s) and advances the Philox counter (no ticket: a
  // follow-up launch does both)
  if (!fp.ticket) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t t2 = __hip_atomic_fetch_add(fp.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t2 != (uint32_t)fk.S - 1u) return;
  if (fp.sums) fin_fold_sums(fk, fp.sums);
  if (tr.out4) fin_loss_assemble(fk, tr);
  __hip_atomic_store(fp.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fk.sample_counter) *fk.sample_counter += fk.sample_counter_inc;
}

// matmul half over pre-sampled bf16 weights (bnn_bbb_sample_weights): no generator work in the launch
template <int XDT, int R, int MT, bool WT = false>
__global__ __launch_bounds__(768) void bbb_fwd_pre_kernel(const BbbK p) {
  bbb_fwd_body<BNN_MATH_BF16, XDT, R, true, false, false, true, MT, WT>(p, nullptr);
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
// EPS: the epsilon source is a COMPILE-TIME choice.  As a run-time branch the three sources met in one block ahead of
// w = mu + sigma * eps, and the compiler's wait-count pass, seeing eps come from memory on one path, put s_waitcnt vmcnt(0)
// there on all of them: every wave drained the prefetch it had just issued BEFORE its generator work instead of behind it.
template <int NW, bool SIG, int EPS>
__device__ __forceinline__ void bbb_gemm_body(const BbbK& p, float4 (*xt)[8 * 64], float (*bias_s)[16]) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N, B = p.B;
  const int tbs = (N + 16 * NW - 1) / (16 * NW), mbs = (B + 127) >> 7;
  const int KS = p.ksl;                                       // K-range slices (deterministic split-K)
  int tb, unit, ks;
  if (!xcd2d_work_item(p.xc, tb, unit, ks)) return;            // block-uniform
  const int s = unit / mbs, mb = unit - s * mbs;
  const int item = tb * (p.S * mbs) + unit;                    // dense (group, sample, batch block) index
  const int tile = tb * NW + wave;
  const int n = tile * 16 + r;
  const bool n_ok = n < N;
  const int nc = min(n, N - 1);
  const int m0 = mb * 128;
  const int ksteps = (K + 31) >> 5;
  const int spb = (ksteps + KS - 1) / KS, t_lo = ks * spb, t_hi = min(ksteps, t_lo + spb);
  const uint32_t gs = global_sample(p, s);
  const bool do_stats = p.want_stats && mb == 0;
  const bool do_ls = do_stats && s == 0;
  const bool do_dump = mb == 0;
  const int gpr = (K + 3) >> 2;
  const uint32_t wid = p.layer_id * 4u;
  const __bf16* xs = reinterpret_cast<const __bf16*>(p.x) + (size_t)(s / p.xg) * (size_t)p.x_sstride;
  const int T = (N + 15) >> 4;

  if (do_stats && item == 0 && ks == 0 && threadIdx.x == 0) p.ws[0] = make_float4(__int_as_float(T * KS), 0.f, 0.f, 0.f);

  // LDS-DMA staging: wave w brings batch tiles m = w, w + NW, ... of the k-step's x tile.
  size_t xrow[8 / NW];
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i) xrow[i] = (size_t)min(m0 + (wave + i * NW) * 16 + r, B - 1) * K;
  // Addresses as a SCALAR base + a 32-bit lane offset (global_load[_lds] with an SGPR base): the lane offsets are
  // loop-invariant, a step costs one add and one min (the K tail inside a k-step re-reads the row's last 8) instead of the
  // 64-bit per-lane address arithmetic and v_readfirstlane the pointer forms compile to -- vector issue is what these
  // launches are made of (K1b2's stage_fast).  The plan takes these forms only where the tensors' byte spans fit 32 bits.
  const uint32_t voff_w = ((uint32_t)nc * (uint32_t)K + (uint32_t)(q * 8)) * 4u, voff_w_max = ((uint32_t)nc * (uint32_t)K + (uint32_t)(K - 8)) * 4u;
  uint32_t voff_x[8 / NW], voff_x_max[8 / NW];
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i) {
    voff_x[i] = ((uint32_t)xrow[i] + (uint32_t)(q * 8)) * 2u;
    voff_x_max[i] = ((uint32_t)xrow[i] + (uint32_t)(K - 8)) * 2u;
  }
  auto stage_dma = [&](int t, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      const uint32_t vo = min(voff_x[i] + (uint32_t)t * 64u, voff_x_max[i]);
      const uint32_t m0v = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&xt[buf][(wave + i * NW) * 64];
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(xs), "s"(m0v) : "memory", "m0");
    }
  };

  // (mu, rho | sigma) of a step: four 16-byte loads per lane, issued one step AHEAD into one of two explicit register sets
  // (A / B) that the steps consume alternately.  A single loop-carried set made the compiler place its copies -- and with
  // them s_waitcnt vmcnt(0) -- right behind the loads, ahead of the generator work the prefetch was meant to overlap.
  struct PSet { float4 m_lo, m_hi, g_lo, g_hi; };
  auto load_params = [&](int t, PSet& P) __attribute__((always_inline)) {
    const uint32_t woff = min(voff_w + (uint32_t)t * 128u, voff_w_max);     // (the K tail re-reads the row's last 8)
    const float4* pm = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.w_mu) + woff);
    const float4* pg = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(SIG ? p.w_sigma : p.w_rho) + woff);
    P.m_lo = pm[0]; P.m_hi = pm[1]; P.g_lo = pg[0]; P.g_hi = pg[1];
  };

  // bias of this wave's tile: eps now, applied in the epilogue
  float bmu_pre = 0.f, brho_pre = 0.f, beps_pre = 0.f;
  if (q == 0 && n_ok && ks == 0) {
    bmu_pre = p.b_mu[n];
    brho_pre = p.b_rho[n];
    beps_pre = bias_eps(p, n, s, gs, do_dump);
  }

  PSet PA, PB;
  PA.m_lo = PA.m_hi = PA.g_lo = PA.g_hi = PB.m_lo = PB.m_hi = PB.g_lo = PB.g_hi = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t_lo < t_hi) {
    load_params(t_lo, PA);
    stage_dma(t_lo, t_lo & 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s_e2 = 0.f, s_a = 0.f, s_ls = 0.f;

  // one k-step over the parameters in P (loaded a step ago); MORE: step t + 1 exists and its loads go to Pn -- a
  // compile-time fact of the call site: issued under a run-time branch, the loads made the compiler wait for them
  // (vmcnt(0)) where the step first reads P, behind the generator work they were meant to overlap
  auto step = [&](int t, const PSet& P, PSet& Pn, auto more_) __attribute__((always_inline)) {
    constexpr bool more = decltype(more_)::value;
    const int k = t * 32 + q * 8;
    const bool lane_ok = n_ok && k < K;
#ifdef BNN_TUNE
    // tuning build only (wrong results; tools/k1b_ablate.py): BNN_TUNE_K1B bit 2 = no parameter loads in the loop, bit 3 = no x
    // DMA, bit 4 = no LDS reads, bit 5 = no MFMAs
    if (more) {
      if (!(p.tune & 8)) stage_dma(t + 1, (t + 1) & 1);
      if (!(p.tune & 4)) load_params(t + 1, Pn);
    }
#else
    if (more) {
      stage_dma(t + 1, (t + 1) & 1);     // buffer (t+1)&1 was last read in step t-1 (barrier since)
      load_params(t + 1, Pn);
    }
#endif
    const float mu[8] = {P.m_lo.x, P.m_lo.y, P.m_lo.z, P.m_lo.w, P.m_hi.x, P.m_hi.y, P.m_hi.z, P.m_hi.w};
    float sg[8] = {P.g_lo.x, P.g_lo.y, P.g_lo.z, P.g_lo.w, P.g_hi.x, P.g_hi.y, P.g_hi.z, P.g_hi.w};
    float e[8], w[8];
    if (EPS == BNN_EPS_PHILOX) {
      const uint32_t g = (uint32_t)n * (uint32_t)gpr + (uint32_t)(k >> 2);
      philox_normal8(g, gs, wid, p.k0, p.k1, e);
    } else if (EPS == BNN_EPS_MEMORY) {
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
    // x fragments by hand-written ds_read_b128: a compiler-visible read of `xt` is ordered behind EVERY LDS-DMA in flight
    // that may alias it (s_waitcnt vmcnt(0) ahead of the first read: the prefetch just issued), although buffer t & 1 was
    // complete at the last barrier.  The "+v" operands of the wait tie the MFMAs to it.
    const uint32_t xa = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&xt[t & 1][q * 16 + r];
#ifdef BNN_TUNE
    const bool tune_nolds = (p.tune & 16) != 0, tune_nomfma = (p.tune & 32) != 0;
#else
    constexpr bool tune_nolds = false, tune_nomfma = false;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 x0, x1, x2, x3;
      if (tune_nolds) {
        x0 = x1 = x2 = x3 = f32x4{mu[0], mu[1], mu[2], mu[3]};
      } else {
        if (h == 0)
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                       : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(xa));
        else
          asm volatile("ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:5120\n\tds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168"
                       : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(xa));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      }
      if (!tune_nomfma) {
        acc[h * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x0), acc[h * 4 + 0], 0, 0, 0);
        acc[h * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x1), acc[h * 4 + 1], 0, 0, 0);
        acc[h * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x2), acc[h * 4 + 2], 0, 0, 0);
        acc[h * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x3), acc[h * 4 + 3], 0, 0, 0);
      } else {
        asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3));
      }
    }
#ifdef BNN_TUNE
    if (!(p.tune & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(p.tune & 1)) __syncthreads();
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces (and the next step's parameters) have landed
    __syncthreads();
#endif
  };
  {
    int t = t_lo;
#pragma nounroll
    for (; t + 2 < t_hi; t += 2) {
      step(t, PA, PB, std::true_type{});
      step(t + 1, PB, PA, std::true_type{});
    }
    if (t + 1 < t_hi) {
      step(t, PA, PB, std::true_type{});
      step(t + 1, PB, PA, std::false_type{});
    } else if (t < t_hi) {
      step(t, PA, PB, std::false_type{});
    }
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
  float bq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bq[i] = bias_s[wave][q * 4 + i];
  if (KS > 1) {
    // ---- the K-range slices of one (tile group, sample, batch block) unit meet here, nobody waits: every wave
    // stores its eight fp32 partial tiles write-through (sc1; 1 KiB lane-linear per tile, whole 128-B lines per
    // store instruction) and drains them, one lane takes the unit's ticket behind the block barrier, and the block
    // whose ticket is last adds the slices up IN SLICE ORDER (its own from registers, the others by sc1 loads:
    // served by L2 / memory, never by this CU's L1), so the result does not depend on who arrived last.
    __shared__ uint32_t last_s;
    float4* const unit_part = p.ks_part + (size_t)item * KS * (NW * 8 * 64);
    float4* const mine = unit_part + ((size_t)ks * NW + wave) * (8 * 64) + lane;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      f32x4 v = acc[m];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += bq[i];           // bias_s is zero outside slice 0
      acc[m] = v;
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine + m * 64), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tk = __hip_atomic_fetch_add(p.ks_ticket + item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_s = (tk == (uint32_t)KS - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (last_s == 0u) return;                                // block-uniform
    if (threadIdx.x == 0) __hip_atomic_store(p.ks_ticket + item, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x4 sum[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) sum[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma nounroll
    for (int j = 0; j < KS; ++j) {                           // slice order; one round of 8 loads per lane per slice
      f32x4 part[8];
      if (j != ks) {                                         // block-uniform
        const float4* src = unit_part + ((size_t)j * NW + wave) * (8 * 64) + lane;
#pragma unroll
        for (int m = 0; m < 8; ++m)
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[m]) : "v"(src + m * 64) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) part[m] = acc[m];
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) sum[m] += part[m];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = sum[m];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = 0.f;                 // already in the partials
  }
  if (nb < N) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int brow = m0 + m * 16 + r;
      if (brow < B) {
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

template <int NW, bool SIG, int EPS>
__global__ __launch_bounds__(NW * 64, 4) void bbb_fwd_gemm_kernel(const BbbK p) {
  __shared__ __attribute__((aligned(16))) float4 xt[2][8 * 64];      // 2 x 8 KiB
  __shared__ float bias_s[NW][16];
  bbb_gemm_body<NW, SIG, EPS>(p, xt, bias_s);
}

// ------------------------------------------------------------------------------------------
// K1b2 -- the block GEMM form with the PARAMETERS shared through LDS as well (hoisted sigma only).
// With its prefetch repaired K1b is bound by the bytes its CUs pull from L2 as much as by its vector work
// (tools/k1b_ablate.py, 1200 x 1200, 256 pairs: 383 us with on-chip eps and statistics, 372 with eps = 0; without the
// (mu, sigma) register loads 225; without the x DMA 259; with neither 174): every (64-feature group, pair) block re-reads
// its group's 16 KiB of (mu, sigma) per k-step from L2 although all pairs of the launch use the same parameters.  Here a
// block is NF feature waves x SB PAIRS (a pair = one (minibatch | MC sample, 128-row batch block) unit): the NF x 4 KiB
// of (mu, sigma) of a k-step are brought ONCE per block by LDS-DMA and read by the SB waves that need them, each pair's
// 8 KiB x tile by LDS-DMA as before.  Per pair and k-step the block pulls 16 / SB + 8 KiB through its CU instead of 24,
// and a wave holds no parameter registers across the generator work.
//   LDS per buffer: [NF tiles][mu lo | mu hi | sigma lo | sigma hi][64 lanes] x 16 B, then [SB pairs][8 batch tiles]
//   [64 lanes] x 16 B; every piece is one lane-linear 1 KiB DMA wave-instruction whose lane l fetches exactly the 16
//   bytes lane l of the consuming wave reads back with one ds_read_b128 (conflict-free by construction).
// Everything else -- epsilon map, statistics, epilogue -- is K1b's; the per-step statistics are summed two lanes-worth at a
// time (packed fp32 FMAs), so they agree with K1b's to fp32 summation order, the outputs bit for bit.
// X3 (BNN_MATH_BF16X3): split-bf16 operands.  x arrives as two planes (hi, lo): a pair's tile of a k-step is 16 KiB
// ([hi | lo][8 batch tiles][64 lanes] x 16 B), the sampled weight fragment is split in registers (wa = bf16(w), wl =
// bf16(w - wa)), and every (batch tile, k-step) takes three MFMAs: wa . xh + wl . xh + wa . xl.  The staging buffers then
// would hold 48 KiB each -- 96 KiB per block, one 8-wave block per CU, two waves per SIMD: measured 636 us per 256-pair
// launch of the 1200 x 1200 layer against the bf16 form's 322 (both forms' steps take ~4000 cycles per WAVE: what a launch
// delivers is waves per SIMD over that latency).  So the X3 form keeps its PARAMETERS IN ONE BUFFER (PS): a step reads its
// (mu, sigma) fragments into registers first, the block meets (a second, cheap barrier right behind the step's start), and
// only then are the next step's parameter pieces requested into the same 16 KiB; x stays double-buffered, the bias goes
// through lane shuffles instead of LDS: 16 + 2 x 32 KiB = 80 KiB, two blocks per CU, four waves per SIMD like the bf16 form.
#ifndef BNN_K1B2_PS
#define BNN_K1B2_PS 0       // build knob: the bf16 form with single-buffered parameters too (48 KiB: three blocks per CU)
#endif
#ifndef BNN_K1B2_WPS
#define BNN_K1B2_WPS 4      // build knob: waves per SIMD the register allocation aims at
#endif
#ifndef BNN_K1B2_STAG
#define BNN_K1B2_STAG 0     // the second pair's waves one MFMA phase behind the first pair's (see STAG in the kernel); 0: in lockstep
#endif
#ifndef BNN_K1B2_SPREAD
#define BNN_K1B2_SPREAD 0   // build knob: the four staging pieces of a step spread over the previous step (K2_STAGE_PIECE) instead of all at its top.  Measured (profiles/r04_k1b2_spread.log): eps = 0 launches 304 -> 294 us, launches with the generator 348 -> 353: off
#endif
template <int NF, int SB, int EPS, int NB = 2, bool X3 = false>
__global__ __launch_bounds__(NF * SB * 64, (X3 ? 4 : BNN_K1B2_WPS)) void bbb_fwd_gemm2_kernel(const BbbK p) {
  constexpr int NW = NF * SB;
  static_assert(NB == 2 || NB == 3, "staging buffers (NB - 1 k-steps of DMA run-ahead)");
  constexpr int WPW = 4 / SB;                 // parameter pieces (of a tile's four) each of the SB waves of a tile brings
  constexpr int XPW = 8 / NF;                 // x pieces (batch tiles of its pair) each of the NF waves of a pair brings
  constexpr int XT = X3 ? 1024 : 512;         // float4s of one pair's x tile (X3: the hi plane's 8 pieces, then the lo plane's)
  static_assert(SB == 1 || SB == 2 || SB == 4, "pairs per block");
  static_assert(NF == 2 || NF == 4 || NF == 8, "feature waves per block");
  static_assert(!X3 || NB == 2, "split-bf16 form: two x buffers");
  constexpr bool PS = X3 || (BNN_K1B2_PS && NB == 2);   // parameters single-buffered (see above)
  // x buffers of the PS form: two, or (build knob BNN_K1B2_PS = 2, bf16 form) a ring of three -- the x pieces of step t + 2 are
  // requested during step t, behind the parameters of step t + 1, and stay in flight over the step's closing wait
  constexpr int XB = (PS && !X3 && BNN_K1B2_PS == 2) ? 3 : 2;
  constexpr int RB = PS ? XB : NB;            // buffers the step index cycles through
  constexpr int BUF = NF * 256 + SB * XT;     // float4s of one staging buffer of the double-buffered-everything layout
  // STAG (bf16 form, two pairs): the waves of the block's SECOND pair run their x reads + MFMAs of step t - 1 at the START of step
  // t.  A CU's SIMD holds wave fw of the first pair and wave fw of the second: in lockstep both queue for the vector-memory pipe
  // at the top of a step (half of a wave's step, tools/stamps_k1b2.py), both run their generator next, both their MFMAs last.
  // One phase apart, the second pair's LDS reads and MFMAs run beside the first pair's staging burst, its staging beside the
  // first pair's parameter reads, its generator beside the first pair's MFMAs.  The late pair keeps its weight fragment over the
  // barrier (4 registers) and its x tile one step longer: a ring of THREE x slots for that pair (the two in the staging buffers
  // + one more 8 KiB behind the bias table), since its waves request x of step t + 1 while others of them still read x of step
  // t - 1.  Same arithmetic in the same order per accumulator: same bits.
#if defined(BNN_TUNE) || defined(BNN_STAMPS)
  constexpr bool STAG = false;
#else
  constexpr bool STAG = BNN_K1B2_STAG && !X3 && !(BNN_K1B2_PS && NB == 2) && NB == 2 && SB == 2;
#endif
  // LDS image, in float4s.  !PS: [buffer][NF tiles' parameter pieces | SB pairs' x tiles], then the bias table.
  //                          PS: [parameter pieces][x buffer 0][x buffer 1].
  auto p_idx = [](int buf) { return PS ? 0 : buf * BUF; };
  auto x_idx = [](int buf) { return PS ? NF * 256 + buf * (SB * XT) : buf * BUF + NF * 256; };
  // ONE shared object (the guide's second-__shared__-object trap), the staging buffers first
  __shared__ __attribute__((aligned(16))) float4 sm_all[PS ? NF * 256 + XB * SB * XT : NB * BUF + NW * 4 + (STAG ? XT : 0)];
  float (*bias_s)[16] = reinterpret_cast<float (*)[16]>(sm_all + (PS ? 0 : NB * BUF));          // (!PS only)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the staging bases below live in SGPRs)
  const int fw = wave % NF, sb = wave / NF;
  const int r = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N, B = p.B;
  const int mbs = (B + 127) >> 7;
  const int U = p.S * mbs;
  int tb, ub, in_;
  if (!xcd2d_work_item(p.xc, tb, ub, in_)) return;             // block-uniform
  const int unit_raw = ub * SB + sb;
  const bool active = unit_raw < U;                           // the last block of a group may hold fewer than SB pairs
  const int unit = active ? unit_raw : U - 1;
  const int s = unit / mbs, mb = unit - s * mbs;
  const int tile = tb * NF + fw;
  const int n = tile * 16 + r;
  const bool n_ok = n < N;
  const int nc = min(n, N - 1);
  const int m0 = mb * 128;
  const int ksteps = (K + 31) >> 5;
  const uint32_t gs = global_sample(p, s);
  const bool do_stats = p.want_stats && mb == 0 && active;
  const bool do_ls = do_stats && s == 0;
  const bool do_dump = mb == 0 && active;
  const int gpr = (K + 3) >> 2;
  const uint32_t wid = p.layer_id * 4u;
  const __bf16* xs = reinterpret_cast<const __bf16*>(p.x) + (size_t)(s / p.xg) * (size_t)p.x_sstride;
  const __bf16* xs_lo = X3 ? reinterpret_cast<const __bf16*>(p.x_lo) + (size_t)(s / p.xg) * (size_t)p.x_sstride : nullptr;
  const int T = (N + 15) >> 4;

  if (do_stats && tb == 0 && unit == 0 && fw == 0 && lane == 0) p.ws[0] = make_float4(__int_as_float(T), 0.f, 0.f, 0.f);

  // ---- staging.  Parameter pieces of tile fw: piece j in {mu lo, mu hi, sigma lo, sigma hi}; wave (fw, sb) brings pieces
  // sb * WPW .. + WPW - 1.  x pieces of pair sb: batch tiles fw, fw + NF, ...
  const size_t wrow = (size_t)nc * K;
  size_t xrow[XPW];
#pragma unroll
  for (int i = 0; i < XPW; ++i) xrow[i] = (size_t)min(m0 + (fw + i * NF) * 16 + r, B - 1) * K;
  // The same pieces for a k-step whose 32 k lie inside K, with NO per-step vector arithmetic: the per-lane byte offsets are
  // loop-invariant 32-bit VGPRs, the step moves the 64-bit SCALAR base (global_load_lds with an SGPR base), the LDS
  // destination (M0) is a scalar sum.  As the compiler writes the builtin form below, a step's four pieces cost ~14 vector
  // instructions of address arithmetic plus 4 v_readfirstlane -- an eighth of the loop's vector issue, and the loop is bound
  // by exactly that (tools/k1b_ablate.py: the launch without its DMA is 70 us shorter, most of it these instructions).
  const uint32_t voff_w = ((uint32_t)nc * (uint32_t)K + (uint32_t)(q * 8)) * 4u;
  uint32_t voff_x[XPW];
#pragma unroll
  for (int i = 0; i < XPW; ++i) voff_x[i] = ((uint32_t)xrow[i] + (uint32_t)(q * 8)) * 2u;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&sm_all[0];
  // float4 index of this wave's pair's x tile `slot`: the pair's region of staging buffer `slot`, or (STAG, the late pair's
  // third slot) the region behind the bias table
  const bool late = STAG && sb == 1;                            // wave-uniform
  auto x_slot = [&](int slot) { return (STAG && slot == 2) ? NB * BUF + NW * 4 : x_idx(slot) + sb * XT; };
  // (the plan takes this form only where the tensors' byte spans fit 32 bits)
  // WP / WX (compile-time): stage the parameter pieces / the x pieces of step t (the PS form requests them at different points
  // of a step, into different buffer indices)
  auto stage_fast = [&](int t, int pbuf, int xbuf, auto wp_, auto wx_) __attribute__((always_inline)) {
    if (decltype(wp_)::value) {
#pragma unroll
      for (int i = 0; i < WPW; ++i) {
        const int j = sb * WPW + i;                               // wave-uniform
        const char* base = reinterpret_cast<const char*>((j & 2) ? p.w_sigma : p.w_mu) + (size_t)t * 128 + (j & 1) * 16;
        const uint32_t m0v = lds0 + (uint32_t)((p_idx(pbuf) + (fw * 4 + j) * 64) * 16);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_w), "s"(base), "s"(m0v) : "memory", "m0");
      }
    }
    if (decltype(wx_)::value) {
#pragma unroll
      for (int i = 0; i < XPW; ++i) {
        const char* base = reinterpret_cast<const char*>(xs) + (size_t)t * 64;
        const uint32_t m0v = lds0 + (uint32_t)((x_slot(xbuf) + (fw + i * NF) * 64) * 16);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_x[i]), "s"(base), "s"(m0v) : "memory", "m0");
      }
      if (X3) {
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
          const char* base = reinterpret_cast<const char*>(xs_lo) + (size_t)t * 64;
          const uint32_t m0v = lds0 + (uint32_t)((x_idx(xbuf) + sb * XT + 512 + (fw + i * NF) * 64) * 16);
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_x[i]), "s"(base), "s"(m0v) : "memory", "m0");
        }
      }
    }
  };
  // ONE piece of step t's staging (full steps only), pinned in the instruction stream by `tie`: a value computed just before and
  // consumed just after.  All four pieces of a wave issued back to back at the top of a step queue behind the other fifteen
  // waves' pieces in the CU's one vector-memory pipe (64 B per clock: the 64 KiB the CU's two blocks request per step take it
  // ~1000 cycles) and the wave cannot issue anything else meanwhile -- in-kernel stamps: 1370 of a wave-step's ~2800 cycles
  // between the top of a step and its parameter fragments (tools/stamps_k1b2.py, profiles/r04_k1b2_stamps.log).  Spread over
  // the step -- top, behind the parameter reads, behind the generator, behind the packing of w -- a piece meets a drained pipe.
  // (a macro, not a generic lambda: `tie` is a float, a uint32_t or a 4-register vector; IDX is a literal)
#define K2_STAGE_PIECE(T, BUF, IDX, TIE)                                                                                                   \
  do {                                                                                                                                     \
    if ((IDX) < WPW) {                                                                                                                     \
      const int j_ = sb * WPW + (IDX);                                                                                                     \
      const char* base_ = reinterpret_cast<const char*>((j_ & 2) ? p.w_sigma : p.w_mu) + (size_t)(T) * 128 + (j_ & 1) * 16;               \
      const uint32_t m0v_ = lds0 + (uint32_t)((p_idx(BUF) + (fw * 4 + j_) * 64) * 16);                                                     \
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : "+v"(TIE) : "v"(voff_w), "s"(base_), "s"(m0v_) : "memory", "m0"); \
    } else {                                                                                                                               \
      const char* base_ = reinterpret_cast<const char*>(xs) + (size_t)(T) * 64;                                                            \
      const uint32_t m0v_ = lds0 + (uint32_t)((x_idx(BUF) + sb * XT + (fw + ((IDX) - WPW) * NF) * 64) * 16);                               \
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : "+v"(TIE) : "v"(voff_x[((IDX) - WPW) % XPW]), "s"(base_), "s"(m0v_) : "memory", "m0"); \
    }                                                                                                                                      \
  } while (0)
  auto stage_slow = [&](int t, int pbuf, int xbuf, auto wp_, auto wx_) __attribute__((always_inline)) {
    const int kk = min(t * 32 + q * 8, K - 8);
    if (decltype(wp_)::value) {
#pragma unroll
      for (int i = 0; i < WPW; ++i) {
#ifdef BNN_TUNE
        const int j = ((p.tune & 64) ? (SB - 1 - sb) : sb) * WPW + i;
#else
        const int j = sb * WPW + i;                               // wave-uniform
#endif
        const float* src = ((j & 2) ? p.w_sigma : p.w_mu) + wrow + kk + (j & 1) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)&sm_all[p_idx(pbuf) + (fw * 4 + j) * 64], 16, 0, 0);
      }
    }
    if (decltype(wx_)::value) {
#pragma unroll
      for (int i = 0; i < XPW; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xs + xrow[i] + kk),
                                         (__attribute__((address_space(3))) void*)&sm_all[x_slot(xbuf) + (fw + i * NF) * 64], 16, 0, 0);
      }
      if (X3) {
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xs_lo + xrow[i] + kk),
                                           (__attribute__((address_space(3))) void*)&sm_all[x_idx(xbuf) + sb * XT + 512 + (fw + i * NF) * 64], 16, 0, 0);
        }
      }
    }
  };
  auto stage_sel = [&](int t, int pbuf, int xbuf, auto wp_, auto wx_) __attribute__((always_inline)) {
    if ((t + 1) * 32 <= K) stage_fast(t, pbuf, xbuf, wp_, wx_);   // block-uniform
    else stage_slow(t, pbuf, xbuf, wp_, wx_);                      // the K tail inside a k-step: clamped per-lane addresses
  };
  auto stage = [&](int t, int buf) __attribute__((always_inline)) { stage_sel(t, buf, buf, std::true_type{}, std::true_type{}); };

  // bias of this wave's tile: eps now, applied in the epilogue
  float bmu_pre = 0.f, brho_pre = 0.f, beps_pre = 0.f;
  if (q == 0 && n_ok) {
    bmu_pre = p.b_mu[n];
    brho_pre = p.b_rho[n];
    beps_pre = bias_eps(p, n, s, gs, do_dump);
  }
  stage(0, 0);
  if (PS && XB == 3) {
    stage_sel(ksteps > 1 ? 1 : 0, 0, 1, std::false_type{}, std::true_type{});      // x of step 1 (a one-step layer: a clamped re-read)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (NB == 3) {
    // two k-steps of run-ahead (build knob BNN_GEMM_RING=3, tools/build_k1b2_variants.sh): step 1 is in flight while step
    // 0's pieces are waited for (a one-step layer issues a clamped re-read of step 0 into the idle buffer 1 so that the
    // counted wait holds on every path)
    stage(ksteps > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + XPW) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s_e2 = 0.f, s_a = 0.f, s_ls = 0.f;

  // One k-step.  FULL (block-uniform, compile-time in the body): every lane of the block holds real weights in this step --
  // all four feature tiles inside N and the step's 32 k inside K -- so the edge masks (11 v_cndmask per step) are
  // compiled out; the statistics and w run two lanes-worth per instruction (v_pk_fma_f32).  The kernel is bound by its
  // vector instruction stream once the parameters come through LDS (tools/k1b_ablate.py), so every one of them counts.
  typedef __attribute__((ext_vector_type(2))) float f32x2;
#ifdef BNN_STAMPS
  // diagnostic build: shader-clock time of every wave's step phases, summed over the steps (tools/stamps_k1b2.py):
  // [0] staging issue + parameter reads, [1] generator / w / statistics, [2] x reads + MFMAs, [3] the closing vmcnt / lgkmcnt
  // wait, [4] the barrier
  unsigned long long ph[5] = {0, 0, 0, 0, 0};
#define K2_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define K2_T(v)
#endif
  // x reads + MFMAs of one k-step: the weight fragment `wa` (and its low half `wl`, X3) against the pair's x tile at float4 index xi
  auto mfma_phase = [&](const bf16x8& wa, const bf16x8& wl, int xi, const f32x4& m_lo) __attribute__((always_inline)) {
    const uint32_t xa = lds0 + (uint32_t)((xi + q * 16 + r) * 16);
#ifdef BNN_TUNE
    const bool tune_nolds = (p.tune & 16) != 0, tune_nomfma = (p.tune & 32) != 0;
#else
    constexpr bool tune_nolds = false, tune_nomfma = false;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 x0, x1, x2, x3;
      if (tune_nolds) {
        x0 = x1 = x2 = x3 = m_lo;
      } else if (h == 0)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(xa));
      else
        asm volatile("ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:5120\n\tds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(xa));
      if (X3) {
        // the lo plane's four fragments are requested behind the hi plane's: the hi products (two MFMAs per batch tile) run
        // while they are in flight
        f32x4 l0, l1, l2, l3;
        if (h == 0)
          asm volatile("ds_read_b128 %0, %4 offset:8192\n\tds_read_b128 %1, %4 offset:9216\n\tds_read_b128 %2, %4 offset:10240\n\tds_read_b128 %3, %4 offset:11264"
                       : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3) : "v"(xa));
        else
          asm volatile("ds_read_b128 %0, %4 offset:12288\n\tds_read_b128 %1, %4 offset:13312\n\tds_read_b128 %2, %4 offset:14336\n\tds_read_b128 %3, %4 offset:15360"
                       : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3) : "v"(xa));
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        acc[h * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x0), acc[h * 4 + 0], 0, 0, 0);
        acc[h * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x1), acc[h * 4 + 1], 0, 0, 0);
        acc[h * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x2), acc[h * 4 + 2], 0, 0, 0);
        acc[h * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x3), acc[h * 4 + 3], 0, 0, 0);
        acc[h * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, __builtin_bit_cast(bf16x8, x0), acc[h * 4 + 0], 0, 0, 0);
        acc[h * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, __builtin_bit_cast(bf16x8, x1), acc[h * 4 + 1], 0, 0, 0);
        acc[h * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, __builtin_bit_cast(bf16x8, x2), acc[h * 4 + 2], 0, 0, 0);
        acc[h * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, __builtin_bit_cast(bf16x8, x3), acc[h * 4 + 3], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
        acc[h * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, l0), acc[h * 4 + 0], 0, 0, 0);
        acc[h * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, l1), acc[h * 4 + 1], 0, 0, 0);
        acc[h * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, l2), acc[h * 4 + 2], 0, 0, 0);
        acc[h * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, l3), acc[h * 4 + 3], 0, 0, 0);
        continue;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      if (!tune_nomfma) {
        acc[h * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x0), acc[h * 4 + 0], 0, 0, 0);
        acc[h * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x1), acc[h * 4 + 1], 0, 0, 0);
        acc[h * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x2), acc[h * 4 + 2], 0, 0, 0);
        acc[h * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x3), acc[h * 4 + 3], 0, 0, 0);
      } else {
        asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(wa));
      }
    }
  };
  bf16x8 wa_prev = {};                                             // (STAG) the late pair's weight fragment of the step before
  // cur: the staging buffer of step t (t % RB); c3 (STAG): t % 3, the late pair's x slot of step t
  auto step = [&](int t, int cur, int c3, auto full_) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_)::value;
    const int k = t * 32 + q * 8;
    const bool lane_ok = FULL || (n_ok && k < K);
    K2_T(st0);
    if (STAG) {
      if (late && t > 0) mfma_phase(wa_prev, wa_prev, x_slot(c3 == 0 ? 2 : c3 - 1), f32x4{0.f, 0.f, 0.f, 0.f});   // wave-uniform branch
    }
    // the buffer staged here was last read in step t - 1 (barrier since)
    const bool staged = t + NB - 1 < ksteps;                     // block-uniform
#ifdef BNN_TUNE
    // tuning build only (wrong results; tools/k1b_ablate.py): BNN_TUNE_K1B bit 0 = no barrier, 1 = no waits, 3 = no DMA (the LDS
    // reads of the parameters stay: stale bytes, the same vector work), 4 = no x reads, 5 = no MFMAs
    if (staged && !(p.tune & 8)) {
      if (PS && XB == 2) stage_sel(t + 1, 0, cur ^ 1, std::false_type{}, std::true_type{});
      else if (!PS) stage(t + NB - 1, NB == 2 ? (cur ^ 1) : (cur == 0 ? 2 : cur - 1));
    }
#else
    // (spread: the four pieces of step t + 1 at four points of this step instead of here -- see stage_piece)
    const bool spread = BNN_K1B2_SPREAD && !STAG && !PS && NB == 2 && WPW + XPW == 4 && staged && (t + 2) * 32 <= K;     // block-uniform
    if (staged && !spread) {
      if (PS && XB == 2) stage_sel(t + 1, 0, cur ^ 1, std::false_type{}, std::true_type{});      // x of step t + 1 now, its parameters below
      else if (STAG) stage_sel(t + 1, cur ^ 1, late ? (c3 == 2 ? 0 : c3 + 1) : (cur ^ 1), std::true_type{}, std::true_type{});
      else if (!PS) stage(t + NB - 1, NB == 2 ? (cur ^ 1) : (cur == 0 ? 2 : cur - 1));
    }
    uint32_t tie0 = voff_w;
    if (spread) K2_STAGE_PIECE(t + 1, cur ^ 1, 2, tie0);            // an x piece first: the pieces longest in flight are the ones the
                                                                 // next step reads last
#endif
    const bool x_ahead = PS && XB == 3 && t + 2 < ksteps;        // block-uniform: this step requests the x pieces of step t + 2
    // LDS reads by hand (ds_read_b128 in asm): a compiler-visible read of `sm` would be ordered behind EVERY LDS-DMA in
    // flight that may alias it -- s_waitcnt vmcnt(0) right behind the prefetch this step has just issued -- although
    // buffer t & 1 was complete at the last barrier.  The "+v" operands of the wait tie the consumers to it; the outputs
    // are early-clobber: a result register must not be the address register of a later read of the same statement.
    const uint32_t pa = lds0 + (uint32_t)((p_idx(cur) + fw * 256 + lane) * 16);
    f32x4 m_lo, m_hi, g_lo, g_hi;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                 : "=&v"(m_lo), "=&v"(m_hi), "=&v"(g_lo), "=&v"(g_hi) : "v"(pa));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m_lo), "+v"(m_hi), "+v"(g_lo), "+v"(g_hi));
    if (PS) {
      // every wave of the block holds its (mu, sigma) fragments in registers: the one parameter buffer is free for step t + 1
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef BNN_TUNE
      if (t + 1 < ksteps && !(p.tune & 8)) stage_sel(t + 1, 0, 0, std::true_type{}, std::false_type{});
      if (x_ahead && !(p.tune & 8)) stage_sel(t + 2, 0, cur == 0 ? 2 : cur - 1, std::false_type{}, std::true_type{});
#else
      if (t + 1 < ksteps) stage_sel(t + 1, 0, 0, std::true_type{}, std::false_type{});
      if (x_ahead) stage_sel(t + 2, 0, cur == 0 ? 2 : cur - 1, std::false_type{}, std::true_type{});   // slot (cur + 2) % 3, behind the parameters
#endif
    }
    K2_T(st1);
#ifndef BNN_TUNE
    if (spread) K2_STAGE_PIECE(t + 1, cur ^ 1, 3, m_lo);
#endif
    const f32x2 mu2[4] = {{m_lo[0], m_lo[1]}, {m_lo[2], m_lo[3]}, {m_hi[0], m_hi[1]}, {m_hi[2], m_hi[3]}};
    const f32x2 sg2[4] = {{g_lo[0], g_lo[1]}, {g_lo[2], g_lo[3]}, {g_hi[0], g_hi[1]}, {g_hi[2], g_hi[3]}};
    float e[8];
    if (EPS == BNN_EPS_PHILOX) {
      const uint32_t g = (uint32_t)n * (uint32_t)gpr + (uint32_t)(k >> 2);
      uint4 pa_, pb_;
      philox_pair<>(g, gs, wid, p.k0, p.k1, pa_, pb_);
#ifndef BNN_TUNE
      if (spread) K2_STAGE_PIECE(t + 1, cur ^ 1, 0, pa_.x);          // between the integer and the transcendental half
#endif
      box_muller8(pa_, pb_, e);
    } else if (EPS == BNN_EPS_MEMORY) {
      load8<true>(p.eps_w + ((size_t)s * N + n) * K + k, lane_ok ? 8 : 0, e);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = 0.f;
    }
    if (p.eps_w_dump && do_dump) store8<true>(p.eps_w_dump + ((size_t)s * N + n) * K + k, lane_ok ? 8 : 0, e);
#ifndef BNN_TUNE
    if (spread) K2_STAGE_PIECE(t + 1, cur ^ 1, 1, e[0]);             // behind the generator (EPS != PHILOX: the third and fourth piece
    if (spread && EPS != BNN_EPS_PHILOX) K2_STAGE_PIECE(t + 1, cur ^ 1, 0, e[1]);   // both here)
#endif
    f32x2 w2[4], e2v = {0.f, 0.f}, av = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 ev = {e[2 * j], e[2 * j + 1]};
      w2[j] = __builtin_elementwise_fma(sg2[j], ev, mu2[j]);
      e2v = __builtin_elementwise_fma(ev, ev, e2v);
    }
    if (do_stats) {
      float a, ls = 0.f;
      if (p.prior_kind == BNN_PRIOR_GAUSS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) av = __builtin_elementwise_fma(w2[j], w2[j], av);
        a = av[0] + av[1];
      } else {
        a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a = add_log(a, mix_p(p, w2[j][0]));
          a = add_log(a, mix_p(p, w2[j][1]));
        }
      }
      if (do_ls) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ls = add_log(ls, sg2[j][0]);
          ls = add_log(ls, sg2[j][1]);
        }
      }
      const float e2 = e2v[0] + e2v[1];
      s_e2 += lane_ok ? e2 : 0.f;
      s_a += lane_ok ? a : 0.f;
      s_ls += lane_ok ? ls : 0.f;
    }
    bf16x8 wa, wl = {};
    if (X3) {
      // the split pair, two weights per instruction where the ISA has one: hi = cvt_pk(w0, w1); its two floats by a shift and
      // a mask of the packed word; the exact differences; lo = cvt_pk(d0, d1): 6 vector instructions per two weights
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
      uint32_t hw[4], lw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 wv = lane_ok ? w2[j] : f32x2{0.f, 0.f};
        const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(wv, bf16x2));
        const f32x2 hf = {__uint_as_float(hb << 16), __uint_as_float(hb & 0xffff0000u)};
        hw[j] = hb;
        lw[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(wv - hf, bf16x2));
      }
      wa = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
      wl = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wa[2 * j] = lane_ok ? (__bf16)w2[j][0] : (__bf16)0.f;
        wa[2 * j + 1] = lane_ok ? (__bf16)w2[j][1] : (__bf16)0.f;
      }
    }
#ifdef BNN_STAMPS
    asm volatile("" :: "v"(wa));
#endif
    K2_T(st2);
    if (STAG && late) wa_prev = wa;                                // its products at the top of the next step
    else mfma_phase(wa, wl, x_idx(cur) + sb * XT, m_lo);
    // this wave's DMA pieces of step t + 1 have landed and its LDS reads of buffer `cur` are back; then the block meets
    // (a bare s_barrier: __syncthreads()'s fence would add the same vmcnt(0)).  With three buffers the pieces of step
    // t + 2, issued at the top of this step, stay in flight: vector-memory operations complete in issue order, and
    // WPW + XPW of them are younger than step t + 1's pieces whenever this step staged anything.
#ifdef BNN_TUNE
    if (!(p.tune & 2)) {
      if (x_ahead) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(XPW) : "memory");
      else if (!PS && NB == 3 && staged) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WPW + XPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (!(p.tune & 1)) __builtin_amdgcn_s_barrier();
#else
#ifdef BNN_STAMPS
    asm volatile("" :: "v"(acc[0]), "v"(acc[7]));
    K2_T(st3);
#endif
    if (x_ahead) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(XPW) : "memory");     // the x pieces of step t + 2 stay in flight
    else if (!PS && NB == 3 && staged) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WPW + XPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    K2_T(st4);
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
#ifdef BNN_STAMPS
    K2_T(st5);
    ph[0] += st1 - st0; ph[1] += st2 - st1; ph[2] += st3 - st2; ph[3] += st4 - st3; ph[4] += st5 - st4;
#endif
  };
  {
    const bool tiles_full = (tb * NF + NF) * 16 <= N;           // block-uniform
    const int full_steps = tiles_full ? (K >> 5) : 0;           // steps whose 32 k are all inside K
    int t = 0, cur = 0, c3 = 0;
#pragma nounroll
    for (; t < full_steps; ++t, cur = (cur + 1 == RB ? 0 : cur + 1), c3 = (c3 == 2 ? 0 : c3 + 1)) step(t, cur, c3, std::true_type{});
#pragma nounroll
    for (; t < ksteps; ++t, cur = (cur + 1 == RB ? 0 : cur + 1), c3 = (c3 == 2 ? 0 : c3 + 1)) step(t, cur, c3, std::false_type{});
    if (STAG) {
      if (late) mfma_phase(wa_prev, wa_prev, x_slot(c3 == 0 ? 2 : c3 - 1), f32x4{0.f, 0.f, 0.f, 0.f});   // the late pair's last step (ksteps >= 1)
    }
  }

#ifdef BNN_STAMPS
  if (p.dbg && lane == 0 && blockIdx.x < 1024) {               // per wave: the five phase sums + the step count
    unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
    d[0] = ph[0]; d[1] = ph[1]; d[2] = ph[2]; d[3] = ph[3]; d[4] = ph[4]; d[5] = (unsigned long long)ksteps;
  }
#endif
#undef K2_T
#undef K2_STAGE_PIECE
  // ---- epilogue: bias, stats, store (no cross-wave reduction)
  float b_own = 0.f;                                           // lanes 0 .. 15 (q == 0): the sampled bias of feature r
  if (q == 0) {
    if (n_ok) b_own = sample_bias(p, bmu_pre, brho_pre, beps_pre, do_stats, do_ls, s_e2, s_a, s_ls);
    if (!PS) bias_s[wave][r] = b_own;
  }
  if (do_stats) {
    const float a = wave_sum(s_e2), b = wave_sum(s_a), cc = wave_sum(s_ls);
    if (lane == 0 && tile < T) p.ws[1 + (size_t)s * T + tile] = make_float4(a, b, cc, 0.f);
  }
  float bq[4];
  if (PS) {                                                    // the PS form has no LDS left for a bias table: lane shuffles
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = __shfl(b_own, q * 4 + i, 64);
  } else {
    __syncthreads();
  }
  if (!active) return;
  const int nb = tile * 16 + q * 4;
  const bool vec_ok = (N & 3) == 0;
  if (!PS) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = bias_s[wave][q * 4 + i];
  }
  if (nb < N) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int brow = m0 + m * 16 + r;
      if (brow < B) {
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
          bf16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
          if (vec_ok) {
            *reinterpret_cast<bf16x4*>(yp) = o;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (nb + i < N) yp[i] = o[i];
          }
          if (X3) {                                  // the low plane of the split pair
            __bf16* lp = reinterpret_cast<__bf16*>(p.y_lo) + yoff;
            bf16x4 l;
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = (__bf16)(v[i] - (float)o[i]);
            if (vec_ok) {
              *reinterpret_cast<bf16x4*>(lp) = l;
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (nb + i < N) lp[i] = l[i];
            }
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

// The block GEMM carrying an independent sampling job (bnn_bbb_fwd_args.rider) as extra 256-thread blocks behind its
// own (whose count is a multiple of 8, so the XCD-aware work order of the layer's blocks is unchanged).
template <bool SIG, int EPS>
__global__ __launch_bounds__(256, 4) void bbb_fwd_gemm_rider_kernel(const BbbK p, const SampleK sk, int n_main) {
  __shared__ __attribute__((aligned(16))) float4 xt[2][8 * 64];
  __shared__ float bias_s[4][16];
  if ((int)blockIdx.x >= n_main) {                         // block-uniform
    sample_block(sk, (int)blockIdx.x - n_main, &bias_s[0][0]);
    return;
  }
  bbb_gemm_body<4, SIG, EPS>(p, xt, bias_s);
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

// One float4 entry per (sample, statistics writer): the narrowest tile form has ceil(N/4) writers per sample, the
// K-sliced GEMM form kMaxSlices per 16-feature tile.
static constexpr int kMaxSlices = 8;
extern "C" size_t bnn_bbb_linear_fwd_workspace_bytes(int32_t n_samples, int32_t out_features) {
  if (n_samples <= 0 || out_features <= 0) return 0;
  return (1 + (size_t)n_samples * (size_t)(((out_features + 15) / 16) * kMaxSlices)) * 4 * sizeof(float);
}

// Scratch of the K-sliced GEMM form: [arrival counters: one uint32 per (64-feature group, sample, 128-row batch block)
// unit, padded to 256 bytes | fp32 partial tiles: unit x kMaxSlices x 32 KiB].  The counter region must be zero
// before the FIRST launch that uses the scratch (every launch leaves it zero again); the partial region needs no
// initialisation.  One scratch serves one launch at a time (launches on one stream, or one scratch per stream).
static size_t split_ticket_bytes(long units) { return (((size_t)units * 4 + 255) / 256) * 256; }
extern "C" size_t bnn_bbb_split_scratch_bytes(int32_t n_samples, int32_t batch, int32_t out_features) {
  if (n_samples <= 0 || batch <= 0 || out_features <= 0) return 0;
  const long units = (long)((out_features + 63) / 64) * n_samples * ((batch + 127) / 128);
  return split_ticket_bytes(units) + (size_t)units * kMaxSlices * (4 * 8 * 64 * 16);
}
extern "C" size_t bnn_bbb_split_scratch_zero_bytes(int32_t n_samples, int32_t batch, int32_t out_features) {
  if (n_samples <= 0 || batch <= 0 || out_features <= 0) return 0;
  return split_ticket_bytes((long)((out_features + 63) / 64) * n_samples * ((batch + 127) / 128));
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

namespace {
// Launch geometry of K1: a pure function of the shape and of what the arguments allow (bnn_bbb_plan exports it).
struct BbbPlan {
  int form;         // bnn_form
  int R, nw, mt;    // TILE: k-range classes, waves per block, 16-row batch tiles per block
  int tiles;        // TILE: feature tiles
  int ksl;          // GEMM_KSLICE: slices
  int pairs;        // GEMM: (sample, batch block) units per block (K1b2; 1 = K1b)
  long blocks;
  size_t lds;
};

constexpr size_t kL2WeightBudget = 2560 * 1024;   // of an XCD's 4 MiB L2: the (mu, rho | sigma) of the feature groups a class of the
                                                  // 2-D work order keeps resident while the units' x tiles stream through
#ifndef BNN_GEMM_PAIRS
#define BNN_GEMM_PAIRS 2
#endif
#ifndef BNN_GEMM_RING
#define BNN_GEMM_RING 2
#endif
constexpr int kGemmPairs = BNN_GEMM_PAIRS;   // K1b2: units that share a block's parameter tiles
constexpr int kGemmRing = BNN_GEMM_RING;     // K1b2: LDS staging buffers (k-steps of DMA run-ahead + 1); 2 x 2 is the measured
                                             // best of {2, 4} pairs x {2, 3} buffers (profiles/r03_k1b2_variants.log)
constexpr int kGemmPairsX3 = 2;              // ... of the split-bf16 variant (48 KiB per staging buffer)
constexpr int kGemmMinBlocks = 450;          // block-GEMM form from this many (64-feature group x sample x batch block) items
constexpr int kSliceMaxBlocks = 2048;        // K-range slices are considered up to this many blocks (1024 are resident at once)
constexpr long kSliceMinWeights = 250000;    // ... for layers of at least this many weights
constexpr int kBlockGemmMinBatch = 512;     // matmul half over sampled weights: the 256 x 256 block form from this many batch rows
constexpr int kSliceMinSamples = 4;          // ... from this many samples per launch (below: the tile form's narrow tiles)

// TILE form.  Narrower tiles (more k-range classes per MFMA tile) until the launch covers the chip; the block's waves
// divide the super-steps evenly.
void tile_plan(int S, int B, int K, int N, bool aligned, int mt, BbbPlan& pl) {
  const int mbs = (B + 16 * mt - 1) / (16 * mt);
  int R = 1;
  if (aligned)
    while (R < 4 && (long)((N + 16 / R - 1) / (16 / R)) * S * mbs < 120) R *= 2;
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
  // one-sample launch of the 1200x1200 layer)
  else if (ssteps >= 4) nw = (ssteps >= 12 ? 12 : ssteps) & ~3;
  if (nw > ssteps) nw = ssteps;
  pl.form = BNN_FORM_TILE;
  pl.R = R;
  pl.nw = nw < 1 ? 1 : nw;
  pl.mt = mt;
  pl.tiles = (N + F - 1) / F;
  pl.ksl = 1;
  pl.blocks = (long)pl.tiles * S * mbs;
  pl.lds = ((size_t)pl.nw * mt * 64 * 4 + 16 + 3 * pl.nw) * sizeof(float);
}

// K-range slices of the GEMM form for `gemm_blocks` (group x sample x batch block) units.  The generator work of a
// launch is ksteps wave-steps per (16-feature tile, sample) whatever the decomposition (~0.6 us of vector issue each);
// what the slicing chooses is how evenly that falls on the 256 CUs and how many waves a SIMD has to interleave.  Blocks
// are 4 waves, one per SIMD, up to four blocks resident per CU, so a CU's time is (blocks it holds) x (k-steps per
// slice) x a stretch for what 1 / 2 / 3 / 4 waves per SIMD leave uncovered -- measured (tools/few_sample_sweep.py,
// 1200 x 1200): a lone wave runs a step in ~3500 cycles, 2.4 x its issue cost; two, three, four interleaved waves
// take 1.4 / 1.27 / 1.2 x -- plus ~5 us of launch boundary, prologue and hand-off, plus the slices' fp32 partial tiles
// (64 KiB written and read per slice and unit), which all leave at the end of the launch when every block is resident
// (~12 MB/us) and hide behind later blocks otherwise.  E.g. 1200 x 1200, 8 samples: 152 units; whole-K blocks would
// put 38 steps on 152 of the CUs, 4 slices 3 x 10 = 30 on all, 5 slices 3 x 8 = 24 (the ideal is 22.3).
double slice_cost_us(long gemm_blocks, int ksteps, int ksl) {
  const long blocks = gemm_blocks * ksl;
  // (1024 blocks are resident at once, four per CU.  More blocks than that, of SHORT slices, run as a further round: priced as
  // ceil(blocks / 256) per CU, the 1330 blocks of 14 samples x 5 slices of 5 k-steps on the 784-wide layer looked cheaper than 798
  // blocks of 3 slices and the evaluation measured 5 % slower; long slices hide the refill -- 32 samples x 2 slices of 19 steps,
  // 1216 blocks, is 3 % faster than one round of whole-K blocks: tools/ksl_eval_sweep.py, profiles/r04_ksl_eval.log)
  const bool short_slices = (ksteps + ksl - 1) / ksl < 8;
  const long per_cu = (blocks <= 1024 || !short_slices) ? (blocks + 255) / 256 : 4 * ((blocks + 1023) / 1024);
  static const double stretch[5] = {0.0, 2.4, 1.4, 1.27, 1.2};
  // (+1.5 steps per block: its prologue and epilogue bubbles)
  double us = 0.5 * (double)per_cu * ((ksteps + ksl - 1) / ksl + 1.5) * stretch[per_cu > 4 ? 4 : per_cu] + 3.2;
  if (ksl > 1) {
    const double mb = (double)blocks * 65536.0 / 1.0e6;
    us += 1.8 + mb / 12.0 * (blocks > 1024 ? 1024.0 / (double)blocks : 1.0);
  }
  return us;
}

int kslices(long gemm_blocks, int K, int min_ksl = 1, int forced = 0) {
  const int ksteps = (K + 31) / 32;
  if (forced > 0) return forced > kMaxSlices ? kMaxSlices : (forced > ksteps ? ksteps : forced);
  if (min_ksl * 2 > ksteps) return 1;
  int best = min_ksl;
  double best_cost = 0;
  for (int ksl = min_ksl; ksl <= kMaxSlices && (ksl == 1 || ksl * 2 <= ksteps); ++ksl) {
    if (ksl > min_ksl && gemm_blocks * ksl > kSliceMaxBlocks) break;
    const double cost = slice_cost_us(gemm_blocks, ksteps, ksl);
    if (ksl == min_ksl || cost < best_cost - 1e-9) {
      best_cost = cost;
      best = ksl;
    }
  }
  return best;
}

template <typename KernelT>
hipError_t allow_big_lds(KernelT kernel, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             160 * 1024);
}
}  // namespace

// `al`: the 16-byte vector path applies (K % 8 == 0, aligned bases).  Returns a bnn_status.
static int bbb_plan(const bnn_bbb_fwd_args* a, bool al, BbbPlan& pl, bool allow_block = true) {
  const int S = a->n_samples, B = a->batch, K = a->in_features, N = a->out_features;
  const int mbs = (B + 127) / 128;
  if (allow_block && a->w_sampled && a->x_dtype == BNN_BF16 && !a->w_sampled_t_out && !a->y_bf16_copy &&
      (a->form == BNN_FORM_BLOCK256 || (a->form == BNN_FORM_AUTO && B >= kBlockGemmMinBatch)) &&
      (double)(B > N ? B : N) * K * 2.0 < 2147483648.0) {
    // matmul half over >= 512 batch rows: 2 * batch flops per sampled weight -- the matrix-core-bound block form (K1g)
    pl.form = BNN_FORM_BLOCK256;
    pl.R = 1; pl.nw = 8; pl.mt = 16; pl.tiles = (N + 15) / 16; pl.ksl = 1;
    pl.blocks = (long)S * ((B + 255) / 256) * ((N + 255) / 256);
    pl.lds = kBgLds;
    return BNN_OK;
  }
  if (a->w_sampled) {
    // matmul half only: same tile machinery, no generator work.  Without it a block's cost is the x it pulls through
    // its CU's L1 -- all of K for its rows -- so 32-row batch blocks (four times the blocks, each ingesting a quarter)
    // and whole 16-feature tiles (5.2 against 8.7 us at 128 x 1200 x 1200).
    tile_plan(S, B, K, N, true, 8, pl);            // the wave count of the sampling form of this shape
    pl.R = 1;
    pl.mt = 2;
    pl.tiles = (N + 15) / 16;
    pl.blocks = (long)pl.tiles * S * ((B + 31) / 32);
    pl.lds = ((size_t)pl.nw * 2 * 64 * 4 + 16 + 3 * pl.nw) * sizeof(float);
    return BNN_OK;
  }
  const long gemm_blocks = (long)((N + 63) / 64) * S * mbs;
  // (the block-GEMM kernels address with a scalar base + 32-bit lane offsets: the tensors' byte spans must fit)
  // split-bf16 math: the block-GEMM form exists as K1b2 only (hoisted sigma, no rider, pairs of units; no K slices)
  const bool x3 = a->math == BNN_MATH_BF16X3;
  const bool gemm_ok = al && (a->math == BNN_MATH_BF16 || x3) && a->x_dtype == BNN_BF16 && K >= 8 &&
                       (double)N * K * 4.0 < 4294967295.0 && (double)B * K * 2.0 < 4294967295.0 &&
                       (!x3 || (a->w_sigma && !a->rider && (long)S * mbs >= 2 * kGemmPairsX3 && !(reinterpret_cast<uintptr_t>(a->x_lo) & 15) &&
                                (a->y_dtype != BNN_BF16 || !(reinterpret_cast<uintptr_t>(a->y_lo) & 7))));
  const bool slice_ok = gemm_ok && !x3 && a->split_scratch && !(reinterpret_cast<uintptr_t>(a->split_scratch) & 15) &&
                        a->split_scratch_bytes >= bnn_bbb_split_scratch_bytes(S, B, N);
  int forced = 0;
#ifdef BNN_TUNE
  if (const char* v = getenv("BNN_TUNE_KSL")) forced = atoi(v);
#endif
  // a preference for the sliced form asks for at least two slices
  const int ksl = slice_ok ? kslices(gemm_blocks, K, a->form == BNN_FORM_GEMM_KSLICE ? 2 : 1, forced) : 1;
  // a->form is a preference: taken when the arguments allow that form, otherwise the plan's own choice
  int form = a->form;
  if ((form == BNN_FORM_GEMM && !gemm_ok) || (form == BNN_FORM_GEMM_KSLICE && ksl <= 1)) form = BNN_FORM_AUTO;
  if (a->y_bf16_copy) form = BNN_FORM_TILE;             // an epilogue of the tile forms
  if (form == BNN_FORM_AUTO) {
    if (ksl > 1 && (long)K * N >= kSliceMinWeights && S >= kSliceMinSamples) form = BNN_FORM_GEMM_KSLICE;
    else if (gemm_ok && gemm_blocks >= kGemmMinBlocks) form = BNN_FORM_GEMM;
    else form = BNN_FORM_TILE;
  }
  if (form == BNN_FORM_TILE) {
    tile_plan(S, B, K, N, al, 8, pl);
    return BNN_OK;
  }
  pl.form = form;
  pl.R = 1; pl.nw = 4; pl.mt = 8; pl.tiles = (N + 15) / 16;
  pl.ksl = form == BNN_FORM_GEMM_KSLICE ? ksl : 1;
  pl.blocks = gemm_blocks * pl.ksl;
  pl.lds = 2 * 8 * 64 * 16 + 4 * 16 * sizeof(float);
  pl.pairs = 1;
  // K1b2: the plain block-GEMM form with hoisted sigma and no rider shares the parameters of a k-step between
  // kGemmPairs (minibatch | sample, batch block) units of one block through LDS
  bool pairs_on = true;
#ifdef BNN_TUNE
  if (const char* v = getenv("BNN_TUNE_PAIRS")) pairs_on = atoi(v) != 0;
#endif
  const int pairs = x3 ? kGemmPairsX3 : kGemmPairs;
  if (pairs_on && form == BNN_FORM_GEMM && a->w_sigma && !a->rider && (long)S * mbs >= 2 * pairs) {
    pl.pairs = pairs;
    pl.nw = 4 * pairs;
    pl.blocks = (long)((N + 63) / 64) * (((long)S * mbs + pairs - 1) / pairs);
    // (split-bf16 form: one parameter buffer + two x buffers of a (hi, lo) plane pair per unit, no bias table)
    pl.lds = x3 ? (4 * 256 + 2 * pairs * 1024) * 16 : kGemmRing * (4 * 256 + pairs * 512) * 16 + pl.nw * 16 * sizeof(float) +
                      ((BNN_K1B2_STAG && !BNN_K1B2_PS && kGemmRing == 2 && pairs == 2) ? 512 * 16 : 0);   // (the late pair's third x slot)
  }
  if (x3 && pl.pairs == 1) {                 // (the plain block-GEMM kernel has no split-bf16 variant)
    tile_plan(S, B, K, N, al, 8, pl);
    return BNN_OK;
  }
  return BNN_OK;
}

// Validate the arguments and fill the kernel parameter block.  `al` = the 16-byte vector path
// applies (K % 8 == 0, aligned bases).
static int prepare(const bnn_bbb_fwd_args* a, BbbK& k, bool& al) {
  k.mask = nullptr; k.mask_sstride = 0;      // input-gradient form only
  k.w_pre = nullptr; k.b_pre = nullptr; k.wt_out = nullptr;     // set below for the matmul-only form
  if (a && a->w_sampled_t_out && !a->w_sampled) return BNN_ERR_NULL;
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_bbb_fwd_args)) return BNN_ERR_ABI;
  if (a->n_samples <= 0 || a->batch <= 0 || a->in_features <= 0 || a->out_features <= 0) return BNN_ERR_SHAPE;
  if ((double)a->n_samples * ((a->batch + 127) / 128) * ((a->out_features + 3) / 4) > 2.0e9) return BNN_ERR_SHAPE;
  if (a->x_per_sample < 0) return BNN_ERR_SHAPE;
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
  if ((unsigned)a->x_dtype > 1u || (unsigned)a->y_dtype > 1u || (unsigned)a->math > 2u || (unsigned)a->eps_mode > 2u ||
      (unsigned)a->prior.kind > 1u || (unsigned)a->form > 4u)
    return BNN_ERR_ENUM;
  if (!pre && a->eps_mode == BNN_EPS_MEMORY && (!a->eps_w || !a->eps_b)) return BNN_ERR_NULL;
  if (a->want_stats) {
    if (!a->workspace || a->workspace_bytes < bnn_bbb_linear_fwd_workspace_bytes(a->n_samples, a->out_features))
      return BNN_ERR_WORKSPACE;
    if (!aligned16(a->workspace)) return BNN_ERR_ALIGN;
  }
  if ((a->log_prior || a->log_q) && !a->want_stats) return BNN_ERR_WORKSPACE;
  k.x_lo = nullptr; k.y_lo = nullptr;
  if (a->math == BNN_MATH_BF16X3) {
    // split-bf16 math: bf16 activations are pairs of planes; a forward sampling form only
    if (pre || a->y_bf16_copy || a->rider) return BNN_ERR_ENUM;
    if (a->x_dtype == BNN_BF16) {
      if (!a->x_lo) return BNN_ERR_NULL;
      if (reinterpret_cast<uintptr_t>(a->x_lo) & 1) return BNN_ERR_ALIGN;
      k.x_lo = a->x_lo;
    }
    if (a->y_dtype == BNN_BF16) {
      if (!a->y_lo) return BNN_ERR_NULL;
      if ((a->out_features % 4 == 0) && (reinterpret_cast<uintptr_t>(a->y_lo) & 7)) return BNN_ERR_ALIGN;
      k.y_lo = a->y_lo;
    }
  }
  k.x = a->x;
  k.x_sstride = a->x_per_sample ? (long)a->batch * a->in_features : 0;
  k.xg = a->x_per_sample > 0 ? a->x_per_sample : 1;
  k.sgrp = a->sample_group; k.sgrp_stride = a->sample_group_stride;
  k.w_mu = a->w_mu; k.w_rho = a->w_rho; k.b_mu = a->b_mu; k.b_rho = a->b_rho;
  k.w_sigma = a->w_sigma;
  if (a->w_sigma && (reinterpret_cast<uintptr_t>(a->w_sigma) & 15)) return BNN_ERR_ALIGN;
  k.eps_w = a->eps_w; k.eps_b = a->eps_b; k.eps_w_dump = a->eps_w_dump; k.eps_b_dump = a->eps_b_dump;
  k.y = a->y;
  k.ws = a->want_stats ? reinterpret_cast<float4*>(a->workspace) : nullptr;
  k.S = a->n_samples; k.B = a->batch; k.K = a->in_features; k.N = a->out_features;
  k.eps_mode = a->eps_mode; k.prior_kind = a->prior.kind; k.want_stats = a->want_stats ? 1 : 0;
  k.y16 = reinterpret_cast<__bf16*>(a->y_bf16_copy);
  if (a->y_bf16_copy && a->y_dtype != BNN_F32) return BNN_ERR_ENUM;
  k.relu = a->relu ? 1 : 0; k.y_bf16 = a->y_dtype == BNN_BF16; k.spb = 1; k.ksl = 1; k.ks_part = nullptr; k.ks_ticket = nullptr; k.ldw = 0;
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
  al = (K % 8 == 0) && aligned16(a->x) && aligned16(a->w_mu) && aligned16(a->w_rho) && (!k.x_lo || aligned16(k.x_lo));
  if (pre) {
    k.w_pre = reinterpret_cast<const __bf16*>(a->w_sampled);
    k.b_pre = a->b_sampled;
    k.wt_out = reinterpret_cast<__bf16*>(a->w_sampled_t_out);
    k.eps_mode = BNN_EPS_ZERO;
    al = true;
  } else if (a->eps_mode == BNN_EPS_MEMORY) al = al && aligned16(a->eps_w);
  if (a->eps_w_dump) al = al && aligned16(a->eps_w_dump);
  const bool ybf = a->y_dtype == BNN_BF16;
  if ((a->out_features % 4 == 0) && (reinterpret_cast<uintptr_t>(a->y) & (ybf ? 7 : 15))) return BNN_ERR_ALIGN;
  return BNN_OK;
}

extern "C" int bnn_bbb_plan(const bnn_bbb_fwd_args* a, bnn_plan* out) {
  if (!out) return BNN_ERR_NULL;
  BbbK k;
  bool al = false;
  int rc = prepare(a, k, al);
  if (rc != BNN_OK) return rc;
  BbbPlan pl{};
  rc = bbb_plan(a, al, pl);
  if (rc != BNN_OK) return rc;
  out->form = pl.form;
  out->k_classes = pl.R;
  out->waves = pl.nw;
  out->batch_rows = 16 * pl.mt;
  out->k_slices = pl.ksl;
  out->blocks = (int32_t)pl.blocks;
  out->lds_bytes = (int32_t)pl.lds;
  out->features_per_block = pl.form == BNN_FORM_TILE ? 16 / pl.R : pl.form == BNN_FORM_BLOCK256 ? 256 : 64;
  return BNN_OK;
}

// `grad_mask` != nullptr: the launch is the INPUT GRADIENT of a layer as the matmul gz . w over that layer's transposed
// sampled weights (bnn_bbb_input_grad_): no bias, the output masked by (grad_mask > 0)
static int bbb_linear_fwd_impl(const bnn_bbb_fwd_args* a, void* stream_, bool no_bias, const float* grad_mask, long mask_sstride) {
  BbbK k;
  bool al = false;
  int rc = prepare(a, k, al);
  if (rc != BNN_OK) return rc;
  if (no_bias) k.b_pre = nullptr;
  if (grad_mask) {
    k.mask = grad_mask;
    k.mask_sstride = mask_sstride;
  }
  BbbPlan pl{};
  rc = bbb_plan(a, al, pl, grad_mask == nullptr);    // the masked input-gradient epilogue belongs to the tile form
  if (rc != BNN_OK) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int K = a->in_features;
  const int xdt = a->x_dtype, math = a->math;
  hipError_t err = hipSuccess;
  const dim3 grid((unsigned)(((pl.blocks + 7) / 8) * 8)), block(pl.nw * 64);
  // an independent sampling job rides on a tile-form launch as extra blocks (whole 256-thread sampling blocks);
  // otherwise it is a launch of its own ahead of the layer
  SampleK sk;
  long rider_blocks = 0;
  bool ride = false;
  if (a->rider) {
    rc = fill_sample(a->rider, sk, rider_blocks);
    if (rc != BNN_OK) return rc;
    ride = !a->w_sampled && al && math == BNN_MATH_BF16 && pl.nw * 64 >= kSampleThreads;
    if (!ride) {
      rc = bnn_bbb_sample_weights(a->rider, stream_);
      if (rc != BNN_OK) return rc;
    }
  }
  if (pl.form == BNN_FORM_BLOCK256) {
    if (a->rider && ride) return BNN_ERR_SHAPE;      // (never: ride needs !w_sampled)
    BlockGemmK g{};
    g.x = reinterpret_cast<const __bf16*>(a->x);
    g.x_sstride = k.x_sstride; g.xg = k.xg;
    g.w = k.w_pre; g.bias = k.b_pre; g.y = a->y;
    g.y_bf16 = k.y_bf16; g.relu = k.relu;
    g.S = a->n_samples; g.M = a->batch; g.N = a->out_features; g.K = K;
    err = launch_block_gemm(g, stream);
    return err == hipSuccess ? BNN_OK : (int)err;
  }
  if (a->w_sampled) {
#define BNN_PRE(XDT)                                                                              \
  do {                                                                                            \
    err = allow_big_lds(bbb_fwd_pre_kernel<XDT, 1, 2>, pl.lds);                                   \
    if (err == hipSuccess)                                                                        \
      hipLaunchKernelGGL((bbb_fwd_pre_kernel<XDT, 1, 2>), grid, block, pl.lds, stream, k);        \
  } while (0)
    if (k.wt_out) {                                  // the same launch also leaves w transposed
      err = allow_big_lds(bbb_fwd_pre_kernel<BNN_BF16, 1, 2, true>, pl.lds);
      if (xdt != BNN_BF16) return BNN_ERR_ENUM;
      if (err == hipSuccess) hipLaunchKernelGGL((bbb_fwd_pre_kernel<BNN_BF16, 1, 2, true>), grid, block, pl.lds, stream, k);
    } else if (xdt == BNN_F32) BNN_PRE(BNN_F32); else BNN_PRE(BNN_BF16);
#undef BNN_PRE
    if (err != hipSuccess) return (int)err;
    err = hipGetLastError();
    return err == hipSuccess ? BNN_OK : (int)err;
  }
  if (pl.form != BNN_FORM_TILE) {
    // Throughput form (block GEMM, x tile shared through LDS); K-range slices leave fp32 partial tiles in the caller's
    // split scratch and the last-arriving slice block of each unit sums them in slice order (deterministic), applies
    // ReLU and the down-conversion: one launch.
    k.ksl = pl.ksl;
#ifdef BNN_TUNE
    k.tune = getenv("BNN_TUNE_K1B") ? atoi(getenv("BNN_TUNE_K1B")) : 0;
#endif
    {
      size_t l2_budget = kL2WeightBudget;
#ifdef BNN_TUNE
      if (const char* v = getenv("BNN_TUNE_L2KB")) l2_budget = (size_t)atol(v) * 1024;
#endif
      const int mbs_ = (a->batch + 127) / 128;
      k.xc = xcd2d_make((a->out_features + 63) / 64, a->n_samples * mbs_, pl.ksl, (size_t)64 * K * 8, l2_budget);
    }
    if (pl.ksl > 1) {
      char* base = reinterpret_cast<char*>(a->split_scratch);
      k.ks_ticket = reinterpret_cast<uint32_t*>(base);
      k.ks_part = reinterpret_cast<float4*>(base + split_ticket_bytes(pl.blocks / pl.ksl));
    }
    if (pl.pairs > 1) {
      const int ub = (int)(((long)a->n_samples * ((a->batch + 127) / 128) + pl.pairs - 1) / pl.pairs);
      k.xc = xcd2d_make((a->out_features + 63) / 64, ub, 1, (size_t)64 * K * 8, kL2WeightBudget);
      const dim3 grid2((unsigned)(((pl.blocks + 7) / 8) * 8)), block2(pl.nw * 64);
      if (math == BNN_MATH_BF16X3) {
        if (k.eps_mode == BNN_EPS_PHILOX) hipLaunchKernelGGL((bbb_fwd_gemm2_kernel<4, kGemmPairsX3, BNN_EPS_PHILOX, 2, true>), grid2, block2, 0, stream, k);
        else if (k.eps_mode == BNN_EPS_MEMORY) hipLaunchKernelGGL((bbb_fwd_gemm2_kernel<4, kGemmPairsX3, BNN_EPS_MEMORY, 2, true>), grid2, block2, 0, stream, k);
        else hipLaunchKernelGGL((bbb_fwd_gemm2_kernel<4, kGemmPairsX3, BNN_EPS_ZERO, 2, true>), grid2, block2, 0, stream, k);
      } else if (k.eps_mode == BNN_EPS_PHILOX) hipLaunchKernelGGL((bbb_fwd_gemm2_kernel<4, kGemmPairs, BNN_EPS_PHILOX, kGemmRing>), grid2, block2, 0, stream, k);
      else if (k.eps_mode == BNN_EPS_MEMORY) hipLaunchKernelGGL((bbb_fwd_gemm2_kernel<4, kGemmPairs, BNN_EPS_MEMORY, kGemmRing>), grid2, block2, 0, stream, k);
      else hipLaunchKernelGGL((bbb_fwd_gemm2_kernel<4, kGemmPairs, BNN_EPS_ZERO, kGemmRing>), grid2, block2, 0, stream, k);
    } else if (ride) {
      const unsigned n_main = grid.x;
      const dim3 grid_r(n_main + (unsigned)rider_blocks);
#define BNN_GEMM_EPS(KERNEL, SIGV, ...)                                                                        \
  do {                                                                                                         \
    if (k.eps_mode == BNN_EPS_PHILOX) hipLaunchKernelGGL((KERNEL<SIGV, BNN_EPS_PHILOX>), __VA_ARGS__);          \
    else if (k.eps_mode == BNN_EPS_MEMORY) hipLaunchKernelGGL((KERNEL<SIGV, BNN_EPS_MEMORY>), __VA_ARGS__);     \
    else hipLaunchKernelGGL((KERNEL<SIGV, BNN_EPS_ZERO>), __VA_ARGS__);                                         \
  } while (0)
      if (a->w_sigma) BNN_GEMM_EPS(bbb_fwd_gemm_rider_kernel, true, grid_r, block, 0, stream, k, sk, (int)n_main);
      else BNN_GEMM_EPS(bbb_fwd_gemm_rider_kernel, false, grid_r, block, 0, stream, k, sk, (int)n_main);
    } else if (a->w_sigma) {
      if (k.eps_mode == BNN_EPS_PHILOX) hipLaunchKernelGGL((bbb_fwd_gemm_kernel<4, true, BNN_EPS_PHILOX>), grid, block, 0, stream, k);
      else if (k.eps_mode == BNN_EPS_MEMORY) hipLaunchKernelGGL((bbb_fwd_gemm_kernel<4, true, BNN_EPS_MEMORY>), grid, block, 0, stream, k);
      else hipLaunchKernelGGL((bbb_fwd_gemm_kernel<4, true, BNN_EPS_ZERO>), grid, block, 0, stream, k);
    } else {
      if (k.eps_mode == BNN_EPS_PHILOX) hipLaunchKernelGGL((bbb_fwd_gemm_kernel<4, false, BNN_EPS_PHILOX>), grid, block, 0, stream, k);
      else if (k.eps_mode == BNN_EPS_MEMORY) hipLaunchKernelGGL((bbb_fwd_gemm_kernel<4, false, BNN_EPS_MEMORY>), grid, block, 0, stream, k);
      else hipLaunchKernelGGL((bbb_fwd_gemm_kernel<4, false, BNN_EPS_ZERO>), grid, block, 0, stream, k);
    }
#undef BNN_GEMM_EPS
  } else if (ride) {
    const unsigned n_main = (unsigned)(((pl.blocks + 7) / 8) * 8);
    const dim3 grid_r(n_main + (unsigned)rider_blocks);
    const size_t lds_r = pl.lds > kSampleRedFloats * sizeof(float) ? pl.lds : kSampleRedFloats * sizeof(float);
#define BNN_RIDE(XDT, RR)                                                                                    \
  do {                                                                                                       \
    err = allow_big_lds(bbb_fwd_rider_kernel<XDT, RR>, lds_r);                                               \
    if (err == hipSuccess)                                                                                   \
      hipLaunchKernelGGL((bbb_fwd_rider_kernel<XDT, RR>), grid_r, block, lds_r, stream, k, sk, (int)n_main); \
  } while (0)
#define BNN_RIDE_R(XDT)                          \
  do {                                           \
    if (pl.R == 1) BNN_RIDE(XDT, 1);             \
    else if (pl.R == 2) BNN_RIDE(XDT, 2);        \
    else BNN_RIDE(XDT, 4);                       \
  } while (0)
    if (xdt == BNN_F32) BNN_RIDE_R(BNN_F32); else BNN_RIDE_R(BNN_BF16);
#undef BNN_RIDE
#undef BNN_RIDE_R
    if (err != hipSuccess) return (int)err;
  } else {
#define BNN_GO(MATH, XDT, RR, AL)                                                              \
  do {                                                                                         \
    err = allow_big_lds(bbb_fwd_kernel<MATH, XDT, RR, AL>, pl.lds);                            \
    if (err == hipSuccess)                                                                     \
      hipLaunchKernelGGL((bbb_fwd_kernel<MATH, XDT, RR, AL>), grid, block, pl.lds, stream, k); \
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
    } else if (math == BNN_MATH_BF16X3) {
      if (xdt == BNN_F32) BNN_GO_R(BNN_MATH_BF16X3, BNN_F32); else BNN_GO_R(BNN_MATH_BF16X3, BNN_BF16);
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

extern "C" int bnn_bbb_linear_fwd(const bnn_bbb_fwd_args* a, void* stream_) {
  return bbb_linear_fwd_impl(a, stream_, false, nullptr, 0);
}

static constexpr int kFinalMaxSlices = 8;
// above this many samples a one-block follow-up launch folds the sums: per-block fences cost more (re-measured this
// round at 20 / 32 / 64 pairs: 16 -> 32 loses 6 % / 4 % / 1 %)
static constexpr int kFinalTicketMaxSamples = 16;
static constexpr int kRowsTicketMaxSamples = 64;     // K1r: up to here the last sample's last block folds the sums (<= 4 round trips)

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
// reduce.hip: the training step's tail as its own launch (f->loss set, and the launch that finalized did not carry it)
extern "C" int bnn_loss_tail_(const bnn_finalize_args* f, void* stream_);
static int loss_tail(const bnn_finalize_args* f, void* stream_) { return bnn_loss_tail_(f, stream_); }

extern "C" int bnn_bbb_final_fwd(const bnn_bbb_fwd_args* a, const bnn_finalize_args* f, void* stream_) {
  BbbK k;
  bool al = false;
  int rc = prepare(a, k, al);
  if (rc != BNN_OK) return rc;
  FinPack fp;
  rc = make_fin(f, fp.k, fp.c);
  if (rc != BNN_OK) return rc;
  const int nl = f->n_layers;
  rc = check_fin_loss(f);
  if (rc != BNN_OK) return rc;
  if (a->w_sampled) {
    // ---- pre-sampled output layer: the row-split form (K1r) when the shapes allow it, else matmul-only K1 + K4
    const int S = a->n_samples, B = a->batch, N = a->out_features, K = a->in_features;
    // (S: a grid of S x (ceil(B / 16) + 1) small blocks; 64 until round 4 -- the launch group of 256 minibatches takes this form
    // behind its own sampling launch since: 28.6 + 4.1 us of one-block-per-pair K1c + the sums launch against ~17 us)
    const bool rows = a->y_dtype == BNN_F32 && N <= 16 && B <= 128 && S <= 4096 && !f->local_reparam &&
                      nl >= 1 && f->n_samples == S && f->classes == N && f->batch == B && f->logits == a->y && f->nll &&
                      f->scratch && f->scratch_bytes >= bnn_bbb_final_scratch_bytes(S) &&
                      !(reinterpret_cast<uintptr_t>(f->scratch) & 15) && (S == 1 || S > kRowsTicketMaxSamples || f->ticket) && !a->rider &&
                      (N % 4 != 0 || !(reinterpret_cast<uintptr_t>(a->y) & 15));
    if (!rows) {
      rc = bnn_bbb_linear_fwd(a, stream_);
      if (rc == BNN_OK) rc = bnn_elbo_finalize(f, stream_);
      return rc != BNN_OK ? rc : loss_tail(f, stream_);
    }
    const FinLoss tr = make_fin_loss(f);
    FinRows fr;
    fr.x = a->x;
    fr.x_sstride = k.x_sstride; fr.xg = k.xg;
    fr.w = k.w_pre; fr.b = k.b_pre;
    fr.y = reinterpret_cast<float*>(a->y);
    fr.S = S; fr.B = B; fr.K = K; fr.N = N; fr.relu = a->relu ? 1 : 0;
    char* base = reinterpret_cast<char*>(f->scratch);
    fr.tickets = reinterpret_cast<uint32_t*>(base);
    fr.parts = reinterpret_cast<float*>(base + (((size_t)S * 4 + 255) / 256) * 256);      // the K-slice statistics' region
    // (many samples: the sums of the per-sample scalars and the counter advance by a one-block follow-up kernel, as for K1c below --
    // one thread folding hundreds of samples behind the last block's ticket would be the longest thing in the launch)
    const bool rows_tail = S > kRowsTicketMaxSamples && !tr.out4;
    fp.sums = f->sums;
    fp.ticket = rows_tail ? nullptr : f->ticket;
    fp.ks = 1; fp.ks_ticket = nullptr; fp.ks_stats = nullptr; fp.ks_tiles = nullptr;
    const int RB = (B + 15) / 16;
    if (a->x_dtype == BNN_BF16)
      hipLaunchKernelGGL(bbb_final_rows_kernel<false>, dim3((unsigned)(S * (RB + 1))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                         fr, fp, tr);
    else
      hipLaunchKernelGGL(bbb_final_rows_kernel<true>, dim3((unsigned)(S * (RB + 1))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                         fr, fp, tr);
    const hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return (int)e2;
    return rows_tail ? bnn_elbo_sums_(f, stream_) : (int)BNN_OK;
  }
  const bool fuse = al && a->out_features <= 16 && a->batch <= 128 && a->want_stats && !f->local_reparam && nl >= 1 &&
                    f->layer_workspace[nl - 1] == a->workspace && f->n_samples == a->n_samples &&
                    f->classes == a->out_features && f->batch == a->batch && a->y_dtype == BNN_F32 &&
                    (a->n_samples == 1 || a->n_samples > kFinalTicketMaxSamples || f->ticket != nullptr) && !(a->log_prior || a->log_q) &&
                    a->form == BNN_FORM_AUTO;
  if (!fuse) {
    rc = bnn_bbb_linear_fwd(a, stream_);
    if (rc != BNN_OK) return rc;
    rc = bnn_elbo_finalize(f, stream_);
    return rc != BNN_OK ? rc : loss_tail(f, stream_);
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  fp.sums = f->sums;
  // Few samples: the last-arriving sample block folds the sums and advances the counter (one
  // release/acquire per block).  Many samples: those fences (an L2 write-back each) cost more
  // than a launch, so a one-block follow-up kernel does it instead.
  const bool tail_kernel = a->n_samples > kFinalTicketMaxSamples;
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
  } else if (a->math == BNN_MATH_BF16X3) {
    if (a->x_dtype == BNN_F32) BNN_FIN(BNN_MATH_BF16X3, BNN_F32); else BNN_FIN(BNN_MATH_BF16X3, BNN_BF16);
  } else {
    if (a->x_dtype == BNN_F32) BNN_FIN(BNN_MATH_F32, BNN_F32); else BNN_FIN(BNN_MATH_F32, BNN_BF16);
  }
#undef BNN_FIN
  if (err != hipSuccess) return (int)err;
  err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  rc = tail_kernel ? bnn_elbo_sums_(f, stream_) : (int)BNN_OK;
  return rc != BNN_OK ? rc : loss_tail(f, stream_);
}

// gx[S,B,K] = gz[S,B,N] . w_s with w regenerated (TRANS form of the K-split kernel).  Called by
// bnn_bbb_linear_bwd (bbb_bwd.hip).
extern "C" int bnn_bbb_input_grad_(const bnn_bbb_bwd_args* a, const float* gz, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (a->w_sampled_t) {
    // the forward left the sampled weights transposed ([S, in, out]): the input gradient is the forward's matmul-only
    // launch with the roles of the dimensions exchanged, gx[s] = gz[s] . wT[s]^T, no bias, masked by (x > 0)
    if (a->math != BNN_MATH_BF16) return BNN_ERR_ENUM;
    bnn_bbb_fwd_args fa{};
    fa.struct_bytes = sizeof(bnn_bbb_fwd_args);
    fa.n_samples = a->n_samples; fa.batch = a->batch;
    fa.in_features = a->out_features;             // reduction length
    fa.out_features = a->in_features;             // produced features
    const bool g16 = a->gy_bf16 && gz == a->gy;   // (a masked gz lives in the fp32 workspace)
    fa.x = g16 ? a->gy_bf16 : static_cast<const void*>(gz);
    fa.x_dtype = g16 ? BNN_BF16 : BNN_F32;
    fa.x_per_sample = 1;
    fa.math = BNN_MATH_BF16; fa.eps_mode = BNN_EPS_ZERO;
    fa.y = a->g_x; fa.y_dtype = BNN_F32;
    fa.y_bf16_copy = a->g_x_bf16;
    fa.w_sampled = a->w_sampled_t;
    fa.b_sampled = a->b_mu;                        // (non-NULL for the argument check; the launch takes no bias)
    fa.prior = a->prior;
    return bbb_linear_fwd_impl(&fa, stream_, true, a->gx_relu_mask ? a->x : nullptr,
                               a->x_per_sample ? (long)a->batch * a->in_features : 0);
  }
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
  k.xg = 1; k.sgrp = 0u; k.sgrp_stride = 0u;
  k.inv2var1 = k.inv2var2 = k.c1 = k.c2 = k.pi = 0.f;
#ifdef BNN_STAMPS
  k.dbg = nullptr;
#endif
  if ((a->in_features % 4 == 0) && (reinterpret_cast<uintptr_t>(a->g_x) & 15)) return BNN_ERR_ALIGN;
  const bool al = (k.K % 8 == 0) && aligned16(gz);
  BbbPlan pl{};
  tile_plan(k.S, k.B, k.K, k.N, al, 8, pl);
  const long total = pl.blocks;
  const dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(pl.nw * 64);
  const size_t lds = pl.lds;
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
    const long totalp = (long)((k.N + 15) / 16) * k.S * ((k.B + 127) / 128);
    const dim3 gridp((unsigned)(((totalp + 7) / 8) * 8));
    const size_t ldsp = pl.lds;
    if (!al) BNN_IGP(1, false, 8); else BNN_IGP(1, true, 8);
  } else if (a->math == BNN_MATH_BF16) BNN_IG_R(BNN_MATH_BF16); else BNN_IG_R(BNN_MATH_F32);
#undef BNN_IGP
#undef BNN_IG
#undef BNN_IG_R
  if (err != hipSuccess) return (int)err;
  err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}
