// K3  lr_linear_fwd — BayesianLinearLR.forward (reference networks.py:116-138) for all
// locally owned MC samples of one layer in one launch:
//
//   m = x . M,  v = x^2 . softplus(rho)^2,  y = m + sqrt(v)*eps_act + (b_mu + sigma_b*eps_b)
//
// Same decomposition as K1a (bbb_linear.hip): 1-D XCD-aware grid over (feature tile, sample,
// batch block); the 16 MFMA A rows carry F = 16/R features x R k-range classes so a single
// sample already covers the chip without cross-block reduction; the block's NW waves split
// the k-steps evenly; x fragments arrive as one batch of unconditional 16-byte loads at
// clamped addresses; (M, rho) of the next step are prefetched.  Differences from K1a:
//   * weights are stored [in, out] (networks.py:95-96): lane (r,q) GATHERS M[k0+8q+j][n],
//     j = 0..7 (for each j the 16 lanes of a quad row read 64 contiguous bytes; the other
//     half of each 128-byte line belongs to the neighbouring tile, which the XCD-aware order
//     places on the same L2);
//   * two accumulator sets (mean, variance) share every x fragment; x^2 is formed in fp32
//     from the fragment and re-rounded;
//   * nothing is sampled per weight: the closed-form KL sums (sum log sigma, sum sigma^2,
//     sum mu^2; networks.py:113) come from the same registers in the blocks of sample 0, so
//     (M, rho) are read once for both GEMMs and the KL;
//   * eps_act is generated in the epilogue in the D-fragment layout (4 consecutive features
//     of one batch row = one Philox call).
// The mean and variance slabs go through the same LDS region one after the other.
#include "bnn_device.h"
#include "bnn_fin.h"
#include "../../include/bnn_hip.h"
#include <string.h>
#include <type_traits>

namespace bnn {

struct LrK {
  const void* x;
  long x_sstride;
  const float* w_mu;
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  const float* eps_act;
  const float* eps_b;
  float* eps_act_dump;
  float* eps_b_dump;
  void* y;
  const void* x_sq; // optional bf16 x*x (same shape as x)
  const void* x_lo; // BNN_MATH_BF16X3: the low plane of x
  void* y_lo;       // BNN_MATH_BF16X3, bf16 y: the low plane of y
  const float4* w_frag;  // optional prepared weights (lr_prepare_kernel): [T][ksteps][2][64] x 16 B (X3: [3][64], see bnn_lr_prepare_x3)
  Xcd2D xc;              // K3b: work order (feature group x (sample, batch block)), see bnn_device.h
#ifdef BNN_TUNE
  int tune;              // tuning build only: 1 = no MFMAs, 2 = no LDS reads, 4 = no loads in the k loop
#endif
  void* y_sq;       // optional bf16 y*y
  __bf16* y16;      // optional bf16 copy of an fp32 y (K3a)
  float* hfac;      // optional eps_act / (2 sqrt(v)) (K3a): what the backward multiplies gz with
  float* v_out;     // optional fp32 variance
  float4* ws;       // float4 ws[1 + T]: header {T}, then {sum log sigma, sum sigma^2, sum mu^2, 0} per tile
  int S, B, K, N;
  int eps_mode, want_kl, relu, y_bf16;
  uint32_t k0, k1, layer_id, sample_offset;
  const uint32_t* sample_counter;
  int xg;                        // samples sharing one x (x index = s / xg)
  uint32_t sgrp, sgrp_stride;    // sample groups (bnn_lr_fwd_args.sample_group): 0 = none
  // K3s: another (narrow, <= 16 features) layer's operand preparation carried as extra blocks behind the main ones
  const float *rd_w_mu, *rd_w_rho, *rd_b_mu, *rd_b_rho;
  float4* rd_frag;               // [k-step][mean | variance][64] x 16 B
  float4* rd_ws;                 // header {blocks}, then one KL entry per rider block
  int rd_K, rd_N, rd_blocks, rd_spb, main_blocks;   // rider blocks, k-steps per rider block, first rider block index
  int ksl, nst;                  // K3s: K-range slices per unit, k-steps (of 32) per slice
  int es;                        // K3s: samples per unit (> 1: all samples share x -- the unit's products are made once and
                                 // its epilogue runs per sample; the units then count ONE sample)
  uint32_t* ks_ticket;           // K3s: [unit] arrival counters of a unit's slice blocks, zero between launches
  float4* ks_part;               // K3s: [unit][slice][wave 8][feature tile 2][mean | variance][64] x 16 B partial tiles
#ifdef BNN_STAMPS
  unsigned long long* dbg;   // diagnostic build only: [block][16] shader-clock stamps of wave 0
#endif
};

// global MC sample index (Philox subsequence) of local sample s
__device__ __forceinline__ uint32_t lr_global_sample(const LrK& p, int s) {
  const uint32_t base = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u);
  if (p.sgrp == 0u) return base + (uint32_t)s;
  const uint32_t g = (uint32_t)s / p.sgrp;
  return base + g * p.sgrp_stride + ((uint32_t)s - g * p.sgrp);
}

#ifdef BNN_STAMPS
#define LR_STAMP(i)                                                              \
  do {                                                                           \
    if (p.dbg && threadIdx.x == 0) p.dbg[((size_t)blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define LR_STAMP_RT(i)                                                           \
  do {                                                                           \
    if (p.dbg && threadIdx.x == 0) p.dbg[((size_t)blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define LR_STAMP(i)
#define LR_STAMP_RT(i)
#endif

// Epilogue of one output item (batch row `brow`, 4 consecutive features from `nb`) of sample s:
//   y = m + sqrt(v) * eps_act + b  [-> ReLU], with every optional by-product of bnn_lr_fwd_args.
__device__ __forceinline__ void lr_epilogue_item(const LrK& p, int s, uint32_t gs, int brow, int nb, const f32x4& vm4, const f32x4& vv4,
                                                 const float* bias4, const float* eps_pre = nullptr) {
  const int N = p.N, B = p.B;
  const bool vec_ok = (N & 3) == 0;
  const int gprN = (N + 3) >> 2;
  const size_t yoff = ((size_t)s * B + brow) * N + nb;
  float e4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.eps_mode == BNN_EPS_PHILOX) {
    if (eps_pre) {                                             // the caller drew them already (same call, earlier)
#pragma unroll
      for (int i = 0; i < 4; ++i) e4[i] = eps_pre[i];
    } else {
      philox_normal4((uint32_t)brow * (uint32_t)gprN + (uint32_t)(nb >> 2), gs, p.layer_id * 4u + 2u, p.k0, p.k1, e4);
    }
  } else if (p.eps_mode == BNN_EPS_MEMORY) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (nb + i < N) e4[i] = p.eps_act[yoff + i];
  }
  if (p.eps_act_dump) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (nb + i < N) p.eps_act_dump[yoff + i] = e4[i];
  }
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float o = __builtin_fmaf(__builtin_amdgcn_sqrtf(vv4[i]), e4[i], vm4[i]) + bias4[i];
    if (p.relu) o = fmaxf(o, 0.f);
    v[i] = o;
  }
  if (p.y_sq) {
    __bf16* qp = reinterpret_cast<__bf16*>(p.y_sq) + yoff;
    if (vec_ok) {
      bf16x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = sq_bf16(v[i]);
      *reinterpret_cast<bf16x4*>(qp) = o;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nb + i < N) qp[i] = sq_bf16(v[i]);
    }
  }
  if (p.v_out) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (nb + i < N) p.v_out[yoff + i] = vv4[i];
  }
  if (p.hfac) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (nb + i < N) {
        const float sd = __builtin_sqrtf(vv4[i]);
        p.hfac[yoff + i] = sd > 0.f ? e4[i] / (2.f * sd) : 0.f;
      }
  }
  if (p.y16) {
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

// MT = batch tiles (of 16 rows) per block: 8, or 2 for a narrow layer (the 10-class output layer is
// 3 feature tiles: with 128 rows per block three blocks would each ingest all of x; 32-row blocks
// make 12 of them, each with a quarter of x, 8 accumulators and room for 12 waves).
template <int MATH, int XDT, int R, int MT>
__device__ __forceinline__ void lr_fwd_body(const LrK& p) {
  constexpr int F = 16 / R, FG = F / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int c = r / F, f = r % F;
  const int K = p.K, N = p.N, B = p.B;
  const int ntiles = (N + F - 1) / F, mbs = (B + 16 * MT - 1) / (16 * MT);
  int item;
  if (!xcd_work_item(ntiles * p.S * mbs, item)) return;       // block-uniform
  const int nt = item / (p.S * mbs), s = (item / mbs) % p.S, mb = item % mbs;
  const int n = nt * F + f;
  const bool n_ok = n < N;
  const int nc = min(n, N - 1);
  const int m0 = mb * 16 * MT;
  const int mtiles = min(MT, (B - m0 + 15) >> 4);
  const int ssteps = (K + 32 * R - 1) / (32 * R);
  const uint32_t gs = lr_global_sample(p, s);
  const bool do_kl = p.want_kl && mb == 0 && s == 0;
  const bool x_al = (K & 7) == 0;                             // 16-byte x fragments possible
  const char* xs = reinterpret_cast<const char*>(p.x) + (size_t)(s / p.xg) * (size_t)p.x_sstride * (XDT == BNN_F32 ? 4 : 2);

  f32x4* slab = reinterpret_cast<f32x4*>(lds);
  float* lds_bias = lds + (size_t)nw * MT * 64 * 4;
  float* lds_red = lds_bias + 16;

  if (do_kl && item == 0 && threadIdx.x == 0) p.ws[0] = make_float4(__int_as_float(ntiles), 0.f, 0.f, 0.f);

  // bias of the tile: parameters and eps fetched now, applied in the epilogue
  float bmu_pre = 0.f, bsig_pre = 0.f, beps_pre = 0.f;
  if (wave == nw - 1 && lane < F && n_ok) {
    bmu_pre = p.b_mu[n];
    bsig_pre = softplus(p.b_rho[n]);
    if (p.eps_mode == BNN_EPS_PHILOX) {
      float e4[4];
      philox_normal4((uint32_t)(n >> 2), gs, p.layer_id * 4u + 1u, p.k0, p.k1, e4);
      beps_pre = (n & 3) == 0 ? e4[0] : (n & 3) == 1 ? e4[1] : (n & 3) == 2 ? e4[2] : e4[3];
    } else if (p.eps_mode == BNN_EPS_MEMORY) {
      beps_pre = p.eps_b[(size_t)s * N + n];
    }
    if (p.eps_b_dump && mb == 0) p.eps_b_dump[(size_t)s * N + n] = beps_pre;
  }

  LR_STAMP(0);
  LR_STAMP_RT(8);
  f32x4 am[MT], av[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    am[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    av[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float s_ls = 0.f, s_s2 = 0.f, s_m2 = 0.f;

  // gathered (M, rho) fragment of super-step t: rows k0+8q+j of column n, clamped; prefetched.
  float mu_n[8], rho_n[8];
  auto load_params = [&](int t) {
    const int k = (t * R + c) * 32 + q * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t off = (size_t)min(k + j, K - 1) * N + nc;
      mu_n[j] = p.w_mu[off];
      rho_n[j] = p.w_rho[off];
    }
  };
  if (wave < ssteps) load_params(wave);

#pragma nounroll
  for (int t = wave; t < ssteps; t += nw) {
    const int k = (t * R + c) * 32 + q * 8;
    constexpr int FR = (XDT == BNN_F32) ? 2 : 1;
    constexpr int MC0 = (8 / (R * FR)) < 1 ? 1 : (8 / (R * FR));
    constexpr int MC = MC0 > MT ? MT : MC0;
    float4 xraw[MC * R * FR];
    auto stage = [&](int ch) {
#pragma unroll
      for (int mm = 0; mm < MC; ++mm) {
        const int row = m0 + (ch * MC + mm) * 16 + r;
#pragma unroll
        for (int cc = 0; cc < R; ++cc) {
          const int xk = (t * R + cc) * 32 + q * 8;
          if (x_al) {
            const size_t off = (size_t)min(row, B - 1) * K + min(xk, K - 8);
            if (XDT == BNN_F32) {
              const float4* px = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xs) + off);
              xraw[(mm * R + cc) * 2 + 0] = px[0];
              xraw[(mm * R + cc) * 2 + 1] = px[1];
            } else {
              xraw[mm * R + cc] = *reinterpret_cast<const float4*>(reinterpret_cast<const __bf16*>(xs) + off);
            }
          } else {                                   // any K: guarded element loads
            const size_t ro = (size_t)min(row, B - 1) * K;
            if (XDT == BNN_F32) {
              float v[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = (xk + j < K) ? reinterpret_cast<const float*>(xs)[ro + xk + j] : 0.f;
              xraw[(mm * R + cc) * 2 + 0] = make_float4(v[0], v[1], v[2], v[3]);
              xraw[(mm * R + cc) * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
            } else {
              bf16x8 vb;
#pragma unroll
              for (int j = 0; j < 8; ++j)
                vb[j] = (xk + j < K) ? reinterpret_cast<const __bf16*>(xs)[ro + xk + j] : (__bf16)0.0f;
              xraw[mm * R + cc] = __builtin_bit_cast(float4, vb);
            }
          }
        }
      }
    };
    stage(0);

    float mu[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = mu_n[j];
      s2[j] = rho_n[j];
    }
    if (t + nw < ssteps) load_params(t + nw);
    if (t == wave) { asm volatile("" :: "v"(mu[0]), "v"(s2[0])); LR_STAMP(1); }

    float ls = 0.f, a2 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = n_ok && (k + j) < K;
      const float sig = softplus(s2[j]);
      if (do_kl) {
        ls += ok ? fast_log(sig) : 0.f;
        a2 += ok ? sig * sig : 0.f;
        m2 += ok ? mu[j] * mu[j] : 0.f;
      }
      mu[j] = ok ? mu[j] : 0.f;
      s2[j] = ok ? sig * sig : 0.f;
    }
    s_ls += ls;
    s_s2 += a2;
    s_m2 += m2;

    bf16x8 ma, sa, wz;
    if (MATH == BNN_MATH_BF16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ma[j] = (__bf16)mu[j];
        sa[j] = (__bf16)s2[j];
        wz[j] = (__bf16)0.f;
      }
    }
#pragma unroll
    for (int ch = 0; ch < MT / MC; ++ch) {
      if (ch * MC < mtiles) {
#pragma unroll
        for (int mm = 0; mm < MC; ++mm) {
#pragma unroll
          for (int cc = 0; cc < R; ++cc) {
            const int m = ch * MC + mm;
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
            if (MATH == BNN_MATH_BF16) {
              bf16x8 xb, x2b;
              if (XDT == BNN_BF16) {
                xb = __builtin_bit_cast(bf16x8, xraw[mm * R + cc]);      // already the MFMA operand
              } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xb[j] = (__bf16)xv[j];
              }
              // squares two at a time (v_pk_mul_f32): the x handling is what the vector pipe spends this kernel on
              typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
              for (int j = 0; j < 8; j += 2) {
                f32x2 pr = {xv[j], xv[j + 1]};
                pr = pr * pr;
                x2b[j] = (__bf16)pr[0];
                x2b[j + 1] = (__bf16)pr[1];
              }
              am[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((R == 1 || c == cc) ? ma : wz, xb, am[m], 0, 0, 0);
              av[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((R == 1 || c == cc) ? sa : wz, x2b, av[m], 0, 0, 0);
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                am[m] = __builtin_amdgcn_mfma_f32_16x16x4f32((R == 1 || c == cc) ? mu[j] : 0.f, xv[j], am[m], 0, 0, 0);
                av[m] = __builtin_amdgcn_mfma_f32_16x16x4f32((R == 1 || c == cc) ? s2[j] : 0.f, xv[j] * xv[j], av[m], 0, 0, 0);
              }
            }
          }
        }
        if ((ch + 1) * MC < MT && (ch + 1) * MC < mtiles) stage(ch + 1);
      }
    }
  }

  LR_STAMP(3);
  // ---- bias + its KL terms
  if (wave == nw - 1 && lane < 16) {
    float b = 0.f;
    if (lane < F && n_ok) {
      b = __builtin_fmaf(bsig_pre, beps_pre, bmu_pre);
      if (do_kl) {
        s_ls += fast_log(bsig_pre);
        s_s2 = __builtin_fmaf(bsig_pre, bsig_pre, s_s2);
        s_m2 = __builtin_fmaf(bmu_pre, bmu_pre, s_m2);
      }
    }
    lds_bias[lane] = b;
  }
  if (do_kl) {
    const float a = wave_sum(s_ls), b = wave_sum(s_s2), cc = wave_sum(s_m2);
    if (lane == 0) {
      lds_red[wave * 3 + 0] = a;
      lds_red[wave * 3 + 1] = b;
      lds_red[wave * 3 + 2] = cc;
    }
  }

  // ---- cross-wave reduction: mean slabs, then variance slabs, through the same LDS region.
  // Each thread owns at most NI output items (batch row x 4 consecutive features).
  constexpr int NI = 2;
  f32x4 vm[NI], vv[NI];
  auto reduce_items = [&](f32x4 (&out)[NI]) {
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) {
      const int it = threadIdx.x + ii * blockDim.x;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (it < mtiles * 16 * FG) {
        const int m = it / (16 * FG), rem = it - m * (16 * FG), fg = rem >> 4, b = rem & 15;
#pragma unroll 4
        for (int wv = 0; wv < nw; ++wv) {
#pragma unroll
          for (int cc = 0; cc < R; ++cc) v += slab[(wv * MT + m) * 64 + (cc * FG + fg) * 16 + b];
        }
      }
      out[ii] = v;
    }
  };
#pragma unroll
  for (int m = 0; m < MT; ++m)
    if (m < mtiles) slab[(wave * MT + m) * 64 + lane] = am[m];
  LR_STAMP(4);
  __syncthreads();
  LR_STAMP(5);
  reduce_items(vm);
  if (do_kl && threadIdx.x == 0) {
    float a = 0.f, b = 0.f, cc = 0.f;
    for (int wv = 0; wv < nw; ++wv) {
      a += lds_red[wv * 3 + 0];
      b += lds_red[wv * 3 + 1];
      cc += lds_red[wv * 3 + 2];
    }
    p.ws[1 + nt] = make_float4(a, b, cc, 0.f);
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MT; ++m)
    if (m < mtiles) slab[(wave * MT + m) * 64 + lane] = av[m];
  __syncthreads();
  reduce_items(vv);
  LR_STAMP(6);

  // ---- epilogue: y = m + sqrt(v) * eps + b  [-> ReLU]
#pragma unroll
  for (int ii = 0; ii < NI; ++ii) {
    const int it = threadIdx.x + ii * blockDim.x;
    if (it >= mtiles * 16 * FG) continue;
    const int m = it / (16 * FG), rem = it - m * (16 * FG), fg = rem >> 4, b = rem & 15;
    const int brow = m0 + m * 16 + b;
    const int nb = nt * F + fg * 4;
    if (brow >= B || nb >= N) continue;
    lr_epilogue_item(p, s, gs, brow, nb, vm[ii], vv[ii], lds_bias + fg * 4);
  }
  LR_STAMP(7);
  LR_STAMP_RT(9);
}

template <int MATH, int XDT, int R, int MT>
__global__ __launch_bounds__(MT == 2 ? 768 : 512) void lr_fwd_kernel(const LrK p) {
  lr_fwd_body<MATH, XDT, R, MT>(p);
}

// ------------------------------------------------------------------------------------------
// K3s  the one-evaluation / training form of a wide LR layer (1-2 samples, bf16 math, bf16 or fp32 x, K % 8 == 0, N % 4 == 0):
// K-sliced, whole parameter lines.
//   Why: K3a's block owns 8 features x all of K x 128 rows.  Its gathers use 32 bytes of every 128-byte (mu | rho) line
//   and every block ingests all of x: 614 KB through one CU's 64 B/clk L1 fill path = 9.6 k cycles before any latency,
//   12-15 us per 1200 x 1200 layer measured.  Here a block owns 32 FEATURES (one whole line per k row and tensor) x
//   128 rows x ONE K SLICE of nst <= 13 k-steps: 154 KB per block for the same layer in 152 blocks, no byte fetched
//   twice by a CU and none unused.
//   phase 1 (waves split the slice's k-steps): a step's (mu, rho) tile = 32 k rows x 128 B, four lane-linear 1 KiB
//     loads per tensor; sigma^2, the KL sums and the two bf16 operand tiles [k][feature] (64-byte rows, the halves of
//     rows 8..15 / 24..31 swapped so that the transposed reads below are conflict-free) into LDS;
//   phase 2 (waves split the batch rows, 16 each): A operands by ds_read_b64_tr_b16 (the [in, out] layout is k-major:
//     the hardware transposes a 4 x 16 block per 16-lane group), B operand = the wave's own x fragment of the step,
//     issued with every other load of the wave at kernel start, x^2 formed in registers: 4 MFMAs per step;
//   hand-off: each wave stores its four fp32 partial tiles write-through (sc1, lane-linear 1 KiB), one lane takes the
//     unit's ticket behind the block barrier, and the block that arrives LAST adds the slices up in slice order (its
//     own from registers, the others by sc1 loads) and runs the epilogue -- one launch, bitwise reproducible, nobody
//     waits (the protocol of K1b's K-sliced form, bbb_linear.hip).
// KL: one workspace entry per (feature group, slice) from the blocks of sample 0 / batch block 0.
// Rider of a K3s launch (bnn_lr_rider): block rb of rd_blocks prepares k-steps [rb * spb, (rb + 1) * spb) of a layer of
// <= 16 output features -- bf16 M and sigma^2 parked as [k][16] images in LDS (zero where k >= K or n >= N), written out in
// MFMA fragment order (lane (r, q) of step t: W[32 t + 8 q .. + 7][r]), and the block's share of the closed-form KL sums.
constexpr int kLrRiderSteps = 16;  // k-steps per rider block (2 x 16 KiB of LDS images)
__device__ __forceinline__ void lr_rider_block(const LrK& p, int rb, char* lds_img, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwv = nthr >> 6;
  const int K = p.rd_K, N = p.rd_N, ksteps = (K + 31) >> 5;
  const int t0 = rb * p.rd_spb, t1 = min(ksteps, t0 + p.rd_spb);
  __bf16* const m_s = reinterpret_cast<__bf16*>(lds_img);
  __bf16* const v_s = m_s + kLrRiderSteps * 32 * 16;
  for (int i = tid; i < kLrRiderSteps * 32 * 16 * 2 / 8; i += nthr) reinterpret_cast<float4*>(lds_img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int k0 = t0 * 32, k1 = min(K, t1 * 32);
  const int count = max(0, k1 - k0) * N;
  float ls = 0.f, s2 = 0.f, m2 = 0.f;
  for (int e0 = tid; e0 < count; e0 += 4 * nthr) {             // four elements per thread in flight
    float mu[4], rh[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + u * nthr, count - 1);
      mu[u] = p.rd_w_mu[(size_t)k0 * N + e];
      rh[u] = p.rd_w_rho[(size_t)k0 * N + e];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * nthr;
      if (e < count) {
        const int kk = e / N, n = e - kk * N;
        const float sig = softplus(rh[u]);
        ls += fast_log(sig);
        s2 = __builtin_fmaf(sig, sig, s2);
        m2 = __builtin_fmaf(mu[u], mu[u], m2);
        m_s[kk * 16 + n] = (__bf16)mu[u];
        v_s[kk * 16 + n] = (__bf16)(sig * sig);
      }
    }
  }
  if (rb == 0 && tid < N) {                                    // the biases' KL terms
    const float sig = softplus(p.rd_b_rho[tid]), mu = p.rd_b_mu[tid];
    ls += fast_log(sig);
    s2 = __builtin_fmaf(sig, sig, s2);
    m2 = __builtin_fmaf(mu, mu, m2);
  }
  {
    const float a = wave_sum(ls), b = wave_sum(s2), c = wave_sum(m2);
    if (lane == 0) {
      red[wave * 3 + 0] = a;
      red[wave * 3 + 1] = b;
      red[wave * 3 + 2] = c;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int w = 0; w < nwv; ++w) {
      a += red[w * 3 + 0];
      b += red[w * 3 + 1];
      c += red[w * 3 + 2];
    }
    p.rd_ws[1 + rb] = make_float4(a, b, c, 0.f);
    if (rb == 0) p.rd_ws[0] = make_float4(__int_as_float(p.rd_blocks), 0.f, 0.f, 0.f);
  }
  const int r = lane & 15, q = lane >> 4;
  for (int t = t0 + wave; t < t1; t += nwv) {
    const int tl = t - t0;
    bf16x8 ma, sa;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ma[j] = m_s[(tl * 32 + q * 8 + j) * 16 + r];
      sa[j] = v_s[(tl * 32 + q * 8 + j) * 16 + r];
    }
    p.rd_frag[(size_t)(t * 2 + 0) * 64 + lane] = __builtin_bit_cast(float4, ma);
    p.rd_frag[(size_t)(t * 2 + 1) * 64 + lane] = __builtin_bit_cast(float4, sa);
  }
}

constexpr int kLrsMaxSteps = 13;   // k-steps per slice (x fragments a wave keeps in registers)
constexpr int kLrsMaxSlices = 8;
constexpr int kLrsMaxShared = 64;  // samples that may share one unit's products (their biases wait in LDS).  With the epilogues spread
                                   // over the unit's slice blocks a further sample costs ~0.45 us (784 x 1200: 4 / 16 / 64 samples
                                   // 18 / 24 / 45 us against K3b's 20 / 25 / 67 + 8.5 us of prepare and cast launches:
                                   // tools/lr_shared_sweep.py, profiles/r03_lr_shared_sweep.log)

// XF32: the layer input is fp32 (the first layer of an evaluation: the minibatch as it arrives -- no cast launch ahead of it);
// a fragment is then two 16-byte loads, rounded to bf16 in registers where the cast kernel would have rounded it
// NX: k-steps a slice may have in this instantiation (4, 8, 10 or 13: the wave's x fragments and the product loop are
// unrolled to it -- a 7-step slice of the 784-wide layer issues 8 fragment loads, not 13)
template <bool XF32, int NX>
__global__ __launch_bounds__(512) void lr_fwd_kslice_kernel(const LrK p) {
  constexpr int NW = 8, XS = NX;
  constexpr int kTilesBytes = XS * 4096 > kLrRiderSteps * 2048 ? XS * 4096 : kLrRiderSteps * 2048;   // (a rider block's two images)
  __shared__ __attribute__((aligned(16))) char tiles[kTilesBytes];   // [step][mean | variance][32 k][64 B]
  __shared__ float lds_bias[kLrsMaxShared * 32];
  __shared__ float lds_red[3 * NW];
  __shared__ uint32_t last_s;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N, B = p.B, KSL = p.ksl;
  // All samples on ONE input (x_sstride == 0, e.g. the first layer of sample_elbo_lr / predict: networks.py:211-225 runs
  // forward(x) `samples` times on the same x): m = x M and v = x^2 sigma^2 do not depend on the sample, only the bias and
  // the activation noise do.  The units then count one sample, and whoever runs a unit's epilogue runs it ES times.
  const int ES = p.es, SU = ES > 1 ? 1 : p.S;
  // From four samples on the epilogues of a shared-input unit are SPREAD over its slice blocks (sample se runs in the block
  // of slice se % KSL) instead of all running in the block that arrives last: every block then waits (a bounded poll) until
  // all slices of the unit have arrived and adds them up itself, in slice order -- the same bits in every block.  Legal
  // because the plan launches these kernels as ONE round of blocks (every block is resident, nobody waits for a block that
  // has not started); costs each block the other slices' tiles and ~2 us of poll, saves (ES - ES / KSL) epilogues in a row.
  const bool spread = ES >= 4 && KSL > 1;                     // block-uniform
  const int G = (N + 31) >> 5, mbs = (B + 127) >> 7;
  if (p.rd_blocks > 0 && (int)blockIdx.x >= p.main_blocks) {   // rider blocks sit behind the (padded) main grid
    lr_rider_block(p, (int)blockIdx.x - p.main_blocks, tiles, lds_red);
    return;
  }
  int item;
  if (!xcd_work_item(G * SU * mbs * KSL, item)) return;        // block-uniform
  const int ks = item % KSL, unit = item / KSL;
  const int g = unit / (SU * mbs), s = (unit / mbs) % SU, mb = unit % mbs;
  const int n0 = g * 32, m0 = mb * 128;
  const int mtiles = min(8, (B - m0 + 15) >> 4);
  const int ksteps = (K + 31) >> 5;
  const int t_begin = ks * p.nst, nst = max(0, min(p.nst, ksteps - t_begin));   // this slice's steps (may be none)
  const uint32_t gs = lr_global_sample(p, s);
  const bool do_kl = p.want_kl && mb == 0 && s == 0;
  const char* xs = reinterpret_cast<const char*>(p.x) + (size_t)(s / p.xg) * (size_t)p.x_sstride * (XF32 ? 4 : 2);

  if (do_kl && item == 0 && threadIdx.x == 0) p.ws[0] = make_float4(__int_as_float(G * KSL), 0.f, 0.f, 0.f);

  LR_STAMP(0);
  LR_STAMP_RT(8);
  // ---- every load of the wave, issued before anything is waited for (branch-free: clamped steps re-read a line)
  // phase-1 share of this wave: the slice's 8-row groups wave, wave + 8, ... (4 per k-step: a lane-linear 1 KiB load
  // of 8 k rows x 32 features per tensor)
  constexpr int PG = (4 * XS + NW - 1) / NW;                   // groups per wave at most
  const int c8 = lane & 7, rr = lane >> 3;                     // 16-byte chunk (4 features) and row within the group
  const int nf = n0 + c8 * 4;                                  // first of the lane's 4 features
  const bool nf_ok = nf < N;                                   // N % 4 == 0: a chunk is in or out as a whole
  const int nfc = min(nf, N - 4);
  const int ngr = nst * 4;
  float4 pm[PG], pr[PG];
#pragma unroll
  for (int u = 0; u < PG; ++u) {
    const int rg = min(wave + u * NW, max(ngr - 1, 0));
    const size_t off = (size_t)min(t_begin * 32 + rg * 8 + rr, K - 1) * N + nfc;
    pm[u] = *reinterpret_cast<const float4*>(p.w_mu + off);
    pr[u] = *reinterpret_cast<const float4*>(p.w_rho + off);
  }
  __builtin_amdgcn_sched_barrier(0);                           // the parameters are wanted first: keep their loads ahead of x's
  // phase-2 x fragments: rows 16 wave + r, k = 32 (t_begin + j) + 8 q
  const char* xrow = xs + (size_t)min(m0 + wave * 16 + r, B - 1) * K * (XF32 ? 4 : 2);
  float4 xf[XS][XF32 ? 2 : 1];
#pragma unroll
  for (int j = 0; j < XS; ++j) {
    const int kx = min((t_begin + min(j, max(nst - 1, 0))) * 32 + q * 8, K - 8);
    if (XF32) {
      const float4* px = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xrow) + kx);
      xf[j][0] = px[0];
      xf[j][XF32 ? 1 : 0] = px[1];
    } else {
      xf[j][0] = *reinterpret_cast<const float4*>(reinterpret_cast<const __bf16*>(xrow) + kx);
    }
  }
  // bias of the group (used by the last wave's lanes 0..31): loaded by every lane at clamped addresses, BEHIND the parameter and x
  // loads and without a branch -- two loads issued under `if (bias_lane)` ahead of the others made the compiler wait for them
  // before it issued the second half of the parameter loads and all of x (a whole round trip in wave 7, which the block
  // then waited for at the barrier)
  const bool bias_lane = wave == NW - 1 && lane < 32 && n0 + lane < N;
  float bmu_pre = p.b_mu[min(n0 + (lane & 31), N - 1)], brho_pre = p.b_rho[min(n0 + (lane & 31), N - 1)];
  // (the injected epsilon stays in its own register until the select below: sharing one variable with the drawn value let the
  // generator's temporaries reuse the load's destination, and the compiler then waited for EVERY load before the draw)
  float beps_mem = 0.f, beps_phx = 0.f;
  if (p.eps_mode == BNN_EPS_MEMORY) beps_mem = p.eps_b[(size_t)s * N + min(n0 + (lane & 31), N - 1)];
  if (bias_lane && p.eps_mode == BNN_EPS_PHILOX) {
    const int n = n0 + lane;
    float e4[4];
    philox_normal4((uint32_t)(n >> 2), gs, p.layer_id * 4u + 1u, p.k0, p.k1, e4);
    beps_phx = (n & 3) == 0 ? e4[0] : (n & 3) == 1 ? e4[1] : (n & 3) == 2 ? e4[2] : e4[3];
  }
#pragma unroll
  for (int u = 0; u < PG; ++u) {                               // the arithmetic starts HERE: issued loads stay ahead of it
    asm volatile("" : "+v"(pm[u].x), "+v"(pm[u].y), "+v"(pm[u].z), "+v"(pm[u].w));
    asm volatile("" : "+v"(pr[u].x), "+v"(pr[u].y), "+v"(pr[u].z), "+v"(pr[u].w));
  }

  LR_STAMP(1);
  // ---- phase 1: sigma^2, KL sums, bf16 operand tiles into LDS
  float s_ls = 0.f, s_s2 = 0.f, s_m2 = 0.f;
  // FULL: every k row of the slice and every feature of the group is real (block-uniform) -- no masks in the arithmetic
  auto park = [&](auto full_, auto kl_) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_)::value, KL = decltype(kl_)::value;
#pragma unroll
    for (int u = 0; u < PG; ++u) {
      const int rg = wave + u * NW;                            // 8-row group of the slice
      if (rg < ngr) {
        const int kk = (rg & 3) * 8 + rr;                      // k row within its step
        const bool ok = FULL || (nf_ok && (t_begin * 32 + rg * 8 + rr) < K);
        const float mu4[4] = {pm[u].x, pm[u].y, pm[u].z, pm[u].w};
        const float rh4[4] = {pr[u].x, pr[u].y, pr[u].z, pr[u].w};
        bf16x4 mb4, sb4;
        float ls = 0.f, a2 = 0.f, m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sig = softplus(rh4[e]);
          const float sg2 = sig * sig;
          if (KL) {
            ls += ok ? fast_log(sig) : 0.f;
            a2 += ok ? sg2 : 0.f;
            m2 += ok ? mu4[e] * mu4[e] : 0.f;
          }
          mb4[e] = (__bf16)(ok ? mu4[e] : 0.f);
          sb4[e] = (__bf16)(ok ? sg2 : 0.f);
        }
        s_ls += ls;
        s_s2 += a2;
        s_m2 += m2;
        const int off = (rg >> 2) * 4096 + kk * 64 + ((c8 * 8) ^ (((kk >> 3) & 1) * 32));
        *reinterpret_cast<bf16x4*>(tiles + off) = mb4;
        *reinterpret_cast<bf16x4*>(tiles + off + 2048) = sb4;
      }
    }
  };
  const bool full = n0 + 32 <= N && (t_begin + nst) * 32 <= K;
  if (full) {
    if (do_kl) park(std::true_type{}, std::true_type{}); else park(std::true_type{}, std::false_type{});
  } else {
    if (do_kl) park(std::false_type{}, std::true_type{}); else park(std::false_type{}, std::false_type{});
  }
  LR_STAMP(2);
  if (wave == NW - 1 && lane < 32) {
    float b = 0.f;
    if (bias_lane) {
      const int n = n0 + lane;
      const float bsig = softplus(brho_pre);
      const float beps_pre = p.eps_mode == BNN_EPS_MEMORY ? beps_mem : beps_phx;
      if (p.eps_b_dump && mb == 0 && ks == 0) p.eps_b_dump[(size_t)s * N + n] = beps_pre;
      b = __builtin_fmaf(bsig, beps_pre, bmu_pre);
      if (do_kl && ks == 0) {
        s_ls += fast_log(bsig);
        s_s2 = __builtin_fmaf(bsig, bsig, s_s2);
        s_m2 = __builtin_fmaf(bmu_pre, bmu_pre, s_m2);
      }
      for (int se = 1; se < ES; ++se) {                        // the other samples' biases (sample 0's is `b`)
        if (spread && se % KSL != ks) continue;                // (spread: the samples whose epilogue runs in this block)
        float be = 0.f;
        if (p.eps_mode == BNN_EPS_PHILOX) {
          float e4[4];
          philox_normal4((uint32_t)(n >> 2), gs + (uint32_t)se, p.layer_id * 4u + 1u, p.k0, p.k1, e4);   // (no sample groups here)
          be = (n & 3) == 0 ? e4[0] : (n & 3) == 1 ? e4[1] : (n & 3) == 2 ? e4[2] : e4[3];
        } else if (p.eps_mode == BNN_EPS_MEMORY) {
          be = p.eps_b[(size_t)se * N + n];
        }
        if (p.eps_b_dump && mb == 0 && (spread || ks == 0)) p.eps_b_dump[(size_t)se * N + n] = be;   // (spread: the one slice block that draws it)
        lds_bias[se * 32 + lane] = __builtin_fmaf(bsig, be, bmu_pre);
      }
    } else {
      for (int se = 1; se < ES; ++se) lds_bias[se * 32 + lane] = 0.f;
    }
    lds_bias[lane] = b;
  }
  if (do_kl) {
    const float a = wave_sum(s_ls), b = wave_sum(s_s2), cc = wave_sum(s_m2);
    if (lane == 0) {
      lds_red[wave * 3 + 0] = a;
      lds_red[wave * 3 + 1] = b;
      lds_red[wave * 3 + 2] = cc;
    }
  }
  __syncthreads();
  LR_STAMP(3);
  if (do_kl && threadIdx.x == 0) {
    float a = 0.f, b = 0.f, cc = 0.f;
    for (int wv = 0; wv < NW; ++wv) {
      a += lds_red[wv * 3 + 0];
      b += lds_red[wv * 3 + 1];
      cc += lds_red[wv * 3 + 2];
    }
    p.ws[1 + g * KSL + ks] = make_float4(a, b, cc, 0.f);
  }

  // ---- phase 2: rows 16 wave .. + 15 against the slice's operand tiles (every wave: the transposed reads want EXEC
  // all ones, and a wave past the batch multiplies clamped rows it never stores)
  f32x4 acc[2][2];                                            // [feature tile][mean | variance]
#pragma unroll
  for (int ft = 0; ft < 2; ++ft) acc[ft][0] = acc[ft][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4;
  // lane 4 q' + p' of a 16-lane group kg supplies row 8 kg + (4 h) + q', columns 4 p' .. 4 p' + 3 of the feature tile
  const int tr_off = (q * 8 + ((lane & 15) >> 2)) * 64;
  const int tr_col = (lane & 3) * 8, tr_sw = (q & 1) * 32;
  auto operands = [&](int j, bf16x8 (&wa)[2][2]) __attribute__((always_inline)) {   // [feature tile][mean | variance] of step j
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      const int a0 = j * 4096 + tr_off + ((ft * 32 + tr_col) ^ tr_sw);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(tiles + a0 + st * 2048));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(tiles + a0 + st * 2048 + 256));
        wa[ft][st] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
  };
  // NS steps fully unrolled and branch-free (a step past the slice multiplies a zeroed x fragment with the last step's
  // tiles), the next step's operands read before this step's products
  bf16x8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = (__bf16)0.f;
  auto products = [&](auto ns_) __attribute__((always_inline)) {
    constexpr int NS = decltype(ns_)::value;
    bf16x8 wa[2][2][2];                                        // [parity][feature tile][mean | variance]
    operands(0, wa[0]);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (j + 1 < NS) operands(min(j + 1, max(nst - 1, 0)), wa[(j + 1) & 1]);
      bf16x8 xb;
      if (XF32) {
        const float xv[8] = {xf[j][0].x, xf[j][0].y, xf[j][0].z, xf[j][0].w, xf[j][XF32 ? 1 : 0].x, xf[j][XF32 ? 1 : 0].y,
                             xf[j][XF32 ? 1 : 0].z, xf[j][XF32 ? 1 : 0].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) xb[e] = (__bf16)xv[e];
      } else {
        xb = __builtin_bit_cast(bf16x8, xf[j][0]);
      }
      if (j >= nst) xb = zero8;
      bf16x8 x2b;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        f32x2 pr2 = {(float)xb[e], (float)xb[e + 1]};
        pr2 = pr2 * pr2;
        x2b[e] = (__bf16)pr2[0];
        x2b[e + 1] = (__bf16)pr2[1];
      }
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        acc[ft][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j & 1][ft][0], xb, acc[ft][0], 0, 0, 0);
        acc[ft][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j & 1][ft][1], x2b, acc[ft][1], 0, 0, 0);
      }
    }
  };
  if (nst > 0) products(std::integral_constant<int, NX>{});

  LR_STAMP(4);
  // activation noise of the lane's two output items (drawn by whoever runs the epilogue; the last arriver draws it while
  // the other slices' tiles are in flight)
  const int brow = m0 + wave * 16 + r;
  const int gprN = (N + 3) >> 2;
  float eps_pre[2][4];
  auto draw_eps = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
      philox_normal4((uint32_t)brow * (uint32_t)gprN + (uint32_t)((n0 + ft * 16 + q * 4) >> 2), gs, p.layer_id * 4u + 2u, p.k0, p.k1, eps_pre[ft]);
  };
  // ---- hand-off between the slices of the unit
  if (KSL > 1) {
    float4* const unit_part = p.ks_part + (size_t)unit * KSL * (NW * 4 * 64);
    float4* const mine = unit_part + ((size_t)ks * NW + wave) * (4 * 64) + lane;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int st = 0; st < 2; ++st)
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine + (ft * 2 + st) * 64), "v"(acc[ft][st]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tk = __hip_atomic_fetch_add(p.ks_ticket + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_s = ((tk & 0xffffu) == (uint32_t)KSL - 1u) ? 1u : 0u;
      if (spread) {
        // arrivals in the low half of the word, departures in the high half; the poll is bounded: a launch whose unit's
        // blocks are not all resident (CUs held by another stream's kernels) must not hang the device -- it ends, and the
        // block that gave up POISONS its share of the outputs (NaN: the evaluation's NLL and ELBO show it) instead of
        // building them from incomplete partial tiles
        uint32_t seen = tk + 1u;
        for (int it = 0; (seen & 0xffffu) < (uint32_t)KSL && it < (1 << 22); ++it) {
          __builtin_amdgcn_s_sleep(4);
          seen = __hip_atomic_load(p.ks_ticket + unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t dp = __hip_atomic_fetch_add(p.ks_ticket + unit, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((dp >> 16) == (uint32_t)KSL - 1u)                  // the last to leave puts the word back to zero
          __hip_atomic_store(p.ks_ticket + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = ((seen & 0xffffu) < (uint32_t)KSL) ? 2u : 1u;   // every block of the unit runs its share of the epilogues
      }                                                        // (2: the poll timed out)
    }
    __syncthreads();
    LR_STAMP(5);
    LR_STAMP_RT(9);
    if (last_s == 0u) return;                                  // block-uniform
    if (!spread && threadIdx.x == 0) __hip_atomic_store(p.ks_ticket + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x4 part[kLrsMaxSlices][4];
#pragma unroll
    for (int sl = 0; sl < kLrsMaxSlices; ++sl) {               // every other slice's tiles, one round of loads
      if (sl < KSL && sl != ks) {                              // block-uniform
        const float4* src = unit_part + ((size_t)sl * NW + wave) * (4 * 64) + lane;
#pragma unroll
        for (int m = 0; m < 4; ++m) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[sl][m]) : "v"(src + m * 64) : "memory");
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) part[sl][m] = acc[m >> 1][m & 1];
      }
    }
    if (p.eps_mode == BNN_EPS_PHILOX && !spread) draw_eps();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 sum[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) sum[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < kLrsMaxSlices; ++sl)                 // slice order, whoever arrived last
      if (sl < KSL) {
#pragma unroll
        for (int m = 0; m < 4; ++m) sum[m] += part[sl][m];
      }
    if (last_s == 2u) {                                        // block-uniform: a slice never arrived
      const float nan = __builtin_nanf("");
#pragma unroll
      for (int m = 0; m < 4; ++m) sum[m] = f32x4{nan, nan, nan, nan};
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m >> 1][m & 1] = sum[m];
  } else if (p.eps_mode == BNN_EPS_PHILOX) {
    draw_eps();
  }
  LR_STAMP(6);
  // ---- epilogue: lane (r, q) of feature tile ft holds batch row m0 + 16 wave + r, features n0 + 16 ft + 4 q ..
  if (wave < mtiles && brow < B) {
    if (!spread) {
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const int nb = n0 + ft * 16 + q * 4;
        if (nb < N) lr_epilogue_item(p, s, gs, brow, nb, acc[ft][0], acc[ft][1], lds_bias + ft * 16 + q * 4, eps_pre[ft]);
      }
    }
#pragma nounroll
    for (int se = spread ? ks : 1; se < ES; se += spread ? KSL : 1) {   // the other samples of a shared-input unit (spread: this block's share)
      const uint32_t gse = gs + (uint32_t)se;                  // (shared input: no sample groups; lr_global_sample would re-read the counter)
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const int nb = n0 + ft * 16 + q * 4;
        if (nb < N) lr_epilogue_item(p, se, gse, brow, nb, acc[ft][0], acc[ft][1], lds_bias + se * 32 + ft * 16 + q * 4);
      }
    }
  }
  LR_STAMP(7);
  LR_STAMP_RT(9);
}

// ------------------------------------------------------------------------------------------
// K3b  throughput form (many MC samples, bf16 math, bf16 x AND x^2 provided): block GEMM.
//   1-D XCD-aware grid over (feature-tile group, sample, batch block), block = NW waves; wave j
//   owns 16 features for ALL of K.  Per k-step the 128 x 32 tiles of x and x^2 are brought once
//   per block into double-buffered LDS by LDS-DMA (global_load_lds_dwordx4: one wave-instruction
//   = one batch tile's lane-linear 1 KiB fragment block), each wave gathers its (M, rho)
//   fragment (prefetched a step ahead), forms sigma^2 and issues 16 MFMAs (mean and variance
//   against the two tiles).  No per-weight sampling: ~150 VALU ops per k-step, one barrier.
// Prepared weights for K3b: the eps-independent half of the LR layer, done ONCE per evaluation instead of once per MC
// sample: sigma^2, bf16 (M, sigma^2) in FRAGMENT ORDER ([tile][k-step][mean | variance][lane] x 16 B), so that the GEMM
// kernel's operand fetch is one coalesced 16-byte load per fragment, and the closed-form KL sums (networks.py:113).
// (The first version gathered its fragments straight from the [in,out] matrices -- 4-byte loads, 64-byte segments --
// in one block per feature tile: 23 us for the 1200 x 1200 layer, 46 us of every LR evaluation preparing two layers.)
// A block owns `kb` k-steps x 128 out-features: it reads (mu, rho) rows as
// 512-byte segments (16 bytes per thread), forms bf16 M and sigma^2 and the KL terms, parks the two 32 x 128 bf16
// tiles in LDS and writes them out in MFMA-fragment order, 16 bytes per lane, 1 KiB per wave.  One KL entry per block
// (the finalize only needs the layer's totals); the biases go with the blocks of the first k range.
// X3 (split-bf16 math): a third plane per (tile, k-step) -- the low part bf16(M - bf16(M)) of the mean operand; the fragment
// block is then [mean hi | variance | mean lo][64 lanes] x 16 B.
// (kblk, grp) of (nkb, ngrp): the block's k-step range and 128-feature group -- the grid of a one-layer launch, or the block's place
// inside its layer's share of a several-layer launch (bnn_lr_prepare_many)
template <bool X3>
__device__ __forceinline__ void lr_prepare_block(const float* __restrict__ w_mu, const float* __restrict__ w_rho,
                                                 const float* __restrict__ b_mu, const float* __restrict__ b_rho,
                                                 int K, int N, int kb, float4* __restrict__ frag, float4* __restrict__ ws,
                                                 int kblk, int grp, int nkb, int ngrp) {
  constexpr int LD = 130;                                   // bf16 elements per LDS row (128 + 2: the four 8-row groups a
  constexpr int FB = X3 ? 192 : 128;                        // float4s per fragment block
  __shared__ __bf16 m_s[32 * LD], v_s[32 * LD];             // fragment read touches fall on different banks)
  __shared__ __bf16 l_s[X3 ? 32 * LD : 1];
  __shared__ float red[4 * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ksteps = (K + 31) >> 5, T = (N + 15) >> 4;
  const int n0 = grp * 128;
  const bool vec = (N & 3) == 0 && !((reinterpret_cast<uintptr_t>(w_mu) | reinterpret_cast<uintptr_t>(w_rho)) & 15);
  float s_ls = 0.f, s_s2 = 0.f, s_m2 = 0.f;
  for (int t = kblk * kb; t < min(ksteps, (kblk + 1) * kb); ++t) {
    // ---- 32 rows x 128 columns: thread -> (row = i * 8 + tid / 32, 4 columns at (tid % 32) * 4), i = 0..3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = i * 8 + ((int)threadIdx.x >> 5), k = t * 32 + kr;
      const int c4 = ((int)threadIdx.x & 31) * 4, n = n0 + c4;
      float mu[4] = {0.f, 0.f, 0.f, 0.f}, rh[4] = {0.f, 0.f, 0.f, 0.f};
      bool ok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ok[j] = k < K && n + j < N;
      if (k < K && n < N) {
        const size_t off = (size_t)k * N + n;
        if (vec) {                                           // n + 3 < N as N % 4 == 0
          const float4 a = *reinterpret_cast<const float4*>(w_mu + off), b = *reinterpret_cast<const float4*>(w_rho + off);
          mu[0] = a.x; mu[1] = a.y; mu[2] = a.z; mu[3] = a.w;
          rh[0] = b.x; rh[1] = b.y; rh[2] = b.z; rh[3] = b.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (ok[j]) {
              mu[j] = w_mu[off + j];
              rh[j] = w_rho[off + j];
            }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sig = softplus(rh[j]);
        if (ws) {
          s_ls += ok[j] ? fast_log(sig) : 0.f;
          s_s2 += ok[j] ? sig * sig : 0.f;
          s_m2 += ok[j] ? mu[j] * mu[j] : 0.f;
        }
        const __bf16 mh = ok[j] ? (__bf16)mu[j] : (__bf16)0.f;
        m_s[kr * LD + c4 + j] = mh;
        v_s[kr * LD + c4 + j] = ok[j] ? (__bf16)(sig * sig) : (__bf16)0.f;
        if (X3) l_s[kr * LD + c4 + j] = ok[j] ? split_lo(mu[j], mh) : (__bf16)0.f;
      }
    }
    __syncthreads();
    // ---- fragments: wave w writes tiles w and w + 4 of the group; lane (r, q) holds W[t*32 + 8q .. +7][tile*16 + r]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tl = wave + 4 * h, tile = grp * 8 + tl;
      if (tile < T) {                                        // wave-uniform
        bf16x8 ma, sa, la;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ma[j] = m_s[(q * 8 + j) * LD + tl * 16 + r];
          sa[j] = v_s[(q * 8 + j) * LD + tl * 16 + r];
          if (X3) la[j] = l_s[(q * 8 + j) * LD + tl * 16 + r];
        }
        float4* dst = frag + ((size_t)tile * ksteps + t) * FB;
        dst[lane] = __builtin_bit_cast(float4, ma);
        dst[64 + lane] = __builtin_bit_cast(float4, sa);
        if (X3) dst[128 + lane] = __builtin_bit_cast(float4, la);
      }
    }
    __syncthreads();
  }
  if (ws) {
    if (kblk == 0 && (int)threadIdx.x < 128 && n0 + (int)threadIdx.x < N) {     // the group's biases
      const float sig = softplus(b_rho[n0 + threadIdx.x]), mu = b_mu[n0 + threadIdx.x];
      s_ls += fast_log(sig);
      s_s2 = __builtin_fmaf(sig, sig, s_s2);
      s_m2 = __builtin_fmaf(mu, mu, s_m2);
    }
    const float a = wave_sum(s_ls), b = wave_sum(s_s2), c = wave_sum(s_m2);
    if (lane == 0) {
      red[wave * 3 + 0] = a;
      red[wave * 3 + 1] = b;
      red[wave * 3 + 2] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float x = 0.f, y = 0.f, z = 0.f;
      for (int w = 0; w < 4; ++w) {
        x += red[w * 3 + 0];
        y += red[w * 3 + 1];
        z += red[w * 3 + 2];
      }
      const int entry = grp * nkb + kblk;
      ws[1 + entry] = make_float4(x, y, z, 0.f);
      if (entry == 0) ws[0] = make_float4(__int_as_float(nkb * ngrp), 0.f, 0.f, 0.f);
    }
  }
}

template <bool X3>
__global__ __launch_bounds__(256) void lr_prepare_tiled_kernel(const float* __restrict__ w_mu, const float* __restrict__ w_rho,
                                                               const float* __restrict__ b_mu, const float* __restrict__ b_rho,
                                                               int K, int N, int kb, float4* __restrict__ frag,
                                                               float4* __restrict__ ws) {
  lr_prepare_block<X3>(w_mu, w_rho, b_mu, b_rho, K, N, kb, frag, ws, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y);
}

// The prepared operands of SEVERAL layers in one launch (bnn_lr_prepare_many: an evaluation's prepare launches depend on no
// activation and were one launch per layer, ~4 us of launch boundary each ahead of the first layer): 1-D grid, job j owns the
// blocks [first[j], first[j + 1]).
constexpr int kPrepManyJobs = 8;
struct LrPrepJobs {
  const float* w_mu[kPrepManyJobs];
  const float* w_rho[kPrepManyJobs];
  const float* b_mu[kPrepManyJobs];
  const float* b_rho[kPrepManyJobs];
  float4* frag[kPrepManyJobs];
  float4* ws[kPrepManyJobs];
  int K[kPrepManyJobs], N[kPrepManyJobs], kb[kPrepManyJobs], nkb[kPrepManyJobs], ngrp[kPrepManyJobs];
  int first[kPrepManyJobs + 1];
  int n;
};
template <bool X3>
__global__ __launch_bounds__(256) void lr_prepare_many_kernel(const LrPrepJobs jb) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < kPrepManyJobs; ++i)
    if (i < jb.n && (int)blockIdx.x >= jb.first[i]) j = i;   // block-uniform
  const int local = (int)blockIdx.x - jb.first[j];
  const int nkb = jb.nkb[j];
  lr_prepare_block<X3>(jb.w_mu[j], jb.w_rho[j], jb.b_mu[j], jb.b_rho[j], jb.K[j], jb.N[j], jb.kb[j], jb.frag[j], jb.ws[j],
                       local % nkb, local / nkb, nkb, jb.ngrp[j]);
}

// DEPTH = k-steps of prefetch (PREP only takes 2).  A step's loads -- x / x^2 tiles by LDS-DMA, prepared fragments to
// registers -- are issued DEPTH steps ahead and waited for one step ahead (s_waitcnt vmcnt(ops of the younger step)),
// in a ring of DEPTH + 1 LDS buffers.  With one step of prefetch every block-step lasted a whole L2 round trip on top
// of its own work: with the fabric traffic out of the way (2-D work order) the counters showed MFMA, LDS, L2 and VALU
// each ~26 % busy and the waves parked two thirds of the time.
// X3 (BNN_MATH_BF16X3, prepared fragments only): the MEAN product in split-bf16 -- M_hi x_hi + M_lo x_hi + M_hi x_lo over the
// plane pair (x, x_lo) and the fragment pair (M_hi, M_lo): the reference's fp32 x . M (networks.py:120) to ~1e-5 of the output
// scale --, the VARIANCE product as in bf16 math (one MFMA over bf16(x^2), bf16(sigma^2): the activation noise sqrt(v) eps is a
// few per cent of the output, a 2^-9 relative error of it is below the mean product's 2^-15; measured on the CPU: NLL 1e-6
// against the fp32 arithmetic either way, tests/test_oracle_golden.py).  Four MFMAs per batch tile and k-step instead of two,
// three tile planes (24 KiB per step) and three fragments per wave-step.
template <int NW, bool PREP, int DEPTH, bool X3 = false>
__global__ __launch_bounds__(NW * 64, PREP ? 4 : 3) void lr_fwd_gemm_kernel(const LrK p) {
  static_assert(DEPTH == 1 || (PREP && DEPTH == 2), "two steps of prefetch: prepared fragments only");
  static_assert(!X3 || PREP, "split-bf16 form: over prepared fragments");
  constexpr int NP = X3 ? 3 : 2;
  typedef float4 XtB[NP][8 * 64];                                   // [x | x^2 (| x lo)][tile]: 16 (24) KiB
  __shared__ __attribute__((aligned(16))) XtB xt[DEPTH + 1];        // ring: step t reads xt[t % (DEPTH + 1)]
  __shared__ float bias_s[NW][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N, B = p.B;
  const int tbs = (N + 16 * NW - 1) / (16 * NW), mbs = (B + 127) >> 7;
  int tb, unit, in_;
  if (!xcd2d_work_item(p.xc, tb, unit, in_)) return;          // block-uniform
  LR_STAMP(0);
  LR_STAMP_RT(8);
  const int s = unit / mbs, mb = unit - s * mbs;
  const int item = tb * (p.S * mbs) + unit;
  const int tile = tb * NW + wave;
  const int n = tile * 16 + r;
  const bool n_ok = n < N;
  const int nc = min(n, N - 1);
  const int m0 = mb * 128;
  const int ksteps = (K + 31) >> 5;
  const uint32_t gs = lr_global_sample(p, s);
  const bool do_kl = !PREP && p.want_kl && mb == 0 && s == 0;   // PREP: the prepare pass owns the KL sums
  const __bf16* xs = reinterpret_cast<const __bf16*>(p.x) + (size_t)(s / p.xg) * (size_t)p.x_sstride;
  const __bf16* xq = reinterpret_cast<const __bf16*>(p.x_sq) + (size_t)(s / p.xg) * (size_t)p.x_sstride;
  const __bf16* xl = X3 ? reinterpret_cast<const __bf16*>(p.x_lo) + (size_t)(s / p.xg) * (size_t)p.x_sstride : nullptr;
  const int T = (N + 15) >> 4;
  if (do_kl && item == 0 && threadIdx.x == 0) p.ws[0] = make_float4(__int_as_float(T), 0.f, 0.f, 0.f);

  // staging: 16 tile pieces per k-step (8 batch tiles of x, 8 of x^2); NW <= 8: wave w brings batch tiles w, w + NW, ...
  // of both; NW == 16: waves 0..7 bring the x tiles, waves 8..15 the x^2 tiles
  constexpr int NX = NW >= 8 ? 1 : 8 / NW;
  // vector-memory operations a wave issues per step (PREP): its DMA pieces + its fragment loads.  X3 with 16 waves: waves 0-7
  // bring a piece of x AND of x lo, waves 8-15 a piece of x^2 -- OPS for the first half, OPS - 1 for the second
  constexpr int OPS = (NW == 16 ? (X3 ? 2 : 1) : NP * NX) + NP;
  size_t xrow[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xrow[i] = (size_t)min(m0 + ((wave & 7) + i * NW) * 16 + r, B - 1) * K;
  auto stage_dma = [&](int t, XtB& xb_) __attribute__((always_inline)) {
    const int kk = min(t * 32 + q * 8, K - 8);
    if (NW == 16) {
      const __bf16* src = (wave < 8 ? xs : xq) + xrow[0] + kk;          // wave-uniform select
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)&xb_[wave >> 3][(wave & 7) * 64], 16, 0, 0);
      if (X3 && wave < 8)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xl + xrow[0] + kk),
                                         (__attribute__((address_space(3))) void*)&xb_[NP - 1][(wave & 7) * 64], 16, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xs + xrow[i] + kk),
                                         (__attribute__((address_space(3))) void*)&xb_[0][(wave + i * NW) * 64], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xq + xrow[i] + kk),
                                         (__attribute__((address_space(3))) void*)&xb_[1][(wave + i * NW) * 64], 16, 0, 0);
        if (X3)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xl + xrow[i] + kk),
                                           (__attribute__((address_space(3))) void*)&xb_[NP - 1][(wave + i * NW) * 64], 16, 0, 0);
      }
    }
  };
  // counted wait "all but this wave's operations of the youngest step have landed"
  auto wait_ring = [&]() __attribute__((always_inline)) {
    if (X3 && NW == 16) {
      if (wave < 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");        // wave-uniform
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
    }
  };
  float mu_n[8], rho_n[8];
  float4 ma_a, sa_a, ma_b, sa_b;                       // PREP: prepared bf16 fragments of the steps in flight
  float4 la_a = make_float4(0.f, 0.f, 0.f, 0.f), la_b = la_a;   // X3: the low part of the mean fragment
  const int tclamp = min(tile, T - 1);
  auto load_frag = [&](int t, float4& ma_r, float4& sa_r, float4& la_r) {
    const float4* src = p.w_frag + ((size_t)tclamp * ksteps + t) * (X3 ? 192 : 128);
    ma_r = src[lane];
    sa_r = src[64 + lane];
    if (X3) la_r = src[128 + lane];
  };
  auto load_raw = [&](int t) {
    const int k = t * 32 + q * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t off = (size_t)min(k + j, K - 1) * N + nc;
      mu_n[j] = p.w_mu[off];
      rho_n[j] = p.w_rho[off];
    }
  };
  float bmu_pre = 0.f, bsig_pre = 0.f, beps_pre = 0.f;
  if (q == 0 && n_ok) {
    bmu_pre = p.b_mu[n];
    bsig_pre = softplus(p.b_rho[n]);
    if (p.eps_mode == BNN_EPS_PHILOX) {
      float e4[4];
      philox_normal4((uint32_t)(n >> 2), gs, p.layer_id * 4u + 1u, p.k0, p.k1, e4);
      beps_pre = (n & 3) == 0 ? e4[0] : (n & 3) == 1 ? e4[1] : (n & 3) == 2 ? e4[2] : e4[3];
    } else if (p.eps_mode == BNN_EPS_MEMORY) {
      beps_pre = p.eps_b[(size_t)s * N + n];
    }
    if (p.eps_b_dump && mb == 0) p.eps_b_dump[(size_t)s * N + n] = beps_pre;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the bias loads are not part of the ring's count
  if (PREP) load_frag(0, ma_a, sa_a, la_a); else load_raw(0);
  stage_dma(0, xt[0]);
  if (DEPTH == 2 && ksteps > 1) {
    load_frag(1, ma_b, sa_b, la_b);
    stage_dma(1, xt[1]);
    wait_ring();
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (PREP) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();
  }

  LR_STAMP(1);
  f32x4 am[8], av[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    am[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    av[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float s_ls = 0.f, s_s2 = 0.f, s_m2 = 0.f;
  // one k-step: consume the fragments of step t (held in ma_r / sa_r), refill them for step t + DEPTH, 16 MFMAs on
  // ring buffer `buf`, then wait for step t + 1's loads and meet the block
  auto step = [&](int t, XtB& cur, XtB& nxt, float4& ma_r, float4& sa_r, float4& la_r, bool steady) __attribute__((always_inline)) {
    bf16x8 ma, sa, ml;
    float mu[8], s2[8];
    if (PREP) {
      ma = __builtin_bit_cast(bf16x8, ma_r);             // a wave past the last tile computes on the last tile's
      sa = __builtin_bit_cast(bf16x8, sa_r);             // fragments (finite) and stores nothing
      if (X3) ml = __builtin_bit_cast(bf16x8, la_r);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        mu[j] = mu_n[j];
        s2[j] = rho_n[j];
      }
    }
    const bool more = steady || t + DEPTH < ksteps;      // block-uniform; `steady`: known at compile time in the main loop
#ifdef BNN_TUNE
    const bool tune_noload = (p.tune & 4) != 0;
#else
    constexpr bool tune_noload = false;
#endif
    if (more && !tune_noload) {
      stage_dma(t + DEPTH, nxt);
      if (PREP) load_frag(t + DEPTH, ma_r, sa_r, la_r); else load_raw(t + DEPTH);
    }
    if (!PREP) {
      const int k = t * 32 + q * 8;
      float ls = 0.f, a2 = 0.f, m2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = n_ok && (k + j) < K;
        const float sig = softplus(s2[j]);
        if (do_kl) {
          ls += ok ? fast_log(sig) : 0.f;
          a2 += ok ? sig * sig : 0.f;
          m2 += ok ? mu[j] * mu[j] : 0.f;
        }
        mu[j] = ok ? mu[j] : 0.f;
        s2[j] = ok ? sig * sig : 0.f;
      }
      s_ls += ls;
      s_s2 += a2;
      s_m2 += m2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ma[j] = (__bf16)mu[j];
        sa[j] = (__bf16)s2[j];
      }
    }
    if (PREP) {
      // The B fragments are read with hand-written ds_read_b128: a compiler-visible LDS load is ordered behind EVERY
      // LDS-DMA in flight that may alias it (s_waitcnt vmcnt(0) ahead of the step's first read), i.e. behind the
      // prefetch this step has just issued -- the step then costs a memory round trip plus its own work.  Here the
      // order is stated by hand: buffer `cur` was complete at the last barrier, and LDS returns in order, so
      // lgkmcnt(2) means "all but the 2 youngest reads are back".  Two fragments in flight behind two in use.
      const uint32_t la = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&cur[0][0] + (uint32_t)((q * 16 + r) * 16);
      f32x4 f0, f1, g0, g1;
#define BNN_LDS2(a, b, O)                                                                         \
  asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                   \
               : "=&v"(a), "=&v"(b)                                                                \
               : "v"(la), "n"((O)), "n"((O) + 1024))
#define BNN_LGKM(N, a, b) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b))
#ifdef BNN_TUNE
#define BNN_MF(ACC, W, F) if (!(p.tune & 1)) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, __builtin_bit_cast(bf16x8, F), ACC, 0, 0, 0)
#define BNN_RD (!(p.tune & 2))
#else
#define BNN_MF(ACC, W, F) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, __builtin_bit_cast(bf16x8, F), ACC, 0, 0, 0)
#define BNN_RD true
#endif
#define BNN_PAIR(M, LAST)                                                                         \
  BNN_LGKM(2, f0, f1);                                                                            \
  BNN_MF(am[M], ma, f0); BNN_MF(am[M + 1], ma, f1);                                               \
  __builtin_amdgcn_sched_barrier(0);                                                              \
  if (!(LAST) && BNN_RD) BNN_LDS2(f0, f1, (M + 2) * 1024);                                                  \
  if (LAST) { BNN_LGKM(0, g0, g1); } else { BNN_LGKM(2, g0, g1); }                                \
  BNN_MF(av[M], sa, g0); BNN_MF(av[M + 1], sa, g1);                                               \
  __builtin_amdgcn_sched_barrier(0);                                                              \
  if (!(LAST) && BNN_RD) BNN_LDS2(g0, g1, 8192 + (M + 2) * 1024);
      // X3: three single fragments in rotation (x, x^2, x lo of batch tile M; single tiles, not pairs: the 16-wave block has 128
      // registers per lane and must not spill -- scratch traffic would enter the counted vmcnt waits of the ring): the x fragment
      // feeds two MFMAs (hi and lo part of the mean operand), x^2 and x lo one each; a fragment is requested one tile ahead
#define BNN_LDS1(a, O) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(a) : "v"(la), "n"((O)))
#define BNN_LGKM1(N, a) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a))
#define BNN_TRIO(M, LAST)                                                                         \
  BNN_LGKM1(2, f0);                                                                               \
  BNN_MF(am[M], ma, f0); BNN_MF(am[M], ml, f0);                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                              \
  if (!(LAST) && BNN_RD) BNN_LDS1(f0, (M + 1) * 1024);                                            \
  if (LAST) { BNN_LGKM1(1, g0); } else { BNN_LGKM1(2, g0); }                                      \
  BNN_MF(av[M], sa, g0);                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                              \
  if (!(LAST) && BNN_RD) BNN_LDS1(g0, 8192 + (M + 1) * 1024);                                     \
  if (LAST) { BNN_LGKM1(0, h0); } else { BNN_LGKM1(2, h0); }                                      \
  BNN_MF(am[M], ma, h0);                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                              \
  if (!(LAST) && BNN_RD) BNN_LDS1(h0, 16384 + (M + 1) * 1024);
      if (X3) {
        f32x4 h0;
        BNN_LDS1(f0, 0);                                 // x, x^2, x lo of batch tile 0
        BNN_LDS1(g0, 8192);
        BNN_LDS1(h0, 16384);
        BNN_TRIO(0, false)
        BNN_TRIO(1, false)
        BNN_TRIO(2, false)
        BNN_TRIO(3, false)
        BNN_TRIO(4, false)
        BNN_TRIO(5, false)
        BNN_TRIO(6, false)
        BNN_TRIO(7, true)
      } else {
        BNN_LDS2(f0, f1, 0);                             // x,   batch tiles 0, 1
        BNN_LDS2(g0, g1, 8192);                          // x^2, batch tiles 0, 1
        BNN_PAIR(0, false)
        BNN_PAIR(2, false)
        BNN_PAIR(4, false)
        BNN_PAIR(6, true)
      }
#undef BNN_TRIO
#undef BNN_LDS1
#undef BNN_LGKM1
#undef BNN_PAIR
#undef BNN_RD
#undef BNN_LDS2
#undef BNN_LGKM
#undef BNN_MF
    } else {
      const float4* xb = cur[0];
      const float4* qb = cur[1];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const bf16x8 xf = __builtin_bit_cast(bf16x8, xb[(m * 4 + q) * 16 + r]);
        const bf16x8 qf = __builtin_bit_cast(bf16x8, qb[(m * 4 + q) * 16 + r]);
        am[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, xf, am[m], 0, 0, 0);
        av[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, qf, av[m], 0, 0, 0);
      }
    }
    // all but this step's own issues have landed; then the block meets.  A bare s_barrier, not __syncthreads(): its
    // workgroup fence would drain every LDS-DMA in flight (vmcnt(0)) and with it the second step of prefetch.  What the
    // barrier must order is stated by hand: this wave's LDS reads of `buf` are complete (the MFMAs consumed them), its
    // DMA pieces of step t + 1 have landed (the vmcnt wait).
    if (DEPTH == 2 && more) wait_ring();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PREP) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      __syncthreads();
    }
  };
  if (DEPTH == 1) {
#pragma nounroll
    for (int t = 0; t < ksteps; ++t) step(t, xt[t & 1], xt[(t + 1) & 1], ma_a, sa_a, la_a, false);
  } else {
    // step t reads ring buffer t % 3 and fills buffer (t + 2) % 3 for step t + 2; the fragments alternate between two
    // register sets.  Steady state: both steps of a pair refill (no branch in the body), so the compiler's own
    // wait-count bookkeeping for the fragment registers stays exact instead of draining to zero at the back edge.
    int t = 0, b0 = 0;
    // (the compiler still drains vmcnt to zero at the loop header -- it cannot see the hand-written waits -- so the
    // body holds six steps: one full drain per six)
#pragma nounroll
    for (; t + 7 < ksteps; t += 6) {
      step(t, xt[0], xt[2], ma_a, sa_a, la_a, true);
      step(t + 1, xt[1], xt[0], ma_b, sa_b, la_b, true);        // step t + 3 reuses the buffer of step t
      step(t + 2, xt[2], xt[1], ma_a, sa_a, la_a, true);
      step(t + 3, xt[0], xt[2], ma_b, sa_b, la_b, true);
      step(t + 4, xt[1], xt[0], ma_a, sa_a, la_a, true);
      step(t + 5, xt[2], xt[1], ma_b, sa_b, la_b, true);
    }
#pragma nounroll
    for (; t < ksteps; t += 2) {
      const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;
      step(t, xt[b0], xt[b2], ma_a, sa_a, la_a, false);
      if (t + 1 < ksteps) step(t + 1, xt[b1], xt[b0], ma_b, sa_b, la_b, false);
      b0 = b2;
    }
  }

  LR_STAMP(2);
  if (q == 0) {
    float b = 0.f;
    if (n_ok) {
      b = __builtin_fmaf(bsig_pre, beps_pre, bmu_pre);
      if (do_kl) {
        s_ls += fast_log(bsig_pre);
        s_s2 = __builtin_fmaf(bsig_pre, bsig_pre, s_s2);
        s_m2 = __builtin_fmaf(bmu_pre, bmu_pre, s_m2);
      }
    }
    bias_s[wave][r] = b;
  }
  if (do_kl) {
    const float a = wave_sum(s_ls), b = wave_sum(s_s2), cc = wave_sum(s_m2);
    if (lane == 0 && tile < T) p.ws[1 + tile] = make_float4(a, b, cc, 0.f);
  }
  __syncthreads();
  const int nb = tile * 16 + q * 4;
  const bool vec_ok = (N & 3) == 0;
  const int gprN = (N + 3) >> 2;
  if (nb < N) {
    float bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = bias_s[wave][q * 4 + i];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int brow = m0 + m * 16 + r;
#ifdef BNN_TUNE
      if (brow < B && !((p.tune & 16) && brow > 0)) {
#else
      if (brow < B) {
#endif
        const size_t yoff = ((size_t)s * B + brow) * N + nb;
        float e4[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef BNN_TUNE
        if (p.eps_mode == BNN_EPS_PHILOX && !(p.tune & 8)) {
#else
        if (p.eps_mode == BNN_EPS_PHILOX) {
#endif
          philox_normal4((uint32_t)brow * (uint32_t)gprN + (uint32_t)(nb >> 2), gs, p.layer_id * 4u + 2u, p.k0, p.k1, e4);
        } else if (p.eps_mode == BNN_EPS_MEMORY) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < N) e4[i] = p.eps_act[yoff + i];
        }
        if (p.eps_act_dump) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < N) p.eps_act_dump[yoff + i] = e4[i];
        }
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float o = __builtin_fmaf(__builtin_amdgcn_sqrtf(av[m][i]), e4[i], am[m][i]) + bq[i];
          if (p.relu) o = fmaxf(o, 0.f);
          v[i] = o;
        }
        if (p.y_sq) {
          __bf16* qp = reinterpret_cast<__bf16*>(p.y_sq) + yoff;
          if (vec_ok) {
            bf16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = X3 ? (__bf16)(v[i] * v[i]) : sq_bf16(v[i]);
            *reinterpret_cast<bf16x4*>(qp) = o;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (nb + i < N) qp[i] = X3 ? (__bf16)(v[i] * v[i]) : sq_bf16(v[i]);
          }
        }
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
          if (X3) {                                  // the low plane of the output's split pair
            __bf16* lp = reinterpret_cast<__bf16*>(p.y_lo) + yoff;
            bf16x4 l;
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = split_lo(v[i], o[i]);
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
#ifdef BNN_STAMPS
  LR_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LR_STAMP(4);
  LR_STAMP_RT(9);
#endif
}

// K3r  the output layer of a few-sample LR evaluation, split by batch rows, with the finalize: the LR counterpart of
// bbb_final_rows_kernel.  Grid: per sample RB = ceil(B / 16) row blocks + 1 statistics block, 256 threads each.
//   row block : m = x . M and v = x^2 . sigma^2 of its 16 batch rows for all (<= 16) output features (4 waves split the
//               k-steps; the [in,out] weights of a 10-column layer are 48 KB per tensor and stay in L2), activation noise
//               and bias in the D layout (networks.py:120-128), logits stored, the rows' NLL (networks.py:183-190);
//   stats     : the KL of every layer (networks.py:109-114, :179-181): the layers below from their KL workspaces, this
//               layer's closed-form sums from its own parameters.
// Hand-off as in bbb_final_rows_kernel (nobody waits): each block publishes its scalar(s) write-through, drains, takes
// the sample's ticket; the last one folds them in a fixed order and writes the sample's outputs; samples meet the
// same way.  A separate K3a launch + bnn_elbo_finalize cost 9.4 + 7.3 us and two launch boundaries at one sample.
struct LrRows {
  const __bf16* x;      // [S | shared, B, K]
  long x_sstride;
  int xg;
  const float *w_mu, *w_rho, *b_mu, *b_rho;   // [K, N], [N]
  float* y;             // [S, B, N]
  float* v_out;         // optional [S, B, N]: the variance the rows were sampled from (a training step's backward reads it)
  int S, B, K, N, relu;
  uint32_t k0, k1, layer_id, sample_offset;
  const uint32_t* sample_counter;
  uint32_t sgrp, sgrp_stride;
  uint32_t* tickets;    // [S], zero between launches
  float* parts;         // [S][16]: 0..7 the row blocks' NLL, 8 the KL
  const float4* w_frag; // optional prepared operands of this layer (bnn_lr_prepare, or the rider of the previous layer's launch):
                        // [k-step][mean | variance][64] x 16 B -- the row blocks then park nothing
  const float4* ws_own; // with w_frag: the KL workspace that came with them (header {entries}, then the sums)
  int rt;               // 16-row tiles a row block takes one after the other (1: one tile per block, the latency form; prepared
                        // operands and many samples: 2 (up to kRowsMaxTiles) -- the block's fragments stay in registers, and at
                        // 188 registers only two blocks fit a CU: 2560 one-tile blocks of a 256-pair launch ran as five rounds, 39 us)
};
constexpr int kRowsBatch = 12;   // 16-byte loads of each weight tensor a thread keeps in flight (12 x 256 x 4 = the 1200 x 10 layer)
constexpr int kRowsX = 10;       // x fragments a wave requests up front (4 waves x 10 k-steps = K up to 1280)
constexpr int kRowsMaxTiles = 4; // 16-row tiles a row block of the many-pairs form takes at most (LrRows.rt)
struct LrFin {
  FinK k;
  FinC c;
  float* sums;
  uint32_t* ticket;
};

// MULTI: a row block takes p.rt 16-row tiles one after the other (a compile-time fact: written as a run-time loop, the one-tile
// form -- the latency form of a few-pair evaluation -- compiled to 164 registers + 528 bytes of scratch instead of 188 and none)
template <bool MULTI>
__global__ __launch_bounds__(256) void lr_final_rows_kernel(const LrRows p, const LrFin fp, const FinLoss tr) {
  __shared__ __attribute__((aligned(16))) f32x4 red_m[4][64], red_v[4][64];
  __shared__ float lg[16][17];
  __shared__ __attribute__((aligned(8))) float part[4 * kFinNV];
  __shared__ float kl_red[4 * 3];
  extern __shared__ __attribute__((aligned(16))) __bf16 wfrag_s[];     // 2 x ceil(K / 32) x 4 x 16 x 8 bf16
  const FinK& fk = fp.k;
  const int RB = (p.B + 15) >> 4;
  const int RT = MULTI ? p.rt : 1, NBLK = (RB + RT - 1) / RT;  // row blocks per sample (+ 1 statistics block)
  const int s = (int)blockIdx.x / (NBLK + 1), bi = (int)blockIdx.x - s * (NBLK + 1);
  int rb = bi == NBLK ? RB : bi * RT;                          // the statistics block | the block's first 16-row tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = p.K, N = p.N, B = p.B;
  float* const mine = p.parts + (size_t)s * 16;
  uint32_t gs = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u);
  if (p.sgrp == 0u) gs += (uint32_t)s;
  else gs += ((uint32_t)s / p.sgrp) * p.sgrp_stride + (uint32_t)s % p.sgrp;
  const bool vec4 = !((reinterpret_cast<uintptr_t>(p.w_mu) | reinterpret_cast<uintptr_t>(p.w_rho)) & 15);
  float pub0 = 0.f;
  int slot = rb;
  if (rb == RB) {
    // ---- statistics block: this layer's closed-form KL sums from its parameters, the layers below from their workspaces
    float own0 = 0.f, own1 = 0.f, own2 = 0.f;
    if (p.ws_own) {                                            // block-uniform: the sums came with the prepared operands
      const int Town = __float_as_int(p.ws_own[0].x);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      for (int e = threadIdx.x; e < Town; e += 256) {
        const float4 v = p.ws_own[1 + e];
        a0 += v.x; a1 += v.y; a2 += v.z;
      }
      a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
      if (lane == 0) {
        kl_red[wave * 3 + 0] = a0;
        kl_red[wave * 3 + 1] = a1;
        kl_red[wave * 3 + 2] = a2;
      }
    } else {
      float ls = 0.f, s2 = 0.f, m2 = 0.f;
      auto kl_term = [&](float mu, float rho) {
        const float sig = softplus(rho);
        ls += fast_log(sig);
        s2 = __builtin_fmaf(sig, sig, s2);
        m2 = __builtin_fmaf(mu, mu, m2);
      };
      const int KN = K * N, n4 = vec4 ? KN >> 2 : 0;
      for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * kRowsBatch) {   // kRowsBatch 16-byte loads of each tensor in flight
        float4 a[kRowsBatch], b[kRowsBatch];
#pragma unroll
        for (int u = 0; u < kRowsBatch; ++u) {
          const int i = min(i0 + u * 256, n4 - 1);
          a[u] = reinterpret_cast<const float4*>(p.w_mu)[i];
          b[u] = reinterpret_cast<const float4*>(p.w_rho)[i];
        }
#pragma unroll
        for (int u = 0; u < kRowsBatch; ++u)
          if (i0 + u * 256 < n4) {
            kl_term(a[u].x, b[u].x); kl_term(a[u].y, b[u].y); kl_term(a[u].z, b[u].z); kl_term(a[u].w, b[u].w);
          }
      }
      for (int i = n4 * 4 + threadIdx.x; i < KN + N; i += 256) {
        const bool w = i < KN;
        kl_term(w ? p.w_mu[i] : p.b_mu[i - KN], w ? p.w_rho[i] : p.b_rho[i - KN]);
      }
      const float a0 = wave_sum(ls), a1 = wave_sum(s2), a2 = wave_sum(m2);
      if (lane == 0) {
        kl_red[wave * 3 + 0] = a0;
        kl_red[wave * 3 + 1] = a1;
        kl_red[wave * 3 + 2] = a2;
      }
    }
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
      own0 += kl_red[wv * 3 + 0];
      own1 += kl_red[wv * 3 + 1];
      own2 += kl_red[wv * 3 + 2];
    }
    const int own_layer = fk.n_layers - 1;
    int T[8];
#pragma unroll
    for (int l = 0; l < 8; ++l)
      T[l] = (l < own_layer) ? __float_as_int(reinterpret_cast<const float4*>(fk.ws[l])[0].x) : 0;
    float a = 0.f, b = 0.f, nll = 0.f;
    fin_sample(fk, fp.c, s, T, nullptr, 0, own_layer, own0, own1, own2, part, a, b, nll);
    pub0 = a;
    slot = 8;
  } else {
    const int r = lane & 15, q = lane >> 4;
    const int rb_end = min(rb + RT, RB);
    int row = min(rb * 16 + r, B - 1);
    const __bf16* xr = p.x + (size_t)(s / p.xg) * (size_t)p.x_sstride + (size_t)row * K;
    const int ksteps = (K + 31) >> 5;
    // ---- the layer's weights, once per block: whole-line reads of the [K, N] matrices (a lane gathering its fragment
    // straight from them -- 16 four-byte loads per k-step, ten dependent rounds per wave -- made this kernel 14 us),
    // bf16 M and sigma^2 parked in LDS in fragment order: [k / 8][feature 0..15][k % 8], 16 bytes per (octet, feature)
    __bf16* const m_s = wfrag_s;
    __bf16* const v_s = wfrag_s + (size_t)ksteps * 4 * 16 * 8;
    // this wave's x fragments (its first kRowsX k-steps) go out BEFORE the weights: one round trip covers both
    // (MULTI: requested at the top of every tile instead -- an array carried around the tile loop is not promoted to registers)
    float4 xq[kRowsX];
    if (!MULTI) {
#pragma unroll
      for (int u = 0; u < kRowsX; ++u)
        xq[u] = *reinterpret_cast<const float4*>(xr + min((wave + 4 * u) * 32 + q * 8, K - 8));
    }
    // the lane's four biases (wave 0 applies them after the reduction): requested now, at clamped addresses by every lane -- read
    // in the epilogue they were a round trip of their own behind the barrier
    float bmu4[4], brho4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bmu4[i] = p.b_mu[min(q * 4 + i, N - 1)];
      brho4[i] = p.b_rho[min(q * 4 + i, N - 1)];
    }
    // prepared operands (p.w_frag, block-uniform): the wave's fragments of its first kRowsX k-steps straight from memory, in the
    // same round trip as x -- nothing is parked, no barrier
    float4 fm[kRowsX], fv[kRowsX];
    if (p.w_frag) {
#pragma unroll
      for (int u = 0; u < kRowsX; ++u) {
        const int t = min(wave + 4 * u, ksteps - 1);
        fm[u] = p.w_frag[(size_t)(t * 2 + 0) * 64 + lane];
        fv[u] = p.w_frag[(size_t)(t * 2 + 1) * 64 + lane];
      }
    } else {
      // zero what no weight will fill: the padding features (n >= N) of every octet and the octets past K
      const int pad = 16 - N, octs = ksteps * 4, full = K >> 3;
      for (int i = threadIdx.x; i < octs * pad; i += 256) {
        const int o = i / pad, n = N + (i - o * pad);
        reinterpret_cast<float4*>(m_s)[o * 16 + n] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(v_s)[o * 16 + n] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      for (int i = threadIdx.x; i < (octs - full) * N; i += 256) {
        const int o = full + i / N, n = i % N;
        reinterpret_cast<float4*>(m_s)[o * 16 + n] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(v_s)[o * 16 + n] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float inv_n = 1.0f / (float)N;
      auto park = [&](int i, float mu, float rho) {               // element i = k * N + n
        int k = (int)((float)i * inv_n);                          // i / N without the integer divide (i < 2^24: one fix-up)
        k += ((k + 1) * N <= i) ? 1 : 0;
        k -= (k * N > i) ? 1 : 0;
        const int n = i - k * N;
        const float sig = softplus(rho);
        const int at = ((k >> 3) * 16 + n) * 8 + (k & 7);
        m_s[at] = (__bf16)mu;
        v_s[at] = (__bf16)(sig * sig);
      };
      {
        const int KN = K * N, n4 = vec4 ? KN >> 2 : 0;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * kRowsBatch) {
          float4 a[kRowsBatch], b[kRowsBatch];
  #pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int i = min(i0 + u * 256, n4 - 1);
            a[u] = reinterpret_cast<const float4*>(p.w_mu)[i];
            b[u] = reinterpret_cast<const float4*>(p.w_rho)[i];
          }
  #pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int i = i0 + u * 256;
            if (i < n4) {
              park(4 * i, a[u].x, b[u].x); park(4 * i + 1, a[u].y, b[u].y); park(4 * i + 2, a[u].z, b[u].z); park(4 * i + 3, a[u].w, b[u].w);
            }
          }
        }
        for (int i = n4 * 4 + threadIdx.x; i < KN; i += 256) park(i, p.w_mu[i], p.w_rho[i]);
      }
      __syncthreads();
    }
    // the block's 16-row tiles (one, unless p.rt > 1): a loop of constant trip count, fully unrolled -- around a run-time loop the
    // x / fragment arrays are not promoted to registers (528 bytes of scratch)
#pragma unroll
    for (int it = 0; it < (MULTI ? kRowsMaxTiles : 1); ++it) {
    if (MULTI) {
#pragma unroll
      for (int u = 0; u < kRowsX; ++u)
        xq[u] = *reinterpret_cast<const float4*>(xr + min((wave + 4 * u) * 32 + q * 8, K - 8));
    }
    f32x4 am = f32x4{0.f, 0.f, 0.f, 0.f}, av = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kRowsX + 0; ++u) {                      // the prefetched steps, then (long K only) the rest
      const int t = wave + 4 * u;
      if (!MULTI && t >= ksteps) break;                         // (MULTI: no second exit, so that the loop unrolls inside the tile
                                                                // loop -- a step past K multiplies zeros: k >= K below)
      const int k = t * 32 + q * 8;
      const bf16x8 xb = __builtin_bit_cast(bf16x8, xq[u]);
      const bf16x8 ma = p.w_frag ? __builtin_bit_cast(bf16x8, fm[u]) : *reinterpret_cast<const bf16x8*>(m_s + ((size_t)(t * 4 + q) * 16 + r) * 8);
      const bf16x8 sa = p.w_frag ? __builtin_bit_cast(bf16x8, fv[u]) : *reinterpret_cast<const bf16x8*>(v_s + ((size_t)(t * 4 + q) * 16 + r) * 8);
      bf16x8 xz, x2b;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = (k < K) ? (float)xb[j] : 0.f;
        xz[j] = (__bf16)xv;
        x2b[j] = (__bf16)(xv * xv);
      }
      am = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, xz, am, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, x2b, av, 0, 0, 0);
    }
#pragma unroll 2
    for (int t = wave + 4 * kRowsX; t < ksteps; t += 4) {
      const int k = t * 32 + q * 8;
      const bf16x8 xb = *reinterpret_cast<const bf16x8*>(xr + min(k, K - 8));
      const bf16x8 ma = p.w_frag ? __builtin_bit_cast(bf16x8, p.w_frag[(size_t)(t * 2 + 0) * 64 + lane])
                                 : *reinterpret_cast<const bf16x8*>(m_s + ((size_t)(t * 4 + q) * 16 + r) * 8);
      const bf16x8 sa = p.w_frag ? __builtin_bit_cast(bf16x8, p.w_frag[(size_t)(t * 2 + 1) * 64 + lane])
                                 : *reinterpret_cast<const bf16x8*>(v_s + ((size_t)(t * 4 + q) * 16 + r) * 8);
      bf16x8 xz, x2b;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = (k + j < K && k <= K - 8) ? (float)xb[j] : 0.f;   // (K % 8 == 0: a fragment is whole or absent)
        xz[j] = (__bf16)xv;
        x2b[j] = (__bf16)(xv * xv);
      }
      am = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, xz, am, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, x2b, av, 0, 0, 0);
    }
    red_m[wave][lane] = am;
    red_v[wave][lane] = av;
    __syncthreads();
    if (wave == 0) {
      // lane (r = batch row of the block, q): features 4q .. 4q+3
      f32x4 m = red_m[0][lane], v = red_v[0][lane];
#pragma unroll
      for (int wv = 1; wv < 4; ++wv) {
        m += red_m[wv][lane];
        v += red_v[wv][lane];
      }
      const int brow = rb * 16 + r;
      const int gprN = (N + 3) >> 2;
      float ea[4], eb[4];
      philox_normal4((uint32_t)min(brow, B - 1) * (uint32_t)gprN + (uint32_t)q, gs, p.layer_id * 4u + 2u, p.k0, p.k1, ea);
      philox_normal4((uint32_t)q, gs, p.layer_id * 4u + 1u, p.k0, p.k1, eb);
      float o4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = q * 4 + i;
        const float bias = n < N ? __builtin_fmaf(softplus(brho4[i]), eb[i], bmu4[i]) : 0.f;
        float o = __builtin_fmaf(__builtin_amdgcn_sqrtf(v[i]), ea[i], m[i]) + bias;
        if (p.relu) o = fmaxf(o, 0.f);
        o4[i] = o;
        lg[r][n] = o;
      }
      if (brow < B) {
        float* yp = p.y + ((size_t)s * B + brow) * N + q * 4;
        if ((N & 3) == 0 && q * 4 < N) {
          *reinterpret_cast<float4*>(yp) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (q * 4 + i < N) yp[i] = o4[i];
        }
        if (p.v_out) {
          float* vp = p.v_out + ((size_t)s * B + brow) * N + q * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (q * 4 + i < N) vp[i] = v[i];
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
      // ---- NLL of the block's rows: lane r < 16 takes row r
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
        if (tr.out4) fin_loss_row_grad(fk, tr, s, B, brow, lg[lane]);
      }
      pub0 = wave_sum(acc_n);
    }
    if (!MULTI || rb + 1 >= rb_end) break;                     // block-uniform: the last (or only) tile publishes below
    // ---- a further tile: publish this one's NLL, request the next tile's x (the fragments stay in registers)
    if (threadIdx.x == 0) __hip_atomic_store(mine + rb, pub0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++rb;
    slot = rb;
    row = min(rb * 16 + r, B - 1);
    xr = p.x + (size_t)(s / p.xg) * (size_t)p.x_sstride + (size_t)row * K;
    __syncthreads();                                           // red_m / red_v / lg of the tile before are free again
    }
  }
  if (threadIdx.x != 0) return;
  __hip_atomic_store(mine + slot, pub0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t tk = __hip_atomic_fetch_add(p.tickets + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tk != (uint32_t)NBLK) return;                         // NBLK + 1 blocks per sample
  float pv[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) pv[i] = __hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one round trip
  double tn = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < RB) tn += pv[i];
  const float a = pv[8];
  const float nll = (float)tn;
  __hip_atomic_store(p.tickets + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
  if (fk.kl) __hip_atomic_store(fk.kl + s, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fk.nll) __hip_atomic_store(fk.nll + s, nll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fk.S == 1) {
    if (fp.sums) {
      fp.sums[0] = a; fp.sums[1] = 0.f; fp.sums[2] = nll; fp.sums[3] = 1.f;
    }
    if (tr.out4) fin_loss_assemble(fk, tr);
    if (fk.sample_counter) *fk.sample_counter += fk.sample_counter_inc;
    return;
  }
  if (!fp.ticket) return;                                   // (the samples meet in a follow-up launch)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t t2 = __hip_atomic_fetch_add(fp.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t2 != (uint32_t)fk.S - 1u) return;
  if (fp.sums) fin_fold_sums(fk, fp.sums);
  if (tr.out4) fin_loss_assemble(fk, tr);
  __hip_atomic_store(fp.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fk.sample_counter) *fk.sample_counter += fk.sample_counter_inc;
}

// KL of one LR layer from its partials (networks.py:113, :134-136).  The partials fold
// weights and biases together (the network only needs the total); the bias term is small
// (N elements) and is recomputed here so that weight_kl_cost / bias_kl_cost can be
// reported separately.  out3 = {kl, weight_kl, bias_kl}.
__global__ void lr_layer_kl_kernel(const float4* __restrict__ ws, int K, int N, float sigma_p,
                                   const float* __restrict__ b_mu, const float* __restrict__ b_rho,
                                   float* __restrict__ out3) {
  __shared__ double scratch[16];
  double ls = 0, s2 = 0, m2 = 0, bls = 0, bs2 = 0, bm2 = 0;
  const int T = __float_as_int(ws[0].x);
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float4 v = ws[1 + t];
    ls += v.x;
    s2 += v.y;
    m2 += v.z;
  }
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const float sig = softplus(b_rho[i]), mu = b_mu[i];
    bls += fast_log(sig);
    bs2 += __builtin_fmaf(sig, sig, 0.f);
    bm2 += __builtin_fmaf(mu, mu, 0.f);
  }
  ls = block_sum(ls, scratch);
  s2 = block_sum(s2, scratch);
  m2 = block_sum(m2, scratch);
  bls = block_sum(bls, scratch);
  bs2 = block_sum(bs2, scratch);
  bm2 = block_sum(bm2, scratch);
  if (threadIdx.x == 0) {
    const double cnt = (double)N * K + N, sp = sigma_p;
    const double kl = 0.5 * (2.0 * cnt * log(sp) - 2.0 * ls - cnt + (s2 + m2) / (sp * sp));
    const double bkl = 0.5 * (2.0 * N * log(sp) - 2.0 * bls - N + (bs2 + bm2) / (sp * sp));
    out3[0] = (float)kl;
    out3[1] = (float)(kl - bkl);
    out3[2] = (float)bkl;
  }
}

}  // namespace bnn

using namespace bnn;

// One float4 entry per KL writer: a feature tile of the layer kernels (at most ceil(N / 4)), or a block of
// bnn_lr_prepare (k-step range x 128 features: up to kPrepEntries, so that it can run as several hundred blocks).
static constexpr int kPrepEntries = 4096;
extern "C" size_t bnn_lr_linear_fwd_workspace_bytes(int32_t out_features) {
  if (out_features <= 0) return 0;
  const size_t tiles = (size_t)((out_features + 3) / 4);
  return (1 + (tiles > (size_t)kPrepEntries ? tiles : (size_t)kPrepEntries)) * 4 * sizeof(float);
}

extern "C" size_t bnn_lr_prepare_bytes(int32_t in_features, int32_t out_features) {
  if (in_features <= 0 || out_features <= 0) return 0;
  return (size_t)((out_features + 15) / 16) * (size_t)((in_features + 31) / 32) * 128 * 16;
}

extern "C" size_t bnn_lr_prepare_x3_bytes(int32_t in_features, int32_t out_features) {
  return bnn_lr_prepare_bytes(in_features, out_features) / 2 * 3;
}

// argument checks of one layer's prepare + its block geometry (kb k-steps x 128 features per block)
static int lr_prepare_geometry(const float* w_mu, const float* w_rho, const float* b_mu, const float* b_rho,
                               int32_t in_features, int32_t out_features, void* w_frag, size_t w_frag_bytes,
                               void* kl_workspace, size_t kl_workspace_bytes, bool x3, int& kb, int& nkb, int& groups) {
  if (!w_mu || !w_rho || !b_mu || !b_rho || !w_frag) return BNN_ERR_NULL;
  if (in_features <= 0 || out_features <= 0) return BNN_ERR_SHAPE;
  if (w_frag_bytes < (x3 ? bnn_lr_prepare_x3_bytes(in_features, out_features) : bnn_lr_prepare_bytes(in_features, out_features))) return BNN_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(w_frag) & 15) return BNN_ERR_ALIGN;
  if (kl_workspace) {
    if (kl_workspace_bytes < bnn_lr_linear_fwd_workspace_bytes(out_features)) return BNN_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(kl_workspace) & 15) return BNN_ERR_ALIGN;
  }
  const int ksteps = (in_features + 31) / 32;
  groups = (out_features + 127) / 128;
  const int max_entries = (out_features + 3) / 4 > kPrepEntries ? (out_features + 3) / 4 : kPrepEntries;
  kb = 1;
  while ((long)((ksteps + kb - 1) / kb) * groups > max_entries) ++kb;
  nkb = (ksteps + kb - 1) / kb;
  return BNN_OK;
}

extern "C" int bnn_lr_prepare_many(const bnn_lr_prepare_job* jobs, int32_t n_jobs, int32_t x3, void* stream_) {
  if (!jobs) return BNN_ERR_NULL;
  if (n_jobs <= 0 || n_jobs > kPrepManyJobs) return BNN_ERR_SHAPE;
  LrPrepJobs jb{};
  jb.n = n_jobs;
  int total = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const bnn_lr_prepare_job& q = jobs[j];
    int kb, nkb, groups;
    const int rc = lr_prepare_geometry(q.w_mu, q.w_rho, q.b_mu, q.b_rho, q.in_features, q.out_features, q.w_frag, q.w_frag_bytes,
                                       q.kl_workspace, q.kl_workspace_bytes, x3 != 0, kb, nkb, groups);
    if (rc != BNN_OK) return rc;
    jb.w_mu[j] = q.w_mu; jb.w_rho[j] = q.w_rho; jb.b_mu[j] = q.b_mu; jb.b_rho[j] = q.b_rho;
    jb.frag[j] = reinterpret_cast<float4*>(q.w_frag);
    jb.ws[j] = reinterpret_cast<float4*>(q.kl_workspace);
    jb.K[j] = q.in_features; jb.N[j] = q.out_features; jb.kb[j] = kb; jb.nkb[j] = nkb; jb.ngrp[j] = groups;
    jb.first[j] = total;
    total += nkb * groups;
  }
  for (int j = n_jobs; j <= kPrepManyJobs; ++j) jb.first[j] = total;
  if (x3) hipLaunchKernelGGL(lr_prepare_many_kernel<true>, dim3((unsigned)total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), jb);
  else hipLaunchKernelGGL(lr_prepare_many_kernel<false>, dim3((unsigned)total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), jb);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

static int lr_prepare_impl(const float* w_mu, const float* w_rho, const float* b_mu, const float* b_rho,
                           int32_t in_features, int32_t out_features, void* w_frag, size_t w_frag_bytes,
                           void* kl_workspace, size_t kl_workspace_bytes, void* stream_, bool x3) {
  if (!w_mu || !w_rho || !b_mu || !b_rho || !w_frag) return BNN_ERR_NULL;
  if (in_features <= 0 || out_features <= 0) return BNN_ERR_SHAPE;
  if (w_frag_bytes < (x3 ? bnn_lr_prepare_x3_bytes(in_features, out_features) : bnn_lr_prepare_bytes(in_features, out_features))) return BNN_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(w_frag) & 15) return BNN_ERR_ALIGN;
  if (kl_workspace) {
    if (kl_workspace_bytes < bnn_lr_linear_fwd_workspace_bytes(out_features)) return BNN_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(kl_workspace) & 15) return BNN_ERR_ALIGN;
  }
  // blocks of kb k-steps x 128 features, at most one KL entry per 4 features (the workspace's size): kb from that
  const int ksteps = (in_features + 31) / 32, groups = (out_features + 127) / 128;
  const int max_entries = (out_features + 3) / 4 > kPrepEntries ? (out_features + 3) / 4 : kPrepEntries;
  int kb = 1;
  while ((long)((ksteps + kb - 1) / kb) * groups > max_entries) ++kb;
  if (x3)
    hipLaunchKernelGGL(lr_prepare_tiled_kernel<true>, dim3((unsigned)((ksteps + kb - 1) / kb), (unsigned)groups), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream_), w_mu, w_rho, b_mu, b_rho, in_features, out_features, kb,
                       reinterpret_cast<float4*>(w_frag), reinterpret_cast<float4*>(kl_workspace));
  else
    hipLaunchKernelGGL(lr_prepare_tiled_kernel<false>, dim3((unsigned)((ksteps + kb - 1) / kb), (unsigned)groups), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream_), w_mu, w_rho, b_mu, b_rho, in_features, out_features, kb,
                       reinterpret_cast<float4*>(w_frag), reinterpret_cast<float4*>(kl_workspace));
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}

extern "C" int bnn_lr_prepare(const float* w_mu, const float* w_rho, const float* b_mu, const float* b_rho,
                              int32_t in_features, int32_t out_features, void* w_frag, size_t w_frag_bytes,
                              void* kl_workspace, size_t kl_workspace_bytes, void* stream_) {
  return lr_prepare_impl(w_mu, w_rho, b_mu, b_rho, in_features, out_features, w_frag, w_frag_bytes, kl_workspace, kl_workspace_bytes, stream_, false);
}

extern "C" int bnn_lr_prepare_x3(const float* w_mu, const float* w_rho, const float* b_mu, const float* b_rho,
                                 int32_t in_features, int32_t out_features, void* w_frag, size_t w_frag_bytes,
                                 void* kl_workspace, size_t kl_workspace_bytes, void* stream_) {
  return lr_prepare_impl(w_mu, w_rho, b_mu, b_rho, in_features, out_features, w_frag, w_frag_bytes, kl_workspace, kl_workspace_bytes, stream_, true);
}

// K3b in split-bf16 math: k-steps of prefetch.  One: the block-wide kernels have 128 registers per lane, and with a second
// set of three prefetched fragments in flight the compiler spills INSIDE the k loop -- scratch loads and stores count in vmcnt
// like every vector-memory operation, so the ring's counted waits ("all but this step's operations have landed") would no
// longer say what they mean.  With one step of prefetch every wait in the loop is vmcnt(0).
#ifndef BNN_LR_X3_DEPTH
#define BNN_LR_X3_DEPTH 1
#endif
static constexpr int kLrX3Depth = BNN_LR_X3_DEPTH;

// validate the arguments and fill the kernel parameter block
static int lr_fill(const bnn_lr_fwd_args* a, LrK& k) {
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_lr_fwd_args)) return BNN_ERR_ABI;
  if (a->n_samples <= 0 || a->batch <= 0 || a->in_features <= 0 || a->out_features <= 0) return BNN_ERR_SHAPE;
  if ((double)a->n_samples * ((a->batch + 127) / 128) * ((a->out_features + 3) / 4) > 2.0e9) return BNN_ERR_SHAPE;
  if (!a->x || !a->w_mu || !a->w_rho || !a->b_mu || !a->b_rho || !a->y) return BNN_ERR_NULL;
  if ((unsigned)a->x_dtype > 1u || (unsigned)a->y_dtype > 1u || (unsigned)a->math > 2u || (unsigned)a->eps_mode > 2u)
    return BNN_ERR_ENUM;
  k.x_lo = nullptr; k.y_lo = nullptr;
  if (a->math == BNN_MATH_BF16X3) {
    // split-bf16 math: the block-GEMM form over bnn_lr_prepare_x3's fragments and the (x, x_lo, x_sq) planes -- nothing else
    if (!a->x_lo || !a->x_sq || !a->w_frag) return BNN_ERR_NULL;
    if (a->x_dtype != BNN_BF16 || a->v_out || a->hfac_out || a->y_bf16_copy || a->rider || (a->in_features % 8)) return BNN_ERR_ENUM;
    if ((reinterpret_cast<uintptr_t>(a->x_lo) | reinterpret_cast<uintptr_t>(a->x_sq) | reinterpret_cast<uintptr_t>(a->w_frag)) & 15) return BNN_ERR_ALIGN;
    k.x_lo = a->x_lo;
    if (a->y_dtype == BNN_BF16) {
      if (!a->y_lo) return BNN_ERR_NULL;
      if ((a->out_features % 4 == 0) && (reinterpret_cast<uintptr_t>(a->y_lo) & 7)) return BNN_ERR_ALIGN;
      k.y_lo = a->y_lo;
    }
  }
  if (a->eps_mode == BNN_EPS_MEMORY && (!a->eps_act || !a->eps_b)) return BNN_ERR_NULL;
  if (a->want_kl) {
    if (!a->workspace || a->workspace_bytes < bnn_lr_linear_fwd_workspace_bytes(a->out_features))
      return BNN_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(a->workspace) & 15) return BNN_ERR_ALIGN;
    if (!(a->sigma_p > 0.f)) return BNN_ERR_SHAPE;
  }
  if (a->kl_out && !a->want_kl) return BNN_ERR_WORKSPACE;
  const bool ybf = a->y_dtype == BNN_BF16;
  if ((a->out_features % 4 == 0) && (reinterpret_cast<uintptr_t>(a->y) & (ybf ? 7 : 15))) return BNN_ERR_ALIGN;
  if ((a->in_features % 8 == 0) && (reinterpret_cast<uintptr_t>(a->x) & 15)) return BNN_ERR_ALIGN;
  k.x = a->x;
  k.x_sstride = a->x_per_sample ? (long)a->batch * a->in_features : 0;
  k.xg = a->x_per_sample > 0 ? a->x_per_sample : 1;
  k.sgrp = a->sample_group; k.sgrp_stride = a->sample_group_stride;
  k.ksl = 1; k.nst = 0; k.es = 1; k.ks_ticket = nullptr; k.ks_part = nullptr;
  k.rd_blocks = 0; k.main_blocks = 0;
  if (a->rider) {
    const bnn_lr_rider* rd = a->rider;
    if (rd->struct_bytes != sizeof(bnn_lr_rider)) return BNN_ERR_ABI;
    if (!rd->w_mu || !rd->w_rho || !rd->b_mu || !rd->b_rho || !rd->w_frag || !rd->kl_workspace) return BNN_ERR_NULL;
    if (rd->in_features <= 0 || rd->out_features <= 0) return BNN_ERR_SHAPE;
    if (rd->w_frag_bytes < bnn_lr_prepare_bytes(rd->in_features, rd->out_features) ||
        rd->kl_workspace_bytes < bnn_lr_linear_fwd_workspace_bytes(rd->out_features))
      return BNN_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(rd->w_frag) | reinterpret_cast<uintptr_t>(rd->kl_workspace)) & 15) return BNN_ERR_ALIGN;
  }
  k.w_mu = a->w_mu; k.w_rho = a->w_rho; k.b_mu = a->b_mu; k.b_rho = a->b_rho;
  k.eps_act = a->eps_act; k.eps_b = a->eps_b; k.eps_act_dump = a->eps_act_dump; k.eps_b_dump = a->eps_b_dump;
  k.y = a->y;
  k.x_sq = a->x_sq; k.y_sq = a->y_sq; k.v_out = a->v_out;
  k.y16 = reinterpret_cast<__bf16*>(a->y_bf16_copy);
  k.hfac = a->hfac_out;
  if ((unsigned)a->form > 3u) return BNN_ERR_ENUM;
  if (a->hfac_out && a->form == BNN_FORM_GEMM) return BNN_ERR_ENUM;
  if (a->y_bf16_copy && (a->y_dtype != BNN_F32 || a->form == BNN_FORM_GEMM)) return BNN_ERR_ENUM;
  k.w_frag = reinterpret_cast<const float4*>(a->w_frag);
  k.ws = a->want_kl ? reinterpret_cast<float4*>(a->workspace) : nullptr;
  k.S = a->n_samples; k.B = a->batch; k.K = a->in_features; k.N = a->out_features;
  k.eps_mode = a->eps_mode; k.want_kl = a->want_kl ? 1 : 0; k.relu = a->relu ? 1 : 0; k.y_bf16 = ybf;
  k.k0 = (uint32_t)a->seed; k.k1 = (uint32_t)(a->seed >> 32);
  k.layer_id = a->layer_id; k.sample_offset = a->sample_offset; k.sample_counter = a->sample_counter;
#ifdef BNN_STAMPS
  {
    const char* v = getenv("BNN_HIP_DBG_PTR");
    k.dbg = v ? reinterpret_cast<unsigned long long*>(strtoull(v, nullptr, 0)) : nullptr;
  }
#endif
  return BNN_OK;
}

// Launch geometry of K3: a pure function of the shape and of what the arguments allow (bnn_lr_plan exports it).
struct LrPlan {
  int form;         // BNN_FORM_TILE (K3a) or BNN_FORM_GEMM (K3b)
  int R, MT, nw;    // K3a: k-range classes, 16-row batch tiles per block, waves;  K3b: nw = 4 or 8
  long total;       // blocks
  size_t lds;
  int ksl, nst;     // K3s: slices per unit, k-steps per slice
  int es;           // K3s: samples per unit (shared input), 1 otherwise
};

// K3s over ONE input for all samples (x_per_sample == 0, 2 .. kLrsMaxShared samples): the unit's two products are made once
static bool lr_shared_input(const bnn_lr_fwd_args* a) {
  return a->x_per_sample == 0 && a->n_samples > 1 && a->n_samples <= kLrsMaxShared && a->sample_group == 0;
}

static size_t lr_ticket_bytes(long units) { return (((size_t)units * 4 + 255) / 256) * 256; }
static long lr_split_units(int n_samples, int batch, int out_features) {
  return (long)((out_features + 31) / 32) * n_samples * ((batch + 127) / 128);
}
extern "C" size_t bnn_lr_split_scratch_zero_bytes(int32_t n_samples, int32_t batch, int32_t out_features) {
  if (n_samples <= 0 || batch <= 0 || out_features <= 0) return 0;
  return lr_ticket_bytes(lr_split_units(n_samples, batch, out_features));
}
extern "C" size_t bnn_lr_split_scratch_bytes(int32_t n_samples, int32_t batch, int32_t out_features) {
  if (n_samples <= 0 || batch <= 0 || out_features <= 0) return 0;
  const long units = lr_split_units(n_samples, batch, out_features);
  return lr_ticket_bytes(units) + (size_t)units * kLrsMaxSlices * (8 * 4 * 64 * 16);
}

static int lr_kslice_plan(const bnn_lr_fwd_args* a, LrPlan& pl);

static int lr_plan(const bnn_lr_fwd_args* a, LrPlan& pl) {
  const int K = a->in_features, N = a->out_features, mbs = (a->batch + 127) / 128;
  const long gemm_blocks = (long)((N + 63) / 64) * a->n_samples * mbs;
  pl.es = 1;
  // all samples on one input: the K-sliced form makes the products once -- ahead of the block-GEMM form, which would
  // make them per sample
  if (a->math != BNN_MATH_BF16X3 && lr_shared_input(a) && a->form == BNN_FORM_AUTO && lr_kslice_plan(a, pl) == BNN_OK) return BNN_OK;
  // K3b needs bf16 x AND x^2 streams; the saved variance (v_out) is a K3a epilogue
  const bool x3 = a->math == BNN_MATH_BF16X3;             // (lr_fill has checked what that mode needs)
  const bool can = !a->v_out && !a->y_bf16_copy && !a->hfac_out && (a->math == BNN_MATH_BF16 || x3) && a->x_dtype == BNN_BF16 && a->x_sq && (K % 8 == 0) && K >= 8 &&
                   !(reinterpret_cast<uintptr_t>(a->x_sq) & 15);
  if (x3 && !(can && a->form != BNN_FORM_TILE && a->form != BNN_FORM_GEMM_KSLICE)) return BNN_ERR_ENUM;
  // a->form is a preference: the block-GEMM form is taken only when the arguments allow it
  // (over prepared fragments the block form pays from 7 samples on: 1200 x 1200, 7 samples 29 us + a 6 us prepare launch against
  // K3a's 47 (its third round of blocks); 10 samples 29 against 49; 16 samples 33 against 80 -- 5 / 6 samples: 28 + 6 against 33 --
  // tools/lr_mid_sweep.py, profiles/r03_lr_mid_sweep.log, r04_lr_mid_k3s.log)
  if (can && (a->form == BNN_FORM_GEMM || x3 ||
              (a->form == BNN_FORM_AUTO && (gemm_blocks >= 300 || (a->w_frag && a->n_samples >= 7 && gemm_blocks >= 130))))) {
    pl.form = BNN_FORM_GEMM;
    pl.R = 1; pl.MT = 8;
    // prepared fragments, wide layer: 8 waves share each x / x^2 tile (twice the MFMA work per LDS-DMA round trip)
    // ... 16 waves on a layer of >= 512 features: the x / x^2 tiles serve 256 features, a quarter less L2 traffic per MFMA
    pl.nw = (a->w_frag && N >= 512) ? 16 : (a->w_frag && N >= 128) ? 8 : 4;
    // ... but few samples want the chip covered first: the narrowest block whose launch is still one round of <= 256 blocks
    // (measured at 8 / 16 / 24 / 32 samples: 4 / 8 / 8 / 16 waves -- 152 / 160 / 240 / 160 blocks -- are the fastest)
    if (a->w_frag && N >= 128) {
      for (int w = 4; w < pl.nw; w *= 2)
        if ((long)((N + 16 * w - 1) / (16 * w)) * a->n_samples * mbs <= 256) { pl.nw = w; break; }
    }
#ifdef BNN_TUNE
    if (const char* v = getenv("BNN_TUNE_LRNW")) { const int f = atoi(v); if (a->w_frag && (f == 4 || f == 8 || f == 16)) pl.nw = f; }
#endif
    pl.total = (long)((N + 16 * pl.nw - 1) / (16 * pl.nw)) * a->n_samples * mbs;
    pl.lds = (size_t)(x3 ? kLrX3Depth + 1 : a->w_frag ? 3 : 2) * (x3 ? 3 : 2) * 8 * 64 * 16 + (size_t)pl.nw * 16 * sizeof(float);   // static: the ring of x / x^2 (/ x lo) tiles + biases
    return BNN_OK;
  }
  // K3s: few samples on a wide layer -- 32-feature groups x K slices, so that ~150-300 blocks exist and each pulls
  // whole parameter lines and a slice of x through its CU (see lr_fwd_kslice_kernel)
  pl.ksl = 1; pl.nst = 0;
  if (lr_kslice_plan(a, pl) == BNN_OK) return BNN_OK;
  if (a->form == BNN_FORM_GEMM_KSLICE) return BNN_ERR_ENUM;  // asked for, not possible
  int R = 1;
  while (R < 4 && (long)((N + 16 / R - 1) / (16 / R)) * a->n_samples * mbs < 120) R *= 2;
#ifdef BNN_TUNE
  if (const char* v = getenv("BNN_TUNE_LRR")) { const int f = atoi(v); if (f == 1 || f == 2 || f == 4) R = f; }
#endif
  const int F = 16 / R;
  const int ssteps = (K + 32 * R - 1) / (32 * R);
  // a narrow layer in few samples: 32-row blocks (MT = 2), so that more than a handful of blocks exist
  // and each ingests a quarter of x; its weights are small enough that re-reading them per block is free
  const int MT = ((long)((N + F - 1) / F) * a->n_samples * mbs < 64 && N <= 64 && a->batch > 32) ? 2 : 8;
  const int max_nw = MT == 2 ? 12 : 8;
  int spw = 1;
  while ((ssteps + spw - 1) / spw > max_nw) ++spw;
  int nw = (ssteps + spw - 1) / spw;
  nw = nw < 1 ? 1 : nw;
  // a CU has 4 SIMDs: keep the waves of a block a multiple of 4 so no SIMD carries one more than the others
  // (as many as there are k-steps, up to the limit) -- see tile_plan in bbb_linear.hip
  if (ssteps >= 4) nw = (ssteps >= max_nw ? max_nw : ssteps) & ~3;
  // the epilogue gives every thread at most 2 output items (batch row x 4 features): a short k range
  // must not leave the block with fewer threads than that needs (extra waves own no k-step and
  // contribute zero slabs)
  const int mt = a->batch >= 16 * MT ? MT : (a->batch + 15) / 16;
  const int need = (mt * 16 * (F / 4) + 127) / 128;
  if (nw < need) nw = need;
  pl.form = BNN_FORM_TILE;
  pl.R = R; pl.MT = MT; pl.nw = nw;
  pl.total = (long)((N + F - 1) / F) * a->n_samples * ((a->batch + 16 * MT - 1) / (16 * MT));
  pl.lds = ((size_t)nw * MT * 64 * 4 + 16 + 3 * nw) * sizeof(float);
  return BNN_OK;
}

// K3s geometry; BNN_OK when the K-sliced form applies (pl filled), BNN_ERR_ENUM otherwise
static int lr_kslice_plan(const bnn_lr_fwd_args* a, LrPlan& pl) {
  const int K = a->in_features, N = a->out_features;
  const bool shared = lr_shared_input(a);
  const long units = lr_split_units(shared ? 1 : a->n_samples, a->batch, N);
  const int ksteps = (K + 31) / 32;
  const bool ok = a->math == BNN_MATH_BF16 && (K % 8 == 0) && K >= 64 && (N % 4 == 0) && N >= 64 &&
                  a->split_scratch && !(reinterpret_cast<uintptr_t>(a->split_scratch) & 15) &&
                  a->split_scratch_bytes >= bnn_lr_split_scratch_bytes(shared ? 1 : a->n_samples, a->batch, N) &&
                  !(reinterpret_cast<uintptr_t>(a->w_mu) & 15) && !(reinterpret_cast<uintptr_t>(a->w_rho) & 15) &&
                  (a->form == BNN_FORM_AUTO || a->form == BNN_FORM_GEMM_KSLICE);
  // One round of blocks (<= 256) -- except where the tile form itself would need a second round for this launch (per-sample
  // inputs, more than 256 of its 16-feature blocks): two K3s blocks fit a CU, so up to 512 are still all resident (1200 x 1200 at
  // 4 samples: 456 blocks 25 us, K3a's 300 blocks 32 us -- at 3 samples K3a's 225 blocks are one round, 17.8 against 22.1:
  // tools/lr_mid_sweep.py, profiles/r04_lr_mid_k3s.log).  The shared-input form polls for its unit's slices: one round only.
  const long tile_blocks = (long)((N + 15) / 16) * a->n_samples * ((a->batch + 127) / 128);
  long max_units = 160, max_blocks = (!shared && tile_blocks > 256) ? 512 : 256;
#if defined(BNN_TUNE)
  if (const char* v = getenv("BNN_TUNE_LRS_MAXUNITS")) max_units = atol(v);
  if (const char* v = getenv("BNN_TUNE_LRS_MAXBLOCKS")) max_blocks = atol(v);
#endif
  if (!ok || units > max_units) return BNN_ERR_ENUM;
  int ksl = (int)(160 / units);
  ksl = ksl < 1 ? 1 : ksl > kLrsMaxSlices ? kLrsMaxSlices : ksl;
#if defined(BNN_TUNE) || defined(BNN_STAMPS)
  if (const char* v = getenv("BNN_TUNE_LRKSL")) { const int f = atoi(v); if (f >= 1 && f <= kLrsMaxSlices) ksl = f; }
#endif
  while (ksl < kLrsMaxSlices && (ksteps + ksl - 1) / ksl > kLrsMaxSteps) ++ksl;
  const int nst = (ksteps + ksl - 1) / ksl;
  ksl = (ksteps + nst - 1) / nst;                          // no empty slices
  if (nst > kLrsMaxSteps || units * ksl > max_blocks) return BNN_ERR_ENUM;   // one round of blocks: slices that queue behind each other gain nothing
  pl.form = BNN_FORM_GEMM_KSLICE;
  pl.R = 1; pl.MT = 8; pl.nw = 8;
  pl.ksl = ksl; pl.nst = nst;
  pl.es = shared ? a->n_samples : 1;
  pl.total = units * ksl;
  pl.lds = (size_t)kLrsMaxSteps * 4096 + (kLrsMaxShared * 32 + 24 + 1) * sizeof(float);
  return BNN_OK;
}

extern "C" int bnn_lr_plan(const bnn_lr_fwd_args* a, bnn_plan* out) {
  if (!out) return BNN_ERR_NULL;
  LrK k;
  int rc = lr_fill(a, k);
  if (rc != BNN_OK) return rc;
  LrPlan pl{};
  rc = lr_plan(a, pl);
  if (rc != BNN_OK) return rc;
  out->form = pl.form;
  out->k_classes = pl.R;
  out->waves = pl.nw;
  out->batch_rows = 16 * pl.MT;
  out->k_slices = pl.form == BNN_FORM_GEMM_KSLICE ? pl.ksl : 1;
  out->blocks = (int32_t)pl.total;
  out->lds_bytes = (int32_t)pl.lds;
  out->features_per_block = pl.form == BNN_FORM_TILE ? 16 / pl.R : pl.form == BNN_FORM_GEMM_KSLICE ? 32 : 16 * pl.nw;
  return BNN_OK;
}

extern "C" int bnn_lr_linear_fwd(const bnn_lr_fwd_args* a, void* stream_) {
  LrK k;
  int rc = lr_fill(a, k);
  if (rc != BNN_OK) return rc;
  LrPlan pl{};
  rc = lr_plan(a, pl);
  if (rc != BNN_OK) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int K = a->in_features, N = a->out_features;
  hipError_t err = hipSuccess;
  dim3 grid((unsigned)(((pl.total + 7) / 8) * 8)), block(pl.nw * 64);
  if (a->rider) {
    // the narrow layer's preparation rides on a K3s launch (extra blocks behind the padded main grid); any other form
    // launches it ahead of itself
    const bnn_lr_rider* rd = a->rider;
    const int rsteps = (rd->in_features + 31) / 32;
    const int entries = (int)(rd->kl_workspace_bytes / 16) - 1;
    int spb = kLrRiderSteps, nb = (rsteps + spb - 1) / spb;
    if (pl.form == BNN_FORM_GEMM_KSLICE && rd->out_features <= 16 && nb <= entries) {
      k.rd_w_mu = rd->w_mu; k.rd_w_rho = rd->w_rho; k.rd_b_mu = rd->b_mu; k.rd_b_rho = rd->b_rho;
      k.rd_frag = reinterpret_cast<float4*>(rd->w_frag); k.rd_ws = reinterpret_cast<float4*>(rd->kl_workspace);
      k.rd_K = rd->in_features; k.rd_N = rd->out_features; k.rd_blocks = nb; k.rd_spb = spb;
      k.main_blocks = (int)grid.x;
      grid.x += (unsigned)nb;
    } else {
      rc = bnn_lr_prepare(rd->w_mu, rd->w_rho, rd->b_mu, rd->b_rho, rd->in_features, rd->out_features, rd->w_frag, rd->w_frag_bytes,
                          rd->kl_workspace, rd->kl_workspace_bytes, stream_);
      if (rc != BNN_OK) return rc;
    }
  }
  if (pl.form == BNN_FORM_GEMM_KSLICE) {
    char* base = reinterpret_cast<char*>(a->split_scratch);
    k.ksl = pl.ksl; k.nst = pl.nst; k.es = pl.es;
    k.ks_ticket = reinterpret_cast<uint32_t*>(base);
    k.ks_part = reinterpret_cast<float4*>(base + lr_ticket_bytes(pl.total / pl.ksl));
#define BNN_LRS(XF)                                                                                             \
  do {                                                                                                          \
    if (pl.nst <= 4) hipLaunchKernelGGL((lr_fwd_kslice_kernel<XF, 4>), grid, block, 0, stream, k);              \
    else if (pl.nst <= 8) hipLaunchKernelGGL((lr_fwd_kslice_kernel<XF, 8>), grid, block, 0, stream, k);         \
    else if (pl.nst <= 10) hipLaunchKernelGGL((lr_fwd_kslice_kernel<XF, 10>), grid, block, 0, stream, k);       \
    else hipLaunchKernelGGL((lr_fwd_kslice_kernel<XF, kLrsMaxSteps>), grid, block, 0, stream, k);               \
  } while (0)
    if (a->x_dtype == BNN_F32) BNN_LRS(true); else BNN_LRS(false);
#undef BNN_LRS
  } else if (pl.form == BNN_FORM_GEMM) {
#ifdef BNN_TUNE
    k.tune = getenv("BNN_TUNE_LRFLAGS") ? atoi(getenv("BNN_TUNE_LRFLAGS")) : 0;
#endif
    {
      // of an XCD's 4 MiB L2 for the prepared weights of a class's feature groups: each unit streams 614 KB of x and
      // x^2 tiles through the same L2 (measured on 1200 x 1200, 256 samples: 1300 KB 290 us, 2560 KB 322, one class 362)
      size_t l2_budget = 1300 * 1024;
#ifdef BNN_TUNE
      if (const char* v = getenv("BNN_TUNE_L2KB")) l2_budget = (size_t)atol(v) * 1024;
#endif
      const int feats = 16 * pl.nw;
      k.xc = xcd2d_make((N + feats - 1) / feats, a->n_samples * ((a->batch + 127) / 128), 1,
                        (size_t)feats * K * (a->math == BNN_MATH_BF16X3 ? 6 : a->w_frag ? 4 : 8), l2_budget);
    }
    if (a->w_frag) {
      if (reinterpret_cast<uintptr_t>(a->w_frag) & 15) return BNN_ERR_ALIGN;
      if (a->math == BNN_MATH_BF16X3) {
        if (pl.nw == 16) hipLaunchKernelGGL((lr_fwd_gemm_kernel<16, true, kLrX3Depth, true>), grid, block, 0, stream, k);
        else if (pl.nw == 8) hipLaunchKernelGGL((lr_fwd_gemm_kernel<8, true, kLrX3Depth, true>), grid, block, 0, stream, k);
        else hipLaunchKernelGGL((lr_fwd_gemm_kernel<4, true, kLrX3Depth, true>), grid, block, 0, stream, k);
      } else if (pl.nw == 16) hipLaunchKernelGGL((lr_fwd_gemm_kernel<16, true, 2>), grid, block, 0, stream, k);
      else if (pl.nw == 8) hipLaunchKernelGGL((lr_fwd_gemm_kernel<8, true, 2>), grid, block, 0, stream, k);
      else hipLaunchKernelGGL((lr_fwd_gemm_kernel<4, true, 2>), grid, block, 0, stream, k);
    } else {
      hipLaunchKernelGGL((lr_fwd_gemm_kernel<4, false, 1>), grid, block, 0, stream, k);
    }
  } else {
    const int R = pl.R, MT = pl.MT;
    const size_t lds = pl.lds;
#define BNN_LR(MATH, XDT, RR, MM)                                                                       \
  do {                                                                                                  \
    if (lds > 64 * 1024)                                                                                \
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(lr_fwd_kernel<MATH, XDT, RR, MM>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
    if (err == hipSuccess) hipLaunchKernelGGL((lr_fwd_kernel<MATH, XDT, RR, MM>), grid, block, lds, stream, k); \
  } while (0)
#define BNN_LR_M(MATH, XDT, RR)                  \
  do {                                           \
    if (MT == 2) BNN_LR(MATH, XDT, RR, 2);       \
    else BNN_LR(MATH, XDT, RR, 8);               \
  } while (0)
#define BNN_LR_R(MATH, XDT)                      \
  do {                                           \
    if (R == 1) BNN_LR_M(MATH, XDT, 1);          \
    else if (R == 2) BNN_LR_M(MATH, XDT, 2);     \
    else BNN_LR_M(MATH, XDT, 4);                 \
  } while (0)
    if (a->math == BNN_MATH_BF16) {
      if (a->x_dtype == BNN_F32) BNN_LR_R(BNN_MATH_BF16, BNN_F32); else BNN_LR_R(BNN_MATH_BF16, BNN_BF16);
    } else {
      if (a->x_dtype == BNN_F32) BNN_LR_R(BNN_MATH_F32, BNN_F32); else BNN_LR_R(BNN_MATH_F32, BNN_BF16);
    }
#undef BNN_LR
#undef BNN_LR_M
#undef BNN_LR_R
    if (err != hipSuccess) return (int)err;
  }
  err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  if (a->kl_out) {
    hipLaunchKernelGGL(lr_layer_kl_kernel, dim3(1), dim3(256), 0, stream, k.ws, K, N, a->sigma_p, a->b_mu, a->b_rho,
                       a->kl_out);
    err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  return BNN_OK;
}

extern "C" size_t bnn_bbb_final_scratch_bytes(int32_t n_samples);   // bbb_linear.hip: the same scratch layout serves K3r
extern "C" int bnn_loss_tail_(const bnn_finalize_args* f, void* stream_);   // reduce.hip

// Last LR layer + ELBO finalize: ONE launch (K3r) for a few-sample evaluation with a narrow output layer, else
// bnn_lr_linear_fwd followed by bnn_elbo_finalize.
#ifndef BNN_LR_ROWS_PREPARED_MAX
#define BNN_LR_ROWS_PREPARED_MAX 4096   // build knob (A/B): 16 = the row-split form for few pairs only, as until round 4
#endif
#ifndef BNN_LR_ROWS_RT_MAX
#define BNN_LR_ROWS_RT_MAX 2            // build knob (A/B): 1 = one 16-row tile per block whatever the launch.  The LR launch group of
                                        // 256 minibatches, one box, alternating: 573-576 us at 1, 552-557 at 2, 559-563 at 4 (profiles/r04_lr_rows_ab.log)
#endif
static_assert(BNN_LR_ROWS_RT_MAX >= 1 && BNN_LR_ROWS_RT_MAX <= bnn::kRowsMaxTiles, "LrRows.rt is at most kRowsMaxTiles");
static constexpr int kLrRowsTicketMaxSamples = 64;   // K3r: up to here the last sample's last block folds the sums
extern "C" int bnn_elbo_sums_(const bnn_finalize_args* f, void* stream_);
extern "C" int bnn_lr_final_fwd(const bnn_lr_fwd_args* a, const bnn_finalize_args* f, void* stream_) {
  LrK k;
  int rc = lr_fill(a, k);
  if (rc != BNN_OK) return rc;
  LrFin fp;
  rc = make_fin(f, fp.k, fp.c);
  if (rc != BNN_OK) return rc;
  rc = check_fin_loss(f);
  if (rc != BNN_OK) return rc;
  const int S = a->n_samples, B = a->batch, N = a->out_features, K = a->in_features, nl = f->n_layers;
  // (S: up to 16 pairs when every row block has to park the layer itself; over PREPARED fragments (bnn_lr_prepare[_many], the
  // rider) any launch group -- its grid is S x (ceil(B / 16) + 1) small blocks that need no LDS)
  const bool prepared = a->w_frag && a->want_kl && a->workspace && !(reinterpret_cast<uintptr_t>(a->w_frag) & 15);
  const bool rows = a->math == BNN_MATH_BF16 && a->x_dtype == BNN_BF16 && a->y_dtype == BNN_F32 && N <= 16 && B <= 128 && S <= (prepared ? BNN_LR_ROWS_PREPARED_MAX : 16) &&
                    (K % 8) == 0 && K <= 2048 && !(reinterpret_cast<uintptr_t>(a->x) & 15) && a->eps_mode == BNN_EPS_PHILOX &&
                    !a->eps_act_dump && !a->eps_b_dump && !a->y_sq && !a->y_bf16_copy && !a->hfac_out && !a->kl_out && a->form == BNN_FORM_AUTO &&
                    f->local_reparam && nl >= 1 && nl <= 8 && f->n_samples == S && f->classes == N && f->batch == B &&
                    f->logits == a->y && f->nll && f->kl && f->layer_in[nl - 1] == K && f->layer_out[nl - 1] == N &&
                    f->scratch && f->scratch_bytes >= bnn_bbb_final_scratch_bytes(S) &&
                    !(reinterpret_cast<uintptr_t>(f->scratch) & 15) && (S == 1 || S > kLrRowsTicketMaxSamples || f->ticket) &&
                    (N % 4 != 0 || !(reinterpret_cast<uintptr_t>(a->y) & 15));
  if (!rows) {
    rc = bnn_lr_linear_fwd(a, stream_);
    if (rc == BNN_OK) rc = bnn_elbo_finalize(f, stream_);
    return rc != BNN_OK ? rc : bnn_loss_tail_(f, stream_);
  }
  const FinLoss tr = make_fin_loss(f);
  LrRows r;
  r.x = reinterpret_cast<const __bf16*>(a->x);
  r.x_sstride = k.x_sstride; r.xg = k.xg;
  r.w_mu = a->w_mu; r.w_rho = a->w_rho; r.b_mu = a->b_mu; r.b_rho = a->b_rho;
  r.y = reinterpret_cast<float*>(a->y);
  r.v_out = a->v_out;
  r.S = S; r.B = B; r.K = K; r.N = N; r.relu = a->relu ? 1 : 0;
  r.k0 = k.k0; r.k1 = k.k1; r.layer_id = k.layer_id; r.sample_offset = k.sample_offset; r.sample_counter = k.sample_counter;
  r.sgrp = k.sgrp; r.sgrp_stride = k.sgrp_stride;
  // prepared operands (bnn_lr_prepare, or the rider of the previous layer's launch) with the KL sums that came with them
  r.w_frag = (a->w_frag && a->want_kl && a->workspace && !(reinterpret_cast<uintptr_t>(a->w_frag) & 15)) ? reinterpret_cast<const float4*>(a->w_frag) : nullptr;
  r.ws_own = r.w_frag ? reinterpret_cast<const float4*>(a->workspace) : nullptr;
  char* base = reinterpret_cast<char*>(f->scratch);
  r.tickets = reinterpret_cast<uint32_t*>(base);
  r.parts = reinterpret_cast<float*>(base + (((size_t)S * 4 + 255) / 256) * 256);
  // (many samples: the sums of the per-sample scalars and the counter advance by a one-block follow-up kernel, as in bnn_bbb_final_fwd)
  const bool rows_tail = S > kLrRowsTicketMaxSamples && !tr.out4;
  fp.sums = f->sums;
  fp.ticket = rows_tail ? nullptr : f->ticket;
  const int RB = (B + 15) / 16;
  // many pairs over prepared fragments: two 16-row tiles per block (two blocks of ~200 registers fit a CU: S x 9 one-tile blocks
  // of a 256-pair launch ran as five rounds); few pairs: one tile per block, the shortest chain
  r.rt = (r.w_frag && (long)S * (RB + 1) > 512) ? (RB < BNN_LR_ROWS_RT_MAX ? RB : BNN_LR_ROWS_RT_MAX) : 1;
  const int NBLK = (RB + r.rt - 1) / r.rt;
  const size_t lds = r.w_frag ? 0 : (size_t)2 * ((K + 31) / 32) * 4 * 16 * 8 * 2;       // bf16 M and sigma^2 fragments of the whole layer (parked by the block itself)
  if (lds > 64 * 1024) {
    const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(lr_final_rows_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e0 != hipSuccess) return (int)e0;
  }
  if (r.rt > 1)
    hipLaunchKernelGGL(lr_final_rows_kernel<true>, dim3((unsigned)(S * (NBLK + 1))), dim3(256), lds, reinterpret_cast<hipStream_t>(stream_), r, fp, tr);
  else
    hipLaunchKernelGGL(lr_final_rows_kernel<false>, dim3((unsigned)(S * (NBLK + 1))), dim3(256), lds, reinterpret_cast<hipStream_t>(stream_), r, fp, tr);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  return rows_tail ? bnn_elbo_sums_(f, stream_) : (int)BNN_OK;
}
