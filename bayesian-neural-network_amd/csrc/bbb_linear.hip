// K1  bbb_linear_fwd — BayesianLinear.forward (reference networks.py:73-88) for all locally
// owned MC samples of one layer in one launch.
//
// Work decomposition (gfx950, wave64):
//   grid  = (ceil(out/16), n_samples, ceil(batch/128))
//   block = NW waves; every wave owns the SAME 16 output features and a strided set of
//           32-deep k-steps (wave w takes k-steps w, w+NW, ...), so a block streams whole
//           (mu, rho) rows with 128-byte segments per lane quad.
//   lane (r = lane&15, q = lane>>4) of a wave holds, per k-step, the 8 consecutive weights
//           W[n0+r][32t + 8q .. +7] — exactly the A-operand fragment of
//           v_mfma_f32_16x16x32_bf16 — so mu/rho are read once as two 16-byte loads each,
//           eps is generated in that layout (2 Philox calls = 8 normals), w = mu + sigma*eps
//           is formed in registers, folded into the log-prob partial sums in fp32, rounded
//           to bf16 and fed to the matrix core.  w never exists in memory.
//   B operand = x[16m + r][32t + 8q .. +7] for the 8 batch tiles m of the block's 128 rows.
//   D[row = out feature 4q+i][col = batch row r]: a lane ends with 4 consecutive output
//           features of one batch row -> one 16-byte (fp32) / 8-byte (bf16) store.
//   The NW partial accumulators meet in LDS (one 8 KiB slab per wave), then bias + ReLU +
//   down-conversion run as the epilogue.  No cross-block reduction, no atomics: results are
//   bitwise reproducible and independent of the grid.
#include "bnn_device.h"
#include "../../include/bnn_hip.h"

namespace bnn {

struct BbbK {
  const void* x;
  long x_sstride;   // elements between samples of x (0: shared)
  const float* w_mu;
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  const float* eps_w;
  const float* eps_b;
  float* eps_w_dump;
  float* eps_b_dump;
  void* y;
  float* partial;   // [S][T][4]
  int S, B, K, N;
  int eps_mode, prior_kind, want_stats, relu, y_bf16;
  uint32_t k0, k1, layer_id, sample_offset;
  const uint32_t* sample_counter;
  float inv2var1, c1, inv2var2, c2, pi;   // mixture: log N(w;0,s_i) = c_i - w^2 * inv2var_i
};

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

// x fragment for one batch tile: 8 consecutive k of row `row`.
template <int XDT, bool ALIGNED>
__device__ __forceinline__ void load_x8(const void* __restrict__ xbase, long row_off, int valid, float v[8],
                                        bf16x8& vb) {
  if (XDT == BNN_F32) {
    load8<ALIGNED>(reinterpret_cast<const float*>(xbase) + row_off, valid, v);
  } else {
    const __bf16* p = reinterpret_cast<const __bf16*>(xbase) + row_off;
    if (ALIGNED) {
      if (valid > 0) {
        vb = *reinterpret_cast<const bf16x8*>(p);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) vb[j] = (__bf16)0.0f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) vb[j] = (j < valid) ? p[j] : (__bf16)0.0f;
    }
  }
}

template <int MATH, int XDT, bool ALIGNED>
__global__ __launch_bounds__(512) void bbb_linear_fwd_kernel(const BbbK p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nt = blockIdx.x, s = blockIdx.y, mb = blockIdx.z;
  const int K = p.K, N = p.N, B = p.B;
  const int n = nt * 16 + r;
  const bool n_ok = n < N;
  const int m0 = mb * 128;
  const int mtiles = min(8, (B - m0 + 15) >> 4);
  const uint32_t gs = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u) + (uint32_t)s;
  const bool do_stats = p.want_stats && mb == 0;
  const bool do_ls = do_stats && s == 0;
  const bool do_dump = mb == 0;
  const int gpr = (K + 3) >> 2;                    // eps groups per weight row
  const int ksteps = (K + 31) >> 5;
  const char* xs = reinterpret_cast<const char*>(p.x) +
                   (size_t)s * (size_t)p.x_sstride * (XDT == BNN_F32 ? 4 : 2);

  f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s_e2 = 0.f, s_a = 0.f, s_ls = 0.f;

  for (int t = wave; t < ksteps; t += nw) {
    const int k = t * 32 + q * 8;
    const int valid = n_ok ? min(8, K - k) : 0;   // <= 0: nothing of this lane's 8 is real
    const size_t woff = (size_t)n * K + k;

    float mu[8], sg[8], e[8], w[8];
    load8<ALIGNED>(p.w_mu + woff, valid, mu);
    load8<ALIGNED>(p.w_rho + woff, valid, sg);
    if (p.eps_mode == BNN_EPS_PHILOX) {
      const uint32_t g = (uint32_t)n * (uint32_t)gpr + (uint32_t)(k >> 2);
      philox_normal4(g, gs, p.layer_id * 4u, p.k0, p.k1, e);
      philox_normal4(g + 1u, gs, p.layer_id * 4u, p.k0, p.k1, e + 4);
    } else if (p.eps_mode == BNN_EPS_MEMORY) {
      load8<ALIGNED>(p.eps_w + ((size_t)s * N + n) * K + k, valid, e);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = 0.f;
    }
    if (p.eps_w_dump && do_dump) store8<ALIGNED>(p.eps_w_dump + ((size_t)s * N + n) * K + k, valid, e);

#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = j < valid;
      const float sig = softplus(sg[j]);
      const float wj = ok ? __builtin_fmaf(sig, e[j], mu[j]) : 0.f;
      w[j] = wj;
      if (do_stats) {
        s_e2 += ok ? e[j] * e[j] : 0.f;
        if (p.prior_kind == BNN_PRIOR_GAUSS) {
          s_a = __builtin_fmaf(wj, wj, s_a);
        } else {
          const float w2 = wj * wj;
          const float p1 = fast_exp(__builtin_fmaf(-w2, p.inv2var1, p.c1));
          const float p2 = fast_exp(__builtin_fmaf(-w2, p.inv2var2, p.c2));
          const float lp = fast_log(p.pi * p1 + (1.0f - p.pi) * p2);
          s_a += ok ? lp : 0.f;
        }
        if (do_ls) s_ls += ok ? fast_log(sig) : 0.f;
      }
    }

    bf16x8 wa;
    if (MATH == BNN_MATH_BF16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) wa[j] = (__bf16)w[j];
    }

#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < mtiles) {
        const int row = m0 + m * 16 + r;
        const int xvalid = (row < B) ? min(8, K - k) : 0;
        float xv[8];
        bf16x8 xb;
        load_x8<XDT, ALIGNED>(xs, (long)row * K + k, xvalid, xv, xb);
        if (MATH == BNN_MATH_BF16) {
          if (XDT == BNN_F32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xb[j] = (__bf16)xv[j];
          }
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[m], 0, 0, 0);
        } else {
          if (XDT == BNN_BF16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = (float)xb[j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], xv[j], acc[m], 0, 0, 0);
        }
      }
    }
  }

  // ---- bias of the tile's 16 features: wave 0, lanes 0..15 (q == 0)
  float* lds_bias = lds + (size_t)nw * 8 * 64 * 4;   // 16 floats
  float* lds_red = lds_bias + 16;                    // 3 * nw floats
  if (wave == 0 && q == 0) {
    float b = 0.f;
    if (n_ok) {
      const float bmu = p.b_mu[n], sig = softplus(p.b_rho[n]);
      float e = 0.f;
      if (p.eps_mode == BNN_EPS_PHILOX) {
        float e4[4];
        philox_normal4((uint32_t)(n >> 2), gs, p.layer_id * 4u + 1u, p.k0, p.k1, e4);
        e = (n & 3) == 0 ? e4[0] : (n & 3) == 1 ? e4[1] : (n & 3) == 2 ? e4[2] : e4[3];
      } else if (p.eps_mode == BNN_EPS_MEMORY) {
        e = p.eps_b[(size_t)s * N + n];
      }
      if (p.eps_b_dump && do_dump) p.eps_b_dump[(size_t)s * N + n] = e;
      b = __builtin_fmaf(sig, e, bmu);
      if (do_stats) {
        s_e2 += e * e;
        if (p.prior_kind == BNN_PRIOR_GAUSS) {
          s_a = __builtin_fmaf(b, b, s_a);
        } else {
          const float w2 = b * b;
          const float p1 = fast_exp(__builtin_fmaf(-w2, p.inv2var1, p.c1));
          const float p2 = fast_exp(__builtin_fmaf(-w2, p.inv2var2, p.c2));
          s_a += fast_log(p.pi * p1 + (1.0f - p.pi) * p2);
        }
        if (do_ls) s_ls += fast_log(sig);
      }
    }
    lds_bias[r] = b;
  }

  // ---- cross-wave reduction of the k-split accumulators through LDS
  f32x4* slab = reinterpret_cast<f32x4*>(lds);
#pragma unroll
  for (int m = 0; m < 8; ++m)
    if (m < mtiles) slab[(wave * 8 + m) * 64 + lane] = acc[m];
  if (do_stats) {
    const float a = wave_sum(s_e2), b = wave_sum(s_a), c = wave_sum(s_ls);
    if (lane == 0) {
      lds_red[wave * 3 + 0] = a;
      lds_red[wave * 3 + 1] = b;
      lds_red[wave * 3 + 2] = c;
    }
  }
  __syncthreads();

  if (do_stats && threadIdx.x == 0) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int wv = 0; wv < nw; ++wv) {
      a += lds_red[wv * 3 + 0];
      b += lds_red[wv * 3 + 1];
      c += lds_red[wv * 3 + 2];
    }
    float4* out = reinterpret_cast<float4*>(p.partial) + ((size_t)s * gridDim.x + nt);
    *out = make_float4(a, b, c, 0.f);
  }

  // ---- epilogue: sum slabs, + bias, ReLU, convert, store 4 consecutive features per item
  const bool vec_ok = (N & 3) == 0;
  for (int item = threadIdx.x; item < mtiles * 64; item += blockDim.x) {
    const int m = item >> 6, l = item & 63;
    f32x4 v = slab[m * 64 + l];
    for (int wv = 1; wv < nw; ++wv) v += slab[(wv * 8 + m) * 64 + l];
    const int brow = m0 + m * 16 + (l & 15);
    const int f0 = (l >> 4) * 4;
    const int nb = nt * 16 + f0;
    if (brow >= B || nb >= N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o = v[i] + lds_bias[f0 + i];
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

// Per-layer reduction of the stats partials into the scalars BayesianLinear stores
// (networks.py:82-83).  One block per sample.
__global__ void bbb_layer_scalars_kernel(const float* __restrict__ partial, int T, int K, int N, bnn_prior prior,
                                         float* __restrict__ log_prior, float* __restrict__ log_q) {
  __shared__ double scratch[16];
  const int s = blockIdx.x;
  double e2 = 0, a = 0, ls = 0;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(partial)[(size_t)s * T + t];
    const float4 v0 = reinterpret_cast<const float4*>(partial)[t];
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
  return (size_t)n_samples * (size_t)((out_features + 15) / 16) * 4 * sizeof(float);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int bnn_bbb_linear_fwd(const bnn_bbb_fwd_args* a, void* stream_) {
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_bbb_fwd_args)) return BNN_ERR_ABI;
  if (a->n_samples <= 0 || a->batch <= 0 || a->in_features <= 0 || a->out_features <= 0) return BNN_ERR_SHAPE;
  if (a->n_samples > 65535 || (a->batch + 127) / 128 > 65535) return BNN_ERR_SHAPE;
  if (!a->x || !a->w_mu || !a->w_rho || !a->b_mu || !a->b_rho || !a->y) return BNN_ERR_NULL;
  if ((unsigned)a->x_dtype > 1u || (unsigned)a->y_dtype > 1u || (unsigned)a->math > 1u || (unsigned)a->eps_mode > 2u ||
      (unsigned)a->prior.kind > 1u)
    return BNN_ERR_ENUM;
  if (a->eps_mode == BNN_EPS_MEMORY && (!a->eps_w || !a->eps_b)) return BNN_ERR_NULL;
  const int T = (a->out_features + 15) / 16;
  if (a->want_stats) {
    if (!a->workspace || a->workspace_bytes < bnn_bbb_linear_fwd_workspace_bytes(a->n_samples, a->out_features))
      return BNN_ERR_WORKSPACE;
    if (!aligned16(a->workspace)) return BNN_ERR_ALIGN;
  }
  if ((a->log_prior || a->log_q) && !a->want_stats) return BNN_ERR_WORKSPACE;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);

  BbbK k;
  k.x = a->x;
  k.x_sstride = a->x_per_sample ? (long)a->batch * a->in_features : 0;
  k.w_mu = a->w_mu; k.w_rho = a->w_rho; k.b_mu = a->b_mu; k.b_rho = a->b_rho;
  k.eps_w = a->eps_w; k.eps_b = a->eps_b; k.eps_w_dump = a->eps_w_dump; k.eps_b_dump = a->eps_b_dump;
  k.y = a->y;
  k.partial = reinterpret_cast<float*>(a->workspace);
  k.S = a->n_samples; k.B = a->batch; k.K = a->in_features; k.N = a->out_features;
  k.eps_mode = a->eps_mode; k.prior_kind = a->prior.kind; k.want_stats = a->want_stats ? 1 : 0;
  k.relu = a->relu ? 1 : 0; k.y_bf16 = a->y_dtype == BNN_BF16;
  k.k0 = (uint32_t)a->seed; k.k1 = (uint32_t)(a->seed >> 32);
  k.layer_id = a->layer_id; k.sample_offset = a->sample_offset; k.sample_counter = a->sample_counter;
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

  // 16-byte vector path: rows of 8 elements stay inside a row and every base is aligned.
  const int K = a->in_features;
  bool al = (K % 8 == 0) && aligned16(a->x) && aligned16(a->w_mu) && aligned16(a->w_rho);
  if (a->eps_mode == BNN_EPS_MEMORY) al = al && aligned16(a->eps_w);
  if (a->eps_w_dump) al = al && aligned16(a->eps_w_dump);
  const bool ybf = a->y_dtype == BNN_BF16;
  if ((a->out_features % 4 == 0) && (reinterpret_cast<uintptr_t>(a->y) & (ybf ? 7 : 15))) return BNN_ERR_ALIGN;

  const int ksteps = (K + 31) / 32;
  int nw = ksteps / 3;
  nw = nw < 1 ? 1 : (nw > 8 ? 8 : nw);
  const dim3 grid(T, a->n_samples, (a->batch + 127) / 128), block(nw * 64);
  const size_t lds = ((size_t)nw * 8 * 64 * 4 + 16 + 3 * nw) * sizeof(float);

#define BNN_LAUNCH(MATH, XDT, AL) \
  hipLaunchKernelGGL((bbb_linear_fwd_kernel<MATH, XDT, AL>), grid, block, lds, stream, k)
  const int xdt = a->x_dtype;
  if (a->math == BNN_MATH_BF16) {
    if (xdt == BNN_F32) { if (al) BNN_LAUNCH(BNN_MATH_BF16, BNN_F32, true); else BNN_LAUNCH(BNN_MATH_BF16, BNN_F32, false); }
    else                { if (al) BNN_LAUNCH(BNN_MATH_BF16, BNN_BF16, true); else BNN_LAUNCH(BNN_MATH_BF16, BNN_BF16, false); }
  } else {
    if (xdt == BNN_F32) { if (al) BNN_LAUNCH(BNN_MATH_F32, BNN_F32, true); else BNN_LAUNCH(BNN_MATH_F32, BNN_F32, false); }
    else                { if (al) BNN_LAUNCH(BNN_MATH_F32, BNN_BF16, true); else BNN_LAUNCH(BNN_MATH_F32, BNN_BF16, false); }
  }
#undef BNN_LAUNCH
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;

  if (a->log_prior || a->log_q) {
    hipLaunchKernelGGL(bbb_layer_scalars_kernel, dim3(a->n_samples), dim3(256), 0, stream, k.partial, T, K,
                       a->out_features, a->prior, a->log_prior, a->log_q);
    err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  return BNN_OK;
}
