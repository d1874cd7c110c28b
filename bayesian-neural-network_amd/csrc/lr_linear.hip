// K3  lr_linear_fwd — BayesianLinearLR.forward (reference networks.py:116-138) for all
// locally owned MC samples of one layer in one launch.
//
//   m = x . M,  v = x^2 . softplus(rho)^2,  y = m + sqrt(v)*eps_act + (b_mu + sigma_b*eps_b)
//
// Same decomposition as K1 (bbb_linear.hip): grid (ceil(out/16), n_samples, ceil(batch/128)),
// NW waves split the k-steps of one 16-feature tile, two accumulator sets (mean, variance)
// share every x fragment.  Weights are stored [in, out] (networks.py:95-96), so lane (r,q)
// gathers M[32t + 8q + j][n0 + r], j = 0..7: for each j the 16 lanes of a quad row read 64
// contiguous bytes.  sigma^2 is formed on the fly from rho; the closed-form KL sums
// (sum log sigma, sum sigma^2, sum mu^2; networks.py:113) are taken from the same registers
// by the blocks of sample 0, so (M, rho) are read once for GEMMs and KL together.
// eps_act is generated in the epilogue, in the D-fragment layout (4 consecutive features
// of one batch row = one Philox call).
#include "bnn_device.h"
#include "../../include/bnn_hip.h"

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
  float* partial;   // [T][4]
  int S, B, K, N;
  int eps_mode, want_kl, relu, y_bf16;
  uint32_t k0, k1, layer_id, sample_offset;
  const uint32_t* sample_counter;
};

template <int MATH, int XDT>
__global__ __launch_bounds__(256) void lr_linear_fwd_kernel(const LrK p) {
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
  const bool do_kl = p.want_kl && mb == 0 && s == 0;
  const bool do_dump = nt >= 0;   // every (s, mb, nt) block owns distinct eps_act elements
  const int ksteps = (K + 31) >> 5;
  const char* xs = reinterpret_cast<const char*>(p.x) +
                   (size_t)s * (size_t)p.x_sstride * (XDT == BNN_F32 ? 4 : 2);

  f32x4 am[8], av[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    am[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    av[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float s_ls = 0.f, s_s2 = 0.f, s_m2 = 0.f;

  for (int t = wave; t < ksteps; t += nw) {
    const int k = t * 32 + q * 8;
    float mu[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = n_ok && (k + j) < K;
      const size_t off = (size_t)(k + j) * N + n;
      const float m_ = ok ? p.w_mu[off] : 0.f;
      const float rho = ok ? p.w_rho[off] : 0.f;
      const float sig = softplus(rho);
      mu[j] = m_;
      s2[j] = ok ? sig * sig : 0.f;
      if (do_kl) {
        s_ls += ok ? fast_log(sig) : 0.f;
        s_s2 += s2[j];
        s_m2 = __builtin_fmaf(m_, m_, s_m2);
      }
    }
    bf16x8 ma, sa;
    if (MATH == BNN_MATH_BF16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ma[j] = (__bf16)mu[j];
        sa[j] = (__bf16)s2[j];
      }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < mtiles) {
        const int row = m0 + m * 16 + r;
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = row < B && (k + j) < K;
          const long off = (long)row * K + k + j;
          if (XDT == BNN_F32)
            xv[j] = ok ? reinterpret_cast<const float*>(xs)[off] : 0.f;
          else
            xv[j] = ok ? (float)reinterpret_cast<const __bf16*>(xs)[off] : 0.f;
        }
        if (MATH == BNN_MATH_BF16) {
          bf16x8 xb, x2b;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            xb[j] = (__bf16)xv[j];
            x2b[j] = (__bf16)(xv[j] * xv[j]);
          }
          am[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, xb, am[m], 0, 0, 0);
          av[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, x2b, av[m], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            am[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(mu[j], xv[j], am[m], 0, 0, 0);
            av[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s2[j], xv[j] * xv[j], av[m], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- bias (wave 0, lanes 0..15)
  const size_t slab_floats = (size_t)nw * 8 * 64 * 4;
  float* lds_bias = lds + 2 * slab_floats;
  float* lds_red = lds_bias + 16;
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
      if (p.eps_b_dump && mb == 0) p.eps_b_dump[(size_t)s * N + n] = e;
      b = __builtin_fmaf(sig, e, bmu);
      if (do_kl) {
        s_ls += fast_log(sig);
        s_s2 = __builtin_fmaf(sig, sig, s_s2);
        s_m2 = __builtin_fmaf(bmu, bmu, s_m2);
      }
    }
    lds_bias[r] = b;
  }

  f32x4* slab_m = reinterpret_cast<f32x4*>(lds);
  f32x4* slab_v = reinterpret_cast<f32x4*>(lds + slab_floats);
#pragma unroll
  for (int m = 0; m < 8; ++m)
    if (m < mtiles) {
      slab_m[(wave * 8 + m) * 64 + lane] = am[m];
      slab_v[(wave * 8 + m) * 64 + lane] = av[m];
    }
  if (do_kl) {
    const float a = wave_sum(s_ls), b = wave_sum(s_s2), c = wave_sum(s_m2);
    if (lane == 0) {
      lds_red[wave * 3 + 0] = a;
      lds_red[wave * 3 + 1] = b;
      lds_red[wave * 3 + 2] = c;
    }
  }
  __syncthreads();

  if (do_kl && threadIdx.x == 0) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int wv = 0; wv < nw; ++wv) {
      a += lds_red[wv * 3 + 0];
      b += lds_red[wv * 3 + 1];
      c += lds_red[wv * 3 + 2];
    }
    reinterpret_cast<float4*>(p.partial)[1 + nt] = make_float4(a, b, c, 0.f);
    if (nt == 0) reinterpret_cast<float4*>(p.partial)[0] = make_float4(__int_as_float((int)gridDim.x), 0.f, 0.f, 0.f);
  }

  const bool vec_ok = (N & 3) == 0;
  const int gprN = (N + 3) >> 2;
  for (int item = threadIdx.x; item < mtiles * 64; item += blockDim.x) {
    const int m = item >> 6, l = item & 63;
    f32x4 vm = slab_m[m * 64 + l], vv = slab_v[m * 64 + l];
    for (int wv = 1; wv < nw; ++wv) {
      vm += slab_m[(wv * 8 + m) * 64 + l];
      vv += slab_v[(wv * 8 + m) * 64 + l];
    }
    const int brow = m0 + m * 16 + (l & 15);
    const int f0 = (l >> 4) * 4;
    const int nb = nt * 16 + f0;
    if (brow >= B || nb >= N) continue;
    const size_t yoff = ((size_t)s * B + brow) * N + nb;
    float e4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.eps_mode == BNN_EPS_PHILOX) {
      philox_normal4((uint32_t)brow * (uint32_t)gprN + (uint32_t)(nb >> 2), gs, p.layer_id * 4u + 2u, p.k0, p.k1, e4);
    } else if (p.eps_mode == BNN_EPS_MEMORY) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nb + i < N) e4[i] = p.eps_act[yoff + i];
    }
    if (p.eps_act_dump && do_dump) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nb + i < N) p.eps_act_dump[yoff + i] = e4[i];
    }
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o = __builtin_fmaf(__builtin_amdgcn_sqrtf(vv[i]), e4[i], vm[i]) + lds_bias[f0 + i];
      if (p.relu) o = fmaxf(o, 0.f);
      v[i] = o;
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
}

// KL of one LR layer from its partials (networks.py:113, :134-136).  The partials fold
// weights and biases together (the network only needs the total); the bias term is small
// (N elements) and is recomputed here so that weight_kl_cost / bias_kl_cost can be
// reported separately.  out3 = {kl, weight_kl, bias_kl}.
__global__ void lr_layer_kl_kernel(const float* __restrict__ partial, int K, int N, float sigma_p,
                                   const float* __restrict__ b_mu, const float* __restrict__ b_rho,
                                   float* __restrict__ out3) {
  __shared__ double scratch[16];
  double ls = 0, s2 = 0, m2 = 0, bls = 0, bs2 = 0, bm2 = 0;
  const int T = __float_as_int(reinterpret_cast<const float4*>(partial)[0].x);
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(partial)[1 + t];
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

extern "C" size_t bnn_lr_linear_fwd_workspace_bytes(int32_t out_features) {
  if (out_features <= 0) return 0;
  return (1 + (size_t)((out_features + 3) / 4)) * 4 * sizeof(float);
}

extern "C" int bnn_lr_linear_fwd(const bnn_lr_fwd_args* a, void* stream_) {
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_lr_fwd_args)) return BNN_ERR_ABI;
  if (a->n_samples <= 0 || a->batch <= 0 || a->in_features <= 0 || a->out_features <= 0) return BNN_ERR_SHAPE;
  if (a->n_samples > 65535 || (a->batch + 127) / 128 > 65535) return BNN_ERR_SHAPE;
  if (!a->x || !a->w_mu || !a->w_rho || !a->b_mu || !a->b_rho || !a->y) return BNN_ERR_NULL;
  if ((unsigned)a->x_dtype > 1u || (unsigned)a->y_dtype > 1u || (unsigned)a->math > 1u || (unsigned)a->eps_mode > 2u)
    return BNN_ERR_ENUM;
  if (a->eps_mode == BNN_EPS_MEMORY && (!a->eps_act || !a->eps_b)) return BNN_ERR_NULL;
  const int T = (a->out_features + 15) / 16;
  if (a->want_kl) {
    if (!a->workspace || a->workspace_bytes < bnn_lr_linear_fwd_workspace_bytes(a->out_features))
      return BNN_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(a->workspace) & 15) return BNN_ERR_ALIGN;
    if (!(a->sigma_p > 0.f)) return BNN_ERR_SHAPE;
  }
  if (a->kl_out && !a->want_kl) return BNN_ERR_WORKSPACE;
  const bool ybf = a->y_dtype == BNN_BF16;
  if ((a->out_features % 4 == 0) && (reinterpret_cast<uintptr_t>(a->y) & (ybf ? 7 : 15))) return BNN_ERR_ALIGN;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);

  LrK k;
  k.x = a->x;
  k.x_sstride = a->x_per_sample ? (long)a->batch * a->in_features : 0;
  k.w_mu = a->w_mu; k.w_rho = a->w_rho; k.b_mu = a->b_mu; k.b_rho = a->b_rho;
  k.eps_act = a->eps_act; k.eps_b = a->eps_b; k.eps_act_dump = a->eps_act_dump; k.eps_b_dump = a->eps_b_dump;
  k.y = a->y;
  k.partial = reinterpret_cast<float*>(a->workspace);
  k.S = a->n_samples; k.B = a->batch; k.K = a->in_features; k.N = a->out_features;
  k.eps_mode = a->eps_mode; k.want_kl = a->want_kl ? 1 : 0; k.relu = a->relu ? 1 : 0; k.y_bf16 = ybf;
  k.k0 = (uint32_t)a->seed; k.k1 = (uint32_t)(a->seed >> 32);
  k.layer_id = a->layer_id; k.sample_offset = a->sample_offset; k.sample_counter = a->sample_counter;

  const int ksteps = (a->in_features + 31) / 32;
  int nw = ksteps / 3;
  nw = nw < 1 ? 1 : (nw > 4 ? 4 : nw);
  const dim3 grid(T, a->n_samples, (a->batch + 127) / 128), block(nw * 64);
  const size_t lds = (2 * (size_t)nw * 8 * 64 * 4 + 16 + 3 * nw) * sizeof(float);
#define BNN_LAUNCH(MATH, XDT) hipLaunchKernelGGL((lr_linear_fwd_kernel<MATH, XDT>), grid, block, lds, stream, k)
  if (a->math == BNN_MATH_BF16) {
    if (a->x_dtype == BNN_F32) BNN_LAUNCH(BNN_MATH_BF16, BNN_F32); else BNN_LAUNCH(BNN_MATH_BF16, BNN_BF16);
  } else {
    if (a->x_dtype == BNN_F32) BNN_LAUNCH(BNN_MATH_F32, BNN_F32); else BNN_LAUNCH(BNN_MATH_F32, BNN_BF16);
  }
#undef BNN_LAUNCH
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  if (a->kl_out) {
    hipLaunchKernelGGL(lr_layer_kl_kernel, dim3(1), dim3(256), 0, stream, k.partial, a->in_features,
                       a->out_features, a->sigma_p, a->b_mu, a->b_rho, a->kl_out);
    err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  return BNN_OK;
}
