// F1  backward of BayesianLinear (what autograd does for reference networks.py:73-88 when
// classification/class_task.py:78 calls loss.backward()), SURVEY Appendix A.5.
//
// For MC samples s = 0..S-1 with upstream gradients gy_s (w.r.t. the layer output), glp_s
// (w.r.t. the layer's log p) and glq_s (w.r.t. its log q):
//     gz_s   = gy_s * (y_s > 0)                       (the ReLU fused into the forward)
//     gW_s   = gz_s^T . x_s                           [out, in]
//     t_s    = gW_s + glp_s * dlogp/dw (w_s),   w_s = mu + sigma * eps_s  (eps REGENERATED)
//     g_mu   = sum_s t_s
//     g_rho  = ( sum_s t_s * eps_s  -  (sum_s glq_s) / sigma ) * sigmoid(rho)
// and likewise for the bias with gb_s = column sums of gz_s.  The input gradient
// gx_s = gz_s . w_s is produced by the forward K-split kernel with a transposed
// weight-fragment generator (bbb_linear.hip, TRANS).
//
// `bbb_bwd_weights_kernel`: grid (ceil(K/64), ceil(N/64)), 4 waves.  Wave j owns feature tile
// n0 = 64*by + 16*j and the 4 k-tiles of the block's 64-wide k strip and walks the samples.
// gW^T tiles come from the exact-fp32 matrix core (v_mfma_f32_16x16x4_f32): with the reduction
// over the batch, A[i = k][kk = b] = x[b][k0 + i] and B[kk = b][j = n] = gz[b][n0 + j] are both
// read along their contiguous dimension.  The four k-tiles of a strip interleave their k's (tile i
// holds k = i mod 4) so one 16-byte x load feeds all four and each D register position, taken
// across the tiles, is 4 consecutive k of one feature — exactly one Philox group — so eps is
// regenerated in the epilogue and neither eps nor w ever exists in memory.  (G, H) = (sum t, sum t*eps) stay in registers across the
// sample loop; one pass writes g_mu and g_rho.  No atomics.
#include "bnn_device.h"
#include "../../include/bnn_hip.h"

namespace bnn {

struct BwdK {
  const float* x;       // [Sx, B, K] fp32
  long x_sstride;
  const float* gz;      // [S, B, N] fp32, ReLU mask applied
  const float* w_mu;
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  const float* eps_w;   // BNN_EPS_MEMORY
  const float* eps_b;
  const float* glp;     // [S] or nullptr (zeros)
  const float* glq;     // [S] or nullptr
  float* g_wmu;
  float* g_wrho;
  float* g_bmu;
  float* g_brho;
  int S, B, K, N;
  int eps_mode, prior_kind;
  uint32_t k0, k1, layer_id, sample_offset;
  const uint32_t* sample_counter;
  float inv_var_p;                                   // Gaussian prior: 1 / sigma_p^2
  float a1, a2, inv2var1, inv2var2, invvar1, invvar2; // mixture: a_i = pi_i / sigma_i
#ifdef BNN_STAMPS
  unsigned long long* dbg;   // diagnostic build only: [block][16] shader-clock stamps of wave 0
#endif
};

#ifdef BNN_STAMPS
#define BWD_STAMP(i)                                                             \
  do {                                                                           \
    if (p.dbg && threadIdx.x == 0) p.dbg[((size_t)blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define BWD_STAMP_RT(i)                                                          \
  do {                                                                           \
    if (p.dbg && threadIdx.x == 0) p.dbg[((size_t)blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define BWD_STAMP(i)
#define BWD_STAMP_RT(i)
#endif

// d log p(w) / dw
__device__ __forceinline__ float dlogp(const BwdK& p, float w) {
  if (p.prior_kind == BNN_PRIOR_GAUSS) return -w * p.inv_var_p;
  const float w2 = w * w;
  const float n1 = p.a1 * fast_exp(-w2 * p.inv2var1);
  const float n2 = p.a2 * fast_exp(-w2 * p.inv2var2);
  return -w * (n1 * p.invvar1 + n2 * p.invvar2) * __builtin_amdgcn_rcpf(n1 + n2);
}

__device__ __forceinline__ float sigmoidf(float r) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-r)); }

// VEC: K % 4 == 0 and 16-byte aligned rows.  No load sits under per-lane control flow (a load in a
// divergent branch gets its own basic block and the waits between blocks serialise the batch):
// indices are clamped and the result selected afterwards.
// MEM: eps comes from memory (BNN_EPS_MEMORY).  The Philox instantiation has no load in its per-sample epilogue,
// so no wait there holds it back while the next sample's first group is in flight.
//
// Block = 32 features x 64 k, four waves: wave = (sub, half).  `sub` picks 16 of the features, `half` the parity of
// the 16-row batch groups the wave reduces over; per sample the two halves swap partial accumulators through LDS and
// each finishes HALF of the tile's weights (D registers 2 half, 2 half + 1: two Philox groups per lane).  The kernel
// is bound by the fp32 matrix core and the generator, so what matters is how evenly the work lands on the SIMDs:
// 64 x 64 blocks of four full waves were 1444 equal units on 1024 SIMDs (makespan 2 units); these are 2888 half
// units (makespan 3 halves) at the 1200 x 1200 layer.
template <bool VEC, bool MEM>
__global__ __launch_bounds__(256, 3) void bbb_bwd_weights_kernel(const BwdK p) {
  __shared__ float xch[2][2][2][8][64];                   // [sample parity][sub][destination half][jr * 4 + tile][lane]
  __shared__ float xcs[2][2][64];                         // [sample parity][sub][lane]: half 1's column sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = wave & 1, half = wave >> 1;
  const int r = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N, B = p.B;
  BWD_STAMP(0);
  BWD_STAMP_RT(8);
  // XCD-aware order: an XCD walks a few feature blocks (its slice of gz) across all k strips, so
  // x and that slice stay in its own L2 instead of every L2 holding everything
  const int nkb = (K + 63) >> 6;
  int item;
  if (!xcd_work_item(nkb * ((N + 31) >> 5), item)) return;   // block-uniform: before any barrier
  const int kblk = item % nkb, nblk = item / nkb;
  const int k0 = kblk * 64;
  const int n = nblk * 32 + sub * 16 + r;                  // this lane's feature (D column / B-operand column)
  const bool n_ok = n < N;                                 // a sub-tile past N runs on clamped data and stores nothing
  const int gpr = (K + 3) >> 2;

  // Tile i (i = 0..3) of the 64-wide k strip holds the k's congruent to i mod 4: A row r of tile i is
  // k = k0 + 4 r + i, so ONE 16-byte load x[b][k0 + 4r .. +3] feeds all four tiles, and D register
  // `reg` of lane quad q, taken across the four tiles, is the 4 consecutive k's
  // k0 + 16 q + 4 reg + {0,1,2,3} of feature n: one Philox group.  This wave finishes reg = 2 half + jr.
  float mu[2][4], sg[2][4], rh[2][4];                      // [jr][i]
  float G[4][2], H[4][2];                                  // [i][jr]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jr = 0; jr < 2; ++jr) G[i][jr] = H[i][jr] = 0.f;
  // raw parameter loads now; they are unpacked (softplus) after the first x / gz group has been requested too,
  // so the launch starts with ONE memory round trip instead of two
  float4 m4r[2], r4r[2];
#pragma unroll
  for (int jr = 0; jr < 2; ++jr) {
    const int kq = k0 + q * 16 + (2 * half + jr) * 4;      // 4 consecutive k of feature n: one 16-byte access
    const size_t rowoff = (size_t)min(n, N - 1) * K;
    if (VEC) {
      m4r[jr] = *reinterpret_cast<const float4*>(p.w_mu + rowoff + min(kq, K - 4));
      r4r[jr] = *reinterpret_cast<const float4*>(p.w_rho + rowoff + min(kq, K - 4));
    } else {
      m4r[jr] = make_float4(p.w_mu[rowoff + min(kq + 0, K - 1)], p.w_mu[rowoff + min(kq + 1, K - 1)],
                            p.w_mu[rowoff + min(kq + 2, K - 1)], p.w_mu[rowoff + min(kq + 3, K - 1)]);
      r4r[jr] = make_float4(p.w_rho[rowoff + min(kq + 0, K - 1)], p.w_rho[rowoff + min(kq + 1, K - 1)],
                            p.w_rho[rowoff + min(kq + 2, K - 1)], p.w_rho[rowoff + min(kq + 3, K - 1)]);
    }
  }
  const bool do_bias = kblk == 0 && half == 0 && q == 0 && n_ok;   // one lane per feature
  const float bmu_raw = p.b_mu[min(n, N - 1)], brh_raw = p.b_rho[min(n, N - 1)];
  float Gb = 0.f, Hb = 0.f, bmu = 0.f, brh = 0.f, bsg = 1.f;   // bias parameters: unpacked with the weights' below
  float cq = 0.f;
  const uint32_t sample_base = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u);
  const int ka = k0 + 4 * r;                               // first k of this lane's A-operand quad

  // The wave's (sample, 16-row group) pairs form one pipeline over two register buffers: the 8 loads of the next
  // group -- of the next SAMPLE at a sample's end -- are in flight while this group's 16 MFMAs issue and while the
  // sample's epilogue regenerates eps.
  constexpr int U = 4;                                     // batch-row quads per group (16 rows): two buffers of 8 loads fit 3 blocks per CU
  const int Gh = (((B + 4 * U - 1) / (4 * U)) + 1) >> 1;   // groups per sample and half (a group past B is all masked)
  const int total_groups = p.S * Gh;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float colsum = 0.f;
  float4 avA[U], avB[U];
  float bvA[U], bvB[U];
  // the sample's upstream scalars ride with its groups (a load inside the epilogue would make it wait for
  // everything in flight): always loaded from a valid address, selected afterwards
  const float* glp_src = p.glp ? p.glp : p.b_mu;
  const float* glq_src = p.glq ? p.glq : p.b_mu;
  float gsA[2], gsB[2];
  auto load_group = [&](int gi, float4 (&av)[U], float (&bv)[U], float (&gsc)[2]) {
    const int s = gi / Gh, b0 = (2 * (gi - s * Gh) + half) * (4 * U);
    gsc[0] = glp_src[p.glp ? s : 0];
    gsc[1] = glq_src[p.glq ? s : 0];
    const float* xs = p.x + (size_t)s * (size_t)p.x_sstride;
    const float* gzs = p.gz + (size_t)s * B * N;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int brow = b0 + 4 * u + q;
      const float* xr = xs + (size_t)min(brow, B - 1) * K;
      if (VEC) {
        av[u] = *reinterpret_cast<const float4*>(xr + min(ka, K - 4));   // k >= K: rows of D that are never stored
      } else {
        av[u].x = xr[min(ka + 0, K - 1)];
        av[u].y = xr[min(ka + 1, K - 1)];
        av[u].z = xr[min(ka + 2, K - 1)];
        av[u].w = xr[min(ka + 3, K - 1)];
      }
      const float gzv = gzs[(size_t)min(brow, B - 1) * N + min(n, N - 1)];
      bv[u] = brow < B ? gzv : 0.f;                                       // rows >= B contribute nothing
    }
  };
  auto mfma_group = [&](const float4 (&av)[U], const float (&bv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      colsum += bv[u];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u], acc[3], 0, 0, 0);
    }
  };
  // end of sample s for this wave: swap partials with the other half, then regenerate eps of the lane's two weight
  // groups and fold the sample's gW into (G, H)
  auto finish_sample = [&](int s, const float (&gsc)[2]) {
    if (s == 0) { asm volatile("" ::"v"(acc[3][3])); BWD_STAMP(3); }
    const int par = s & 1, other = half ^ 1;
    // a wave is never more than one sample (one barrier) ahead of its block, so two parities of the buffer suffice
#pragma unroll
    for (int jr = 0; jr < 2; ++jr)
#pragma unroll
      for (int i = 0; i < 4; ++i) xch[par][sub][other][jr * 4 + i][lane] = half ? acc[i][jr] : acc[i][2 + jr];   // the OTHER half's registers
    // bias: gb_s[n] = sum_b gz_s[b][n]; the 4 lane quads hold b = q (mod 4)
    colsum += __shfl_xor(colsum, 16, kWave);
    colsum += __shfl_xor(colsum, 32, kWave);
    if (half == 1) xcs[par][sub][lane] = colsum;
    __syncthreads();
    const float glp = p.glp ? gsc[0] : 0.f;
    cq += p.glq ? gsc[1] : 0.f;
    const uint32_t gs = sample_base + (uint32_t)s;
#pragma unroll
    for (int jr = 0; jr < 2; ++jr) {
      const int reg = 2 * half + jr;
      const int kb = k0 + q * 16 + reg * 4;
      float tot[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) tot[i] = (half ? acc[i][2 + jr] : acc[i][jr]) + xch[par][sub][half][jr * 4 + i][lane];
      if (n_ok && kb < K) {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.eps_mode == BNN_EPS_PHILOX) {
          philox_normal4((uint32_t)n * (uint32_t)gpr + (uint32_t)(kb >> 2), gs, p.layer_id * 4u, p.k0, p.k1, e);
        } else if (MEM && p.eps_mode == BNN_EPS_MEMORY) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (kb + i < K) e[i] = p.eps_w[((size_t)s * N + n) * K + kb + i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float w = __builtin_fmaf(sg[jr][i], e[i], mu[jr][i]);
          const float t = tot[i] + glp * dlogp(p, w);
          G[i][jr] += t;
          H[i][jr] = __builtin_fmaf(t, e[i], H[i][jr]);
        }
      }
    }
    if (do_bias) {
      float e = 0.f;
      if (p.eps_mode == BNN_EPS_PHILOX) {
        float e4[4];
        philox_normal4((uint32_t)(n >> 2), gs, p.layer_id * 4u + 1u, p.k0, p.k1, e4);
        e = (n & 3) == 0 ? e4[0] : (n & 3) == 1 ? e4[1] : (n & 3) == 2 ? e4[2] : e4[3];
      } else if (MEM && p.eps_mode == BNN_EPS_MEMORY) {
        e = p.eps_b[(size_t)s * N + n];
      }
      const float bw = __builtin_fmaf(bsg, e, bmu);
      const float t = colsum + xcs[par][sub][lane] + glp * dlogp(p, bw);
      Gb += t;
      Hb = __builtin_fmaf(t, e, Hb);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    colsum = 0.f;
    if (s == 0) { asm volatile("" ::"v"(G[0][0]), "v"(H[3][1])); BWD_STAMP(4); }
  };
  // Loads are issued unconditionally (the last ones re-read the final group): a load under a branch would make the
  // join point wait for vmcnt(0), i.e. for the group just issued, which is the round trip this pipeline hides.
  const int last_group = total_groups - 1;
  load_group(0, avA, bvA, gsA);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int jr = 0; jr < 2; ++jr) {
    mu[jr][0] = m4r[jr].x; mu[jr][1] = m4r[jr].y; mu[jr][2] = m4r[jr].z; mu[jr][3] = m4r[jr].w;
    rh[jr][0] = r4r[jr].x; rh[jr][1] = r4r[jr].y; rh[jr][2] = r4r[jr].z; rh[jr][3] = r4r[jr].w;
#pragma unroll
    for (int i = 0; i < 4; ++i) sg[jr][i] = softplus(rh[jr][i]);
  }
  if (do_bias) {
    bmu = bmu_raw;
    brh = brh_raw;
    bsg = softplus(brh);
  }
  for (int gi = 0; gi < total_groups; gi += 2) {
    load_group(min(gi + 1, last_group), avB, bvB, gsB);
    __builtin_amdgcn_sched_barrier(0);                     // the loads stay a batch, issued ahead of the MFMAs
    if (gi == 0) { asm volatile("" ::"v"(sg[0][0]), "v"(avA[0].x)); BWD_STAMP(1); }
    mfma_group(avA, bvA);
    if (gi == 0) { asm volatile("" ::"v"(acc[0][0])); BWD_STAMP(2); }
    if ((gi + 1) % Gh == 0) finish_sample(gi / Gh, gsA);
    load_group(min(gi + 2, last_group), avA, bvA, gsA);
    __builtin_amdgcn_sched_barrier(0);
    if (gi + 1 < total_groups) {
      mfma_group(avB, bvB);
      if ((gi + 2) % Gh == 0) finish_sample((gi + 1) / Gh, gsB);
    }
  }

  // ---- g_mu = G;  g_rho = (H - cq / sigma) * sigmoid(rho)
#pragma unroll
  for (int jr = 0; jr < 2; ++jr) {
    const int kq = k0 + q * 16 + (2 * half + jr) * 4;
    if (!n_ok || kq >= K) continue;
    float gr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gr[i] = (H[i][jr] - cq * __builtin_amdgcn_rcpf(sg[jr][i])) * sigmoidf(rh[jr][i]);
    const size_t off = (size_t)n * K + kq;
    if (VEC) {
      *reinterpret_cast<float4*>(p.g_wmu + off) = make_float4(G[0][jr], G[1][jr], G[2][jr], G[3][jr]);
      *reinterpret_cast<float4*>(p.g_wrho + off) = make_float4(gr[0], gr[1], gr[2], gr[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (kq + i < K) {
          p.g_wmu[off + i] = G[i][jr];
          p.g_wrho[off + i] = gr[i];
        }
    }
  }
  if (do_bias) {
    p.g_bmu[n] = Gb;
    p.g_brho[n] = (Hb - cq * __builtin_amdgcn_rcpf(bsg)) * sigmoidf(brh);
  }
  BWD_STAMP(7);
  BWD_STAMP_RT(9);
}

// weight (and bias) gradients of 16 k columns: the weight role of bbb_out_layer_bwd_kernel
template <bool PAIR>
__device__ __forceinline__ void out_bwd_weight_role(const BwdK& p, float* gzs, float* red) {
  const int K = p.K, N = p.N, B = p.B, S = p.S;
  const int tid = threadIdx.x;
  const int k0 = (int)blockIdx.x * 16;
  const int kc = tid & 15, bg = tid >> 4;
  const int kx = min(k0 + kc, K - 1);
  const bool k_ok = k0 + kc < K;
  // epilogue threads: (feature, group of 4 k)
  const int en = tid >> 2, eg = tid & 3;
  const int ekb = k0 + eg * 4;
  const bool e_ok = tid < 64 && en < N && ekb < K;          // K % 4 == 0: a group is whole or absent
  const bool b_ok = blockIdx.x == 0 && tid >= 64 && tid < 64 + N;   // bias: thread 64 + n
  const int bn = min(max(tid - 64, 0), N - 1);
  const int gpr = (K + 3) >> 2;
  float mu[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {1.f, 1.f, 1.f, 1.f}, rh[4] = {0.f, 0.f, 0.f, 0.f};
  if (e_ok) {
    const float4 m4 = *reinterpret_cast<const float4*>(p.w_mu + (size_t)en * K + ekb);
    const float4 r4 = *reinterpret_cast<const float4*>(p.w_rho + (size_t)en * K + ekb);
    mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
    rh[0] = r4.x; rh[1] = r4.y; rh[2] = r4.z; rh[3] = r4.w;
#pragma unroll
    for (int i = 0; i < 4; ++i) sg[i] = softplus(rh[i]);
  }
  const float bmu = p.b_mu[bn], brh = p.b_rho[bn];
  const float bsg = softplus(brh);
  float G[4] = {0.f, 0.f, 0.f, 0.f}, H[4] = {0.f, 0.f, 0.f, 0.f}, Gb = 0.f, Hb = 0.f, cq = 0.f;
  const uint32_t sample_base = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u);
  // PAIR (112 < batch <= 128, the MNIST shape: J = 8 rows per thread and sample, known at compile time): samples go
  // through in ROUNDS of two -- the x loads and the gz staging of both are one memory round trip, the per-sample
  // epilogues that follow touch LDS only.
  const int J = PAIR ? 8 : (B + 15) >> 4;                   // rows per thread and sample
  constexpr int SP = PAIR ? 2 : 1;
  for (int s0 = 0; s0 < S; s0 += SP) {
    float xv[16], glp2[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int sp = PAIR ? (j >> 3) : 0;
      const int b = bg + 16 * (j - sp * J);
      const int sm = min(s0 + sp, S - 1);
      const float v = p.x[(size_t)sm * (size_t)p.x_sstride + (size_t)min(b, B - 1) * K + kx];
      xv[j] = (j < J * SP && s0 + sp < S && b < B && k_ok) ? v : 0.f;
    }
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const int sm = min(s0 + sp, S - 1);
      glp2[sp] = p.glp ? p.glp[sm] : 0.f;
      if (sp < SP && s0 + sp < S) cq += p.glq ? p.glq[sm] : 0.f;
    }
    __syncthreads();                                        // the previous round's readers of gzs / red are done
    for (int sp = 0; sp < SP; ++sp) {
      const float* gzg = p.gz + (size_t)min(s0 + sp, S - 1) * B * N;
      for (int i = tid; i < B * 16; i += 256) {
        const int b = i >> 4, n = i & 15;
        gzs[sp * 128 * 16 + i] = n < N ? gzg[(size_t)b * N + n] : 0.f;
      }
    }
    __syncthreads();
    float acc[2][16];
#pragma unroll
    for (int n = 0; n < 16; ++n) acc[0][n] = acc[1][n] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < J * SP) {
        const int sp = PAIR ? (j >> 3) : 0;
        const int b = min(bg + 16 * (j - sp * J), B - 1);
        const float4* g4 = reinterpret_cast<const float4*>(gzs + (sp * 128 + b) * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float4 g = g4[v];
          if (sp == 0) {
            acc[0][4 * v + 0] = __builtin_fmaf(xv[j], g.x, acc[0][4 * v + 0]);
            acc[0][4 * v + 1] = __builtin_fmaf(xv[j], g.y, acc[0][4 * v + 1]);
            acc[0][4 * v + 2] = __builtin_fmaf(xv[j], g.z, acc[0][4 * v + 2]);
            acc[0][4 * v + 3] = __builtin_fmaf(xv[j], g.w, acc[0][4 * v + 3]);
          } else {
            acc[1][4 * v + 0] = __builtin_fmaf(xv[j], g.x, acc[1][4 * v + 0]);
            acc[1][4 * v + 1] = __builtin_fmaf(xv[j], g.y, acc[1][4 * v + 1]);
            acc[1][4 * v + 2] = __builtin_fmaf(xv[j], g.z, acc[1][4 * v + 2]);
            acc[1][4 * v + 3] = __builtin_fmaf(xv[j], g.w, acc[1][4 * v + 3]);
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      red[(bg * 16 + n) * 16 + kc] = acc[0][n];
      red[4096 + (bg * 16 + n) * 16 + kc] = acc[1][n];
    }
    __syncthreads();
    for (int sp = 0; sp < SP && s0 + sp < S; ++sp) {
      const float glp = glp2[sp];
      const uint32_t gs = sample_base + (uint32_t)(s0 + sp);
      const float* rd = red + sp * 4096;
      if (e_ok) {
        float tot[4] = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < 16; ++g) {
#pragma unroll
          for (int i = 0; i < 4; ++i) tot[i] += rd[(g * 16 + en) * 16 + eg * 4 + i];
        }
        float e[4];
        philox_normal4((uint32_t)en * (uint32_t)gpr + (uint32_t)(ekb >> 2), gs, p.layer_id * 4u, p.k0, p.k1, e);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float w = __builtin_fmaf(sg[i], e[i], mu[i]);
          const float t = tot[i] + glp * dlogp(p, w);
          G[i] += t;
          H[i] = __builtin_fmaf(t, e[i], H[i]);
        }
      }
      if (b_ok) {
        float colsum = 0.f;
        for (int b = 0; b < B; ++b) colsum += gzs[(sp * 128 + b) * 16 + bn];
        float e4[4];
        philox_normal4((uint32_t)(bn >> 2), gs, p.layer_id * 4u + 1u, p.k0, p.k1, e4);
        const float e = (bn & 3) == 0 ? e4[0] : (bn & 3) == 1 ? e4[1] : (bn & 3) == 2 ? e4[2] : e4[3];
        const float bw = __builtin_fmaf(bsg, e, bmu);
        const float t = colsum + glp * dlogp(p, bw);
        Gb += t;
        Hb = __builtin_fmaf(t, e, Hb);
      }
    }
  }
  if (e_ok) {
    float gr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gr[i] = (H[i] - cq * __builtin_amdgcn_rcpf(sg[i])) * sigmoidf(rh[i]);
    const size_t off = (size_t)en * K + ekb;
    *reinterpret_cast<float4*>(p.g_wmu + off) = make_float4(G[0], G[1], G[2], G[3]);
    *reinterpret_cast<float4*>(p.g_wrho + off) = make_float4(gr[0], gr[1], gr[2], gr[3]);
  }
  if (b_ok) {
    p.g_bmu[bn] = Gb;
    p.g_brho[bn] = (Hb - cq * __builtin_amdgcn_rcpf(bsg)) * sigmoidf(brh);
  }
}

// Backward of a NARROW output layer (N <= 16: the 10 classes / the 1 regression output) over the step's pre-sampled
// weights, weight gradients and input gradient in ONE launch.  The general kernels above and the K-split input
// gradient are built for wide layers: at 1200 -> 10 they are 19 and 152 one-or-four-wave blocks walking the batch
// serially (9.7 + 14 us of a 0.18 ms training step, two launches).  Here the work is a few MFLOP, so plain fp32 FMAs
// spread over many short blocks do it:
//   weight role (blocks < wblocks): 16 k columns x all features.  Thread (kc, bg) accumulates the rows b = bg (mod 16)
//     of gW_s[n][k0 + kc] = sum_b gz_s[b][n] x_s[b][k] against the sample's gz staged in LDS, one LDS round adds the 16
//     row classes up, 64 threads (feature, Philox group of 4 k) regenerate eps and fold the sample into (G, H) exactly
//     as bbb_bwd_weights_kernel's epilogue does; block 0 also takes the bias.
//   input role: a thread owns 8 consecutive k of one (sample, row): gx = (x > 0) * sum_n bf16(gz[n]) * w_s[n][k]
//     (gz rounded to bf16 as the matrix-core form of the same product does; the products are then exact in fp32).
struct OutBwd {
  const __bf16* w;      // [S, N, K] the forward's sampled weights
  float* gx;            // [S, B, K]
  __bf16* gx16;         // optional bf16 copy of gx (the layer below's input-gradient launch reads it)
  int relu_mask;        // gx *= (x > 0)
  int wblocks;
};

template <bool PAIR>
__global__ __launch_bounds__(256) void bbb_out_layer_bwd_kernel(const BwdK p, const OutBwd o) {
  __shared__ __attribute__((aligned(16))) float gzs[256 * 16];     // [b][n] of the current sample, features padded to 16
  __shared__ float red[2 * 16 * 16 * 16];                          // [sample of the round][bg][n][kc]
  const int K = p.K, N = p.N, B = p.B, S = p.S;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= o.wblocks) {
    // ---- input gradient
    const int K8 = K >> 3;
    const long idx = (long)((int)blockIdx.x - o.wblocks) * 256 + tid;
    if (idx >= (long)S * B * K8) return;
    const int kch = (int)(idx % K8);
    const long row = idx / K8;                             // s * B + b
    const int s = (int)(row / B), b = (int)(row - (long)s * B);
    const float* xr = p.x + (size_t)s * (size_t)p.x_sstride + (size_t)b * K + kch * 8;
    const float4 x0 = *reinterpret_cast<const float4*>(xr), x1 = *reinterpret_cast<const float4*>(xr + 4);
    const float* gr = p.gz + (size_t)row * N;
    const __bf16* wr = o.w + (size_t)s * N * K + kch * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gv[16];
    bf16x8 wv[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      const int nn = min(n, N - 1);
      gv[n] = gr[nn];
      wv[n] = *reinterpret_cast<const bf16x8*>(wr + (size_t)nn * K);
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      const float g = n < N ? (float)(__bf16)gv[n] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(g, (float)wv[n][j], acc[j]);
    }
    if (o.relu_mask) {
      const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = xv[j] > 0.f ? acc[j] : 0.f;
    }
    float* out = o.gx + (size_t)row * K + kch * 8;
    *reinterpret_cast<float4*>(out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(out + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    if (o.gx16) {
      bf16x8 ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) ob[j] = (__bf16)acc[j];
      *reinterpret_cast<bf16x8*>(o.gx16 + (size_t)row * K + kch * 8) = ob;
    }
    return;
  }
  // ---- weight (and bias) gradients of 16 k columns
  out_bwd_weight_role<PAIR>(p, gzs, red);
}

// gz = gy * (y > 0)  (or a plain copy when there was no ReLU)
__global__ void relu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gz,
                                long n, int relu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    gz[i] = (!relu || y[i] > 0.f) ? gy[i] : 0.f;
}

}  // namespace bnn

using namespace bnn;

// defined in bbb_linear.hip: gx[S,B,K] = gz[S,B,N] . w_s  (w regenerated, transposed fragments)
extern "C" int bnn_bbb_input_grad_(const bnn_bbb_bwd_args* a, const float* gz, void* stream);

extern "C" size_t bnn_bbb_linear_bwd_workspace_bytes(int32_t n_samples, int32_t batch, int32_t out_features) {
  if (n_samples <= 0 || batch <= 0 || out_features <= 0) return 0;
  return (size_t)n_samples * batch * out_features * sizeof(float);
}

extern "C" int bnn_bbb_linear_bwd(const bnn_bbb_bwd_args* a, void* stream_) {
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_bbb_bwd_args)) return BNN_ERR_ABI;
  if (a->n_samples <= 0 || a->batch <= 0 || a->in_features <= 0 || a->out_features <= 0) return BNN_ERR_SHAPE;
  if (!a->x || !a->gy || !a->w_mu || !a->w_rho || !a->b_mu || !a->b_rho || !a->g_w_mu || !a->g_w_rho || !a->g_b_mu ||
      !a->g_b_rho)
    return BNN_ERR_NULL;
  if ((unsigned)a->eps_mode > 2u || (unsigned)a->prior.kind > 1u || (unsigned)a->math > 1u) return BNN_ERR_ENUM;
  if (a->eps_mode == BNN_EPS_MEMORY && (!a->eps_w || !a->eps_b)) return BNN_ERR_NULL;
  if (a->relu && !a->y) return BNN_ERR_NULL;
  if (!a->workspace || a->workspace_bytes < bnn_bbb_linear_bwd_workspace_bytes(a->n_samples, a->batch, a->out_features))
    return BNN_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(a->workspace) & 15) return BNN_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(a->x) | reinterpret_cast<uintptr_t>(a->w_mu) | reinterpret_cast<uintptr_t>(a->w_rho) |
       reinterpret_cast<uintptr_t>(a->g_w_mu) | reinterpret_cast<uintptr_t>(a->g_w_rho)) & 15)
    return BNN_ERR_ALIGN;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  // gz = gy masked by the ReLU the forward fused in; a layer without one (the output layer) reads gy itself
  // (the copy launch it used to get was ~5 us of a 0.2 ms training step)
  const float* gz = a->gy;
  hipError_t err = hipSuccess;
  if (a->relu) {
    float* gzw = reinterpret_cast<float*>(a->workspace);
    const long cnt = (long)a->n_samples * a->batch * a->out_features;
    long nb = (cnt + 255) / 256;
    nb = nb > 2048 ? 2048 : nb;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a->gy, a->y, gzw, cnt, 1);
    err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
    gz = gzw;
  }

  BwdK k;
  k.x = a->x;
  k.x_sstride = a->x_per_sample ? (long)a->batch * a->in_features : 0;
  k.gz = gz;
  k.w_mu = a->w_mu; k.w_rho = a->w_rho; k.b_mu = a->b_mu; k.b_rho = a->b_rho;
  k.eps_w = a->eps_w; k.eps_b = a->eps_b; k.glp = a->g_log_prior; k.glq = a->g_log_q;
  k.g_wmu = a->g_w_mu; k.g_wrho = a->g_w_rho; k.g_bmu = a->g_b_mu; k.g_brho = a->g_b_rho;
  k.S = a->n_samples; k.B = a->batch; k.K = a->in_features; k.N = a->out_features;
  k.eps_mode = a->eps_mode; k.prior_kind = a->prior.kind;
  k.k0 = (uint32_t)a->seed; k.k1 = (uint32_t)(a->seed >> 32);
  k.layer_id = a->layer_id; k.sample_offset = a->sample_offset; k.sample_counter = a->sample_counter;
#ifdef BNN_STAMPS
  {
    const char* v = getenv("BNN_HIP_DBG_PTR");
    k.dbg = v ? reinterpret_cast<unsigned long long*>(strtoull(v, nullptr, 0)) : nullptr;
  }
#endif
  k.inv_var_p = 0.f; k.a1 = k.a2 = k.inv2var1 = k.inv2var2 = k.invvar1 = k.invvar2 = 0.f;
  if (a->prior.kind == BNN_PRIOR_MIXTURE) {
    if (!(a->prior.sigma1 > 0.f) || !(a->prior.sigma2 > 0.f)) return BNN_ERR_SHAPE;
    const double s1 = a->prior.sigma1, s2 = a->prior.sigma2;
    k.a1 = (float)(a->prior.pi / s1);
    k.a2 = (float)((1.0 - a->prior.pi) / s2);
    k.inv2var1 = (float)(1.0 / (2.0 * s1 * s1));
    k.inv2var2 = (float)(1.0 / (2.0 * s2 * s2));
    k.invvar1 = (float)(1.0 / (s1 * s1));
    k.invvar2 = (float)(1.0 / (s2 * s2));
  } else {
    if (!(a->prior.sigma_p > 0.f)) return BNN_ERR_SHAPE;
    k.inv_var_p = (float)(1.0 / ((double)a->prior.sigma_p * a->prior.sigma_p));
  }
  if (a->g_x_bf16 && (!a->g_x || (reinterpret_cast<uintptr_t>(a->g_x_bf16) & 15))) return BNN_ERR_ALIGN;
  if (a->w_sampled && a->g_x && a->out_features <= 16 && a->batch <= 256 && (a->in_features & 7) == 0 &&
      a->eps_mode == BNN_EPS_PHILOX && a->math == BNN_MATH_BF16 &&
      !((reinterpret_cast<uintptr_t>(a->w_sampled) | reinterpret_cast<uintptr_t>(a->g_x)) & 15)) {
    // narrow output layer over the step's sampled weights: both gradients in one launch
    OutBwd o;
    o.w = reinterpret_cast<const __bf16*>(a->w_sampled);
    o.gx = a->g_x;
    o.gx16 = reinterpret_cast<__bf16*>(a->g_x_bf16);
    o.relu_mask = a->gx_relu_mask ? 1 : 0;
    o.wblocks = (a->in_features + 15) / 16;
    const long xthreads = (long)a->n_samples * a->batch * (a->in_features / 8);
    const long xblocks = (xthreads + 255) / 256;
    if (xblocks + o.wblocks < (1L << 30)) {
      if (a->batch > 112 && a->batch <= 128 && a->n_samples > 1)
        hipLaunchKernelGGL(bbb_out_layer_bwd_kernel<true>, dim3((unsigned)(o.wblocks + xblocks)), dim3(256), 0, stream, k, o);
      else
        hipLaunchKernelGGL(bbb_out_layer_bwd_kernel<false>, dim3((unsigned)(o.wblocks + xblocks)), dim3(256), 0, stream, k, o);
      err = hipGetLastError();
      return err == hipSuccess ? BNN_OK : (int)err;
    }
  }
  const int nblocks = ((a->in_features + 63) / 64) * ((a->out_features + 31) / 32);
  const dim3 grid((unsigned)(((nblocks + 7) / 8) * 8)), block(256);
  const bool mem = a->eps_mode == BNN_EPS_MEMORY;
  if ((a->in_features & 3) == 0) {
    if (mem) hipLaunchKernelGGL((bbb_bwd_weights_kernel<true, true>), grid, block, 0, stream, k);
    else hipLaunchKernelGGL((bbb_bwd_weights_kernel<true, false>), grid, block, 0, stream, k);
  } else {
    if (mem) hipLaunchKernelGGL((bbb_bwd_weights_kernel<false, true>), grid, block, 0, stream, k);
    else hipLaunchKernelGGL((bbb_bwd_weights_kernel<false, false>), grid, block, 0, stream, k);
  }
  err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  if (a->g_x) {
    const int rc = bnn_bbb_input_grad_(a, gz, stream_);
    if (rc != BNN_OK || !a->g_x_bf16 || a->w_sampled_t) return rc;     // (the matmul form over w_sampled_t writes the copy itself)
    return bnn_cast_bf16(a->g_x, a->g_x_bf16, nullptr, (int64_t)a->n_samples * a->batch * a->in_features, stream_);
  }
  return BNN_OK;
}
