// Device side of K1s (bnn_bbb_sample_weights): the sampling half of BayesianLinear.forward as a block-level device
// function, shared by the stand-alone kernel (bbb_sample.hip) and by the layer launches of bbb_linear.hip that carry
// another layer's sampling as extra blocks (the output layer's weights ride on the launch of the layer before it).
#pragma once
#include "bnn_device.h"
#include "../../include/bnn_hip.h"
#include <math.h>

namespace bnn {

constexpr int kSampleThreads = 256;                   // threads of a block that sample (a wider block's other threads idle)
constexpr int kSampleOctets = 2 * kSampleThreads;     // octets per block
constexpr int kSampleGroup = 4;                       // samples per block: one read of (mu, rho), one softplus, for all of them
constexpr int kSampleRedFloats = (kSampleThreads / 64) * 3 * kSampleGroup;   // LDS scratch a block needs (floats)

struct SampleL {
  const float* w_mu;
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  __bf16* w_out;
  float* b_out;
  float4* ws;
  int K, N;
  uint32_t layer_id;
  int first_block;      // of the layer
  int T;                // chunks (= statistics entries) per sample; a block = one chunk x one group of kSampleGroup samples
  int bias_per_block;   // block `chunk` also samples biases [chunk * bpb, (chunk + 1) * bpb)
  int prior_kind;
  float inv2var1, c1, inv2var2, c2, pi;
};

struct SampleK {
  SampleL L[BNN_SAMPLE_MAX_LAYERS];
  int n_layers, S;
  uint32_t k0, k1, sample_offset;
  const uint32_t* sample_counter;
  uint32_t sgrp, sgrp_stride;   // sample groups (bnn_bbb_sample_args.sample_group): 0 = none
  const float* cast_src;   // optional rider: cast_dst[i] = bf16(cast_src[i]), the evaluation's input batch
  __bf16* cast_dst;
  long cast_n;
  int cast_first;          // first block of the cast job (after every layer's blocks)
};

// the mixture density (argument of the log; accumulated with add_log: explicit contraction, see bnn_device.h)
struct SampleMix { float inv2var1, c1, inv2var2, c2, pi; };
__device__ __forceinline__ float sample_mix_p(const SampleMix& L, float w) {
  const float w2 = w * w;
  const float p1 = fast_exp(__builtin_fmaf(-w2, L.inv2var1, L.c1));
  const float p2 = fast_exp(__builtin_fmaf(-w2, L.inv2var2, L.c2));
  return __builtin_fmaf(L.pi, p1, (1.0f - L.pi) * p2);
}

// One block of the sampling job: `block` = index within the job's own block range, `red` = LDS scratch of
// >= kSampleRedFloats floats.  Every thread of the block must call it (one barrier inside); threads beyond
// kSampleThreads take no octet.
// A block owns one 512-octet chunk of a layer for a GROUP of up to kSampleGroup samples: (mu, rho) are read and
// softplus is taken once for the group (the pass was bound by its 8 B read + 2 B written per weight AND sample: 98 us per
// 4 samples of a 4096 x 4096 layer at 0.85 of the HBM figure; per group of four it moves 16 B per weight instead of 40).
// Each sample's epsilon, sums and stores are what a one-sample block produced: same statistics entry (sample, chunk),
// same bits.
__device__ __forceinline__ void sample_block(const SampleK& p, int block, float* red) {
  if (block >= p.cast_first) {                             // rider: fp32 -> bf16 of the input batch, 8 per thread
    if ((int)threadIdx.x >= kSampleThreads) return;        // (block-uniform branch above: no barrier on this path)
    const long i = ((long)(block - p.cast_first) * kSampleThreads + threadIdx.x) * 8;
    if (i + 7 < p.cast_n && (p.cast_n & 7) == 0) {
      const float4 a = *reinterpret_cast<const float4*>(p.cast_src + i), b = *reinterpret_cast<const float4*>(p.cast_src + i + 4);
      bf16x8 o;
      o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
      o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
      *reinterpret_cast<bf16x8*>(p.cast_dst + i) = o;
    } else {
      for (long j = i; j < i + 8 && j < p.cast_n; ++j) p.cast_dst[j] = (__bf16)p.cast_src[j];
    }
    return;
  }
  int l = 0;
#pragma unroll 1
  while (l + 1 < p.n_layers && block >= p.L[l + 1].first_block) ++l;
  const SampleL& L = p.L[l];
  const int local = block - L.first_block;
  const int grp = local / L.T, chunk = local - grp * L.T;
  const int s0 = grp * kSampleGroup;
  const int ns = min(kSampleGroup, p.S - s0);              // samples of this block (block-uniform)
  const int K = L.K, N = L.N;
  // the layer's prior in registers, and ONE branch on its kind per octet: read per weight from the parameter block (the
  // layer index is a run-time value) the kind cost a scalar load, a wait and a branch for each of the 16 weights of a sample
  const bool gauss = L.prior_kind == BNN_PRIOR_GAUSS;      // block-uniform
  const SampleMix mix = {L.inv2var1, L.c1, L.inv2var2, L.c2, L.pi};
  const int opr = K >> 3;                                  // octets per row
  const long total = (long)N * opr;
  const uint32_t gs_base = p.sample_offset + (p.sample_counter ? *p.sample_counter : 0u);
  const uint32_t gpr = (uint32_t)(K >> 2);
  const uint32_t wid = L.layer_id * 4u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool active = (int)threadIdx.x < kSampleThreads;
  if (!active) {                                          // whole waves (kSampleThreads is a multiple of 64): a wider block's
    __syncthreads();                                      // extra waves only meet the barrier
    return;
  }

  // ---- all parameter loads of the thread's two octets first (clamped addresses: no load under a branch)
  long o[2];
  int n[2], k[2];
  float4 m[2][2], r[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    o[u] = (long)chunk * kSampleOctets + u * kSampleThreads + threadIdx.x;
    const long oc = o[u] < total ? o[u] : total - 1;
    n[u] = (int)(oc / opr);
    k[u] = (int)(oc - (long)n[u] * opr) << 3;
    const size_t off = (size_t)n[u] * K + k[u];
    m[u][0] = *reinterpret_cast<const float4*>(L.w_mu + off);
    m[u][1] = *reinterpret_cast<const float4*>(L.w_mu + off + 4);
    r[u][0] = *reinterpret_cast<const float4*>(L.w_rho + off);
    r[u][1] = *reinterpret_cast<const float4*>(L.w_rho + off + 4);
  }
  // this block's share of the biases: one per thread of the first few threads
  const int bn = chunk * L.bias_per_block + (int)threadIdx.x;
  const bool has_bias = (int)threadIdx.x < L.bias_per_block && bn < N;
  float bmu = 0.f, brho = 0.f;
  if (has_bias) {
    bmu = L.b_mu[bn];
    brho = L.b_rho[bn];
  }
  __builtin_amdgcn_sched_barrier(0);                      // the loads stay one batch ahead of the generator work

  // ---- once per group: sigma (and, for the group that holds sample 0, the sum of log sigma)
  float mu[2][8], sg[2][8];
  float ls_oct[2] = {0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float mm[8] = {m[u][0].x, m[u][0].y, m[u][0].z, m[u][0].w, m[u][1].x, m[u][1].y, m[u][1].z, m[u][1].w};
    const float rh[8] = {r[u][0].x, r[u][0].y, r[u][0].z, r[u][0].w, r[u][1].x, r[u][1].y, r[u][1].z, r[u][1].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[u][j] = mm[j];
      sg[u][j] = softplus(rh[j]);
    }
    if (s0 == 0) {                                         // block-uniform
#pragma unroll
      for (int j = 0; j < 8; ++j) ls_oct[u] = add_log(ls_oct[u], sg[u][j]);
    }
  }
  const float bsg = has_bias ? softplus(brho) : 0.f;

#pragma unroll 1
  for (int si = 0; si < ns; ++si) {
    const int s = s0 + si;
    uint32_t gs = gs_base;
    if (p.sgrp == 0u) gs += (uint32_t)s;
    else gs += ((uint32_t)s / p.sgrp) * p.sgrp_stride + (uint32_t)s % p.sgrp;
    const bool do_ls = s == 0;
    float s_e2 = 0.f, s_a = 0.f, s_ls = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool ok = o[u] < total;
      const uint32_t g = (uint32_t)n[u] * gpr + (uint32_t)(k[u] >> 2);
      float e[8];
      philox_normal4(g, gs, wid, p.k0, p.k1, e);
      philox_normal4(g + 1u, gs, wid, p.k0, p.k1, e + 4);
      float w[8], e2 = 0.f, a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        w[j] = __builtin_fmaf(sg[u][j], e[j], mu[u][j]);
        e2 = __builtin_fmaf(e[j], e[j], e2);
      }
      if (gauss) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a = __builtin_fmaf(w[j], w[j], a);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) a = add_log(a, sample_mix_p(mix, w[j]));
      }
      s_e2 += ok ? e2 : 0.f;
      s_a += ok ? a : 0.f;
      s_ls += (ok && do_ls) ? ls_oct[u] : 0.f;
      if (ok) {
        bf16x8 wb;
#pragma unroll
        for (int j = 0; j < 8; ++j) wb[j] = (__bf16)w[j];
        *reinterpret_cast<bf16x8*>(L.w_out + ((size_t)s * N + n[u]) * K + k[u]) = wb;
      }
    }
    if (has_bias) {
      float e4[4];
      philox_normal4((uint32_t)(bn >> 2), gs, wid + 1u, p.k0, p.k1, e4);
      const float e = (bn & 3) == 0 ? e4[0] : (bn & 3) == 1 ? e4[1] : (bn & 3) == 2 ? e4[2] : e4[3];
      const float b = __builtin_fmaf(bsg, e, bmu);
      L.b_out[(size_t)s * N + bn] = b;
      s_e2 = __builtin_fmaf(e, e, s_e2);
      s_a = gauss ? __builtin_fmaf(b, b, s_a) : add_log(s_a, sample_mix_p(mix, b));
      if (do_ls) s_ls = add_log(s_ls, bsg);
    }
    const float a0 = wave_sum(s_e2), a1 = wave_sum(s_a), a2 = wave_sum(s_ls);
    if (lane == 0) {
      float* rd = red + (si * (kSampleThreads / 64) + wave) * 3;
      rd[0] = a0;
      rd[1] = a1;
      rd[2] = a2;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < ns) {                            // thread si folds sample s0 + si: the four waves in wave order
    const int si = threadIdx.x;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int wv = 0; wv < kSampleThreads / 64; ++wv) {
      const float* rd = red + (si * (kSampleThreads / 64) + wv) * 3;
      t0 += rd[0];
      t1 += rd[1];
      t2 += rd[2];
    }
    L.ws[1 + (size_t)(s0 + si) * L.T + chunk] = make_float4(t0, t1, t2, 0.f);
    if (local == 0 && si == 0) L.ws[0] = make_float4(__int_as_float(L.T), 0.f, 0.f, 0.f);
  }
}

}  // namespace bnn

// ---- host side
extern "C" size_t bnn_bbb_sample_workspace_bytes(int32_t n_samples, int32_t in_features, int32_t out_features);

static inline int sample_chunks(int K, int N) {
  using bnn::kSampleOctets;
  const long octets = (long)N * (K >> 3);
  return (int)((octets + kSampleOctets - 1) / kSampleOctets);
}


// Validate `a` and fill the kernel parameter block; `blocks` = blocks the job needs.  Returns a bnn_status.
static inline int fill_sample(const bnn_bbb_sample_args* a, bnn::SampleK& k, long& blocks) {
  using namespace bnn;
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_bbb_sample_args)) return BNN_ERR_ABI;
  if (a->n_layers <= 0 || a->n_layers > BNN_SAMPLE_MAX_LAYERS || a->n_samples <= 0) return BNN_ERR_SHAPE;
  blocks = 0;
  const double c0 = -0.91893853320467274178;
  for (int i = 0; i < a->n_layers; ++i) {
    const bnn_bbb_sample_layer& l = a->layer[i];
    if (l.in_features <= 0 || l.out_features <= 0 || (l.in_features & 7)) return BNN_ERR_SHAPE;
    if (!l.w_mu || !l.w_rho || !l.b_mu || !l.b_rho || !l.w_out || !l.b_out || !l.workspace) return BNN_ERR_NULL;
    if ((unsigned)l.prior.kind > 1u) return BNN_ERR_ENUM;
    const uintptr_t al = reinterpret_cast<uintptr_t>(l.w_mu) | reinterpret_cast<uintptr_t>(l.w_rho) |
                         reinterpret_cast<uintptr_t>(l.w_out) | reinterpret_cast<uintptr_t>(l.workspace);
    if (al & 15) return BNN_ERR_ALIGN;
    if (l.workspace_bytes < bnn_bbb_sample_workspace_bytes(a->n_samples, l.in_features, l.out_features))
      return BNN_ERR_WORKSPACE;
    SampleL& o = k.L[i];
    o.w_mu = l.w_mu; o.w_rho = l.w_rho; o.b_mu = l.b_mu; o.b_rho = l.b_rho;
    o.w_out = reinterpret_cast<__bf16*>(l.w_out); o.b_out = l.b_out; o.ws = reinterpret_cast<float4*>(l.workspace);
    o.K = l.in_features; o.N = l.out_features; o.layer_id = l.layer_id;
    o.first_block = (int)blocks;
    o.T = sample_chunks(l.in_features, l.out_features);
    o.bias_per_block = (l.out_features + o.T - 1) / o.T;
    if (o.bias_per_block > kSampleThreads) return BNN_ERR_SHAPE;     // in_features < 8 per 256 outputs: not a layer shape
    o.prior_kind = l.prior.kind;
    o.pi = l.prior.pi;
    if (l.prior.kind == BNN_PRIOR_MIXTURE) {
      if (!(l.prior.sigma1 > 0.f) || !(l.prior.sigma2 > 0.f)) return BNN_ERR_SHAPE;
      o.inv2var1 = (float)(1.0 / (2.0 * (double)l.prior.sigma1 * l.prior.sigma1));
      o.inv2var2 = (float)(1.0 / (2.0 * (double)l.prior.sigma2 * l.prior.sigma2));
      o.c1 = (float)(c0 - log((double)l.prior.sigma1));
      o.c2 = (float)(c0 - log((double)l.prior.sigma2));
    } else {
      if (!(l.prior.sigma_p > 0.f)) return BNN_ERR_SHAPE;
      o.inv2var1 = o.inv2var2 = o.c1 = o.c2 = 0.f;
    }
    blocks += (long)o.T * ((a->n_samples + kSampleGroup - 1) / kSampleGroup);
    if (blocks > 0x3fffffff) return BNN_ERR_SHAPE;
  }
  for (int i = a->n_layers; i < BNN_SAMPLE_MAX_LAYERS; ++i) {
    k.L[i] = k.L[0];
    k.L[i].first_block = (int)blocks;
  }
  k.n_layers = a->n_layers; k.S = a->n_samples;
  k.k0 = (uint32_t)a->seed; k.k1 = (uint32_t)(a->seed >> 32);
  k.sample_offset = a->sample_offset; k.sample_counter = a->sample_counter;
  k.sgrp = a->sample_group; k.sgrp_stride = a->sample_group_stride;
  k.cast_src = a->cast_src; k.cast_dst = reinterpret_cast<__bf16*>(a->cast_dst); k.cast_n = (long)a->cast_n;
  k.cast_first = (int)blocks;
  if (a->cast_n > 0) {
    if (!a->cast_src || !a->cast_dst) return BNN_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(a->cast_src) | reinterpret_cast<uintptr_t>(a->cast_dst)) & 15) return BNN_ERR_ALIGN;
    blocks += (a->cast_n + kSampleThreads * 8 - 1) / (kSampleThreads * 8);
    if (blocks > 0x3fffffff) return BNN_ERR_SHAPE;
  } else if (a->cast_n < 0) {
    return BNN_ERR_SHAPE;
  }
    return BNN_OK;
}
