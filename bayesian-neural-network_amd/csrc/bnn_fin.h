// ELBO finalize pieces shared by the standalone K4 kernel (reduce.hip) and the fused tail of
// the last layer's kernel (bbb_linear.hip): per-sample log p / log q (or KL) from the layers'
// stats partials plus the NLL of the sample's logits (reference networks.py:174-190).
#pragma once
#include "bnn_device.h"
#include "../../include/bnn_hip.h"
#include <math.h>

namespace bnn {

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
  int group;                     // MC samples per minibatch when the launch holds several (0: one evaluation)
  long tgt_stride;               // elements between the targets of consecutive minibatches (0: one target for all)
  const float* nll_partial;      // optional [S][nll_rb]: the NLL of row blocks, summed here instead of walking the logits
  int nll_rb;
};

// All transcendental constants of the priors, precomputed on the host in fp64.
struct FinC {
  double cnt_c0[8];      // count * c0                                  (log q constant)
  double lp_const[8];    // count * (c0 - log sigma_p)                  (Gaussian log p constant)
  double kl_const[8];    // 0.5 * (2 count log sigma_p - count)         (LR)
  double inv2var;        // 1 / (2 sigma_p^2)
  double reg_const;      // log(nll_sigma) - c0                          (regression NLL per element)
  double reg_inv2var;    // 1 / (2 nll_sigma^2)
};

constexpr int kFinNV = 25;            // 3 sums x 8 layers + nll
constexpr int kFinMaxWaves = 16;

static inline int make_fin(const bnn_finalize_args* a, FinK& k, FinC& cst) {
  if (!a) return BNN_ERR_NULL;
  if (a->struct_bytes != sizeof(bnn_finalize_args)) return BNN_ERR_ABI;
  if (a->n_layers < 0 || a->n_layers > 8 || a->n_samples <= 0 || a->n_samples > 65535) return BNN_ERR_SHAPE;
  if ((unsigned)a->prior.kind > 1u || (unsigned)a->nll_mode > 1u) return BNN_ERR_ENUM;
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
  if (a->group_samples < 0 || (a->group_samples > 0 && a->n_samples % a->group_samples != 0)) return BNN_ERR_SHAPE;
  if (a->target_per_group && a->group_samples <= 0) return BNN_ERR_SHAPE;
  if (a->nll) {
    if (!a->logits || !a->target) return BNN_ERR_NULL;
    if (a->batch <= 0 || a->classes <= 0) return BNN_ERR_SHAPE;
    if (a->nll_mode == BNN_NLL_REGRESSION && !(a->nll_sigma > 0.f)) return BNN_ERR_SHAPE;
  }
  k.n_layers = a->n_layers; k.local_reparam = a->local_reparam; k.S = a->n_samples; k.B = a->batch; k.C = a->classes;
  k.prior = a->prior; k.logits = a->logits; k.target = a->target; k.nll_mode = a->nll_mode;
  k.nll_sigma = a->nll_sigma; k.log_prior = a->log_prior; k.log_q = a->log_q; k.kl = a->kl; k.nll = a->nll;
  k.sample_counter = a->sample_counter; k.sample_counter_inc = a->sample_counter_inc;
  k.group = a->group_samples;
  k.nll_partial = nullptr; k.nll_rb = 0;
  k.tgt_stride = a->target_per_group
                     ? (a->nll_mode == BNN_NLL_CLASSIFICATION ? (long)a->batch : (long)a->batch * a->classes) : 0;
  const double c0 = -0.91893853320467274178;
  const double sp = (a->prior.kind == BNN_PRIOR_MIXTURE) ? 1.0 : (double)a->prior.sigma_p;
  for (int l = 0; l < 8; ++l) cst.cnt_c0[l] = cst.lp_const[l] = cst.kl_const[l] = 0.0;
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
  return BNN_OK;
}

// This thread's share of the NLL of rows [row0, row1) of sample s (networks.py:183-190): cross-entropy with
// reduction='sum', or -sum log N(target; out, sigma).  Summed over the block by the caller.
__device__ __forceinline__ float fin_nll(const FinK& p, const FinC& cst, int s, const float* lg, int ldc, int row0, int row1) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
    if (p.nll_mode == BNN_NLL_CLASSIFICATION) {
    const long long* tgt = reinterpret_cast<const long long*>(p.target) + (p.group > 0 ? (s / p.group) * p.tgt_stride : 0);
    if (p.C <= 32) {                           // a thread per row
      for (int b = row0 + (int)threadIdx.x; b < row1; b += blockDim.x) {
        const float* row = lg + (size_t)b * ldc;
        const long long tc = tgt[b];
        float mx, se = 0.f;
        if (ldc == 16 && p.C <= 16) {
          // a row of the LDS tile [rows][16] held in registers: lane b starts at column b so that a wave's 64
          // reads spread over all 32 banks (walking the same column put them on two); hardware exp2 / log2
          float vr[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) vr[i] = row[(i + b) & 15];
          mx = -3.0e38f;
#pragma unroll
          for (int i = 0; i < 16; ++i) mx = (((i + b) & 15) < p.C) ? fmaxf(mx, vr[i]) : mx;
#pragma unroll
          for (int i = 0; i < 16; ++i) se += (((i + b) & 15) < p.C) ? __expf(vr[i] - mx) : 0.f;
          se = __logf(se);
        } else if (p.C <= 16) {
          // a row in global memory: all of its loads issued at once (a rolled loop waits for each in turn)
          float vr[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) vr[i] = row[min(i, p.C - 1)];
          mx = vr[0];
#pragma unroll
          for (int i = 1; i < 16; ++i) mx = i < p.C ? fmaxf(mx, vr[i]) : mx;
#pragma unroll
          for (int i = 0; i < 16; ++i) se += i < p.C ? __expf(vr[i] - mx) : 0.f;
          se = __logf(se);
        } else {
          mx = row[0];
          for (int cc = 1; cc < p.C; ++cc) mx = fmaxf(mx, row[cc]);
          for (int cc = 0; cc < p.C; ++cc) se += expf(row[cc] - mx);
          se = logf(se);
        }
        const float picked = (tc >= 0 && tc < p.C) ? row[tc] : __builtin_nanf("");   // bad label: NaN loss, loudly
        acc += (mx + se) - picked;
      }
    } else {                                   // a wave per row, lanes stride over the classes
      const int nwv = blockDim.x >> 6;
      for (int b = row0 + wave; b < row1; b += nwv) {
        const float* row = lg + (size_t)b * ldc;
        float mx = -3.0e38f;
#pragma unroll 8
        for (int cc = lane; cc < p.C; cc += 64) mx = fmaxf(mx, row[cc]);     // 8 loads in flight
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float se = 0.f;
#pragma unroll 8
        for (int cc = lane; cc < p.C; cc += 64) se += __expf(row[cc] - mx);
        se = wave_sum(se);
        if (lane == 0) {
          const long long tc = tgt[b];
          const float picked = (tc >= 0 && tc < p.C) ? row[tc] : __builtin_nanf("");   // bad label: NaN loss, loudly
          acc += (mx + logf(se)) - picked;
        }
      }
    }
  } else {
    const float* tgt = reinterpret_cast<const float*>(p.target) + (p.group > 0 ? (s / p.group) * p.tgt_stride : 0);
    if (p.C <= 8) {                            // a thread per row (the 1-output regression net)
      for (int b = row0 + (int)threadIdx.x; b < row1; b += blockDim.x)
        for (int cc = 0; cc < p.C; ++cc) {
          const float d = tgt[(size_t)b * p.C + cc] - lg[(size_t)b * ldc + cc];
          acc += (float)((double)(d * d) * cst.reg_inv2var + cst.reg_const);
        }
    } else {                                   // wide outputs: a wave per row, lanes stride over the outputs
      // sum of squares in fp32 per lane (<= B*C/threads terms), the affine map applied once per lane
      const int nwv = blockDim.x >> 6;
      float d2 = 0.f;
      int cnt = 0;
      for (int b = row0 + wave; b < row1; b += nwv) {
        const float* row = lg + (size_t)b * ldc;
        const float* trow = tgt + (size_t)b * p.C;
        if (((p.C | ldc) & 3) == 0 && ((reinterpret_cast<uintptr_t>(row) | reinterpret_cast<uintptr_t>(trow)) & 15) == 0) {
#pragma unroll 4
          for (int cc = lane * 4; cc < p.C; cc += 256) {                    // 16-byte loads, 4 of each stream in flight
            const float4 t4 = *reinterpret_cast<const float4*>(trow + cc), r4 = *reinterpret_cast<const float4*>(row + cc);
            const float dx = t4.x - r4.x, dy = t4.y - r4.y, dz = t4.z - r4.z, dw = t4.w - r4.w;
            d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, __builtin_fmaf(dz, dz, __builtin_fmaf(dw, dw, d2))));
            cnt += 4;
          }
        } else {
#pragma unroll 8
          for (int cc = lane; cc < p.C; cc += 64) {                         // 8 loads of each stream in flight
            const float d = trow[cc] - row[cc];
            d2 = __builtin_fmaf(d, d, d2);
            ++cnt;
          }
        }
      }
      acc = (float)((double)d2 * cst.reg_inv2var + (double)cnt * cst.reg_const);
    }
  }
  return acc;
}

// ELBO scalars of sample s, valid in thread 0 on return.
//   lg / ldc : the sample's logits [B][ldc] (global memory or LDS);
//   T[]      : tile counts of the layers' workspaces (ws[l][0].x), 0 for own_layer;
//   own_layer: a layer whose three partial sums thread 0 supplies directly (own0..2) because
//              this very block produced them (-1: none);
//   part     : LDS scratch, 8-byte aligned, >= max((blockDim.x/64) * kFinNV floats, kFinNV doubles).
// Partial sums go through fp32 wave shuffles (each thread holds at most a few partials) and
// fp64 only across waves; one barrier.
__device__ __forceinline__ void fin_sample(const FinK& p, const FinC& cst, int s, const int T[8], const float* lg, int ldc,
                                           int own_layer, float own0, float own1, float own2, float* part,
                                           float& out_a, float& out_b, float& out_nll, unsigned long long* dbg = nullptr) {
// diagnostic build (-DBNN_STAMPS, tools/stamps_final.py) passes a stamp buffer; production callers pass none and
// the stamps fold away
#define FIN_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
  constexpr int NV = kFinNV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.f;
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    if (l < p.n_layers && l != own_layer) {
      const float4* ws = reinterpret_cast<const float4*>(p.ws[l]);
      const int Tl = T[l], bd = blockDim.x;
      // eight entries per thread in flight (clamped re-reads are not added; the order of a thread's additions is unchanged): a
      // wide layer's sampling launch leaves 4096 entries per sample, 16 per thread -- as `v += ws[..]` in a plain loop that was
      // one memory round trip per entry, 34 us of a 4-sample evaluation of the 4096-wide network
      constexpr int FU = 8;
      for (int t0 = threadIdx.x; t0 < Tl; t0 += FU * bd) {
        float4 qv[FU];
        float lsz[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const int t = min(t0 + u * bd, Tl - 1);
          qv[u] = p.local_reparam ? ws[1 + t] : ws[1 + (size_t)s * Tl + t];
          lsz[u] = p.local_reparam ? 0.f : ws[1 + t].z;
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          if (t0 + u * bd < Tl) {
            v[3 * l + 0] += qv[u].x;                               // LR: sum log sigma      | BBB: sum eps^2
            v[3 * l + 1] += qv[u].y;                               //     sum sigma^2        |      sum w^2 | sum log p_mix
            v[3 * l + 2] += p.local_reparam ? qv[u].z : lsz[u];    //     sum mu^2           |      sum log sigma (stored with sample 0)
          }
        }
      }
    }
  }
  FIN_STAMP(12);
  if (p.nll && p.nll_partial) {               // row blocks already reduced by nll_rows_kernel: add them up
    float acc = 0.f;
    for (int t = threadIdx.x; t < p.nll_rb; t += blockDim.x) acc += p.nll_partial[(size_t)s * p.nll_rb + t];
    v[NV - 1] = acc;
  } else if (p.nll && lg) {
    v[NV - 1] = fin_nll(p, cst, s, lg, ldc, 0, p.B);
  }
  FIN_STAMP(13);
  const int nv = 3 * p.n_layers;
  if (nv <= 9) {                               // up to 3 layers (block-uniform): ten sums as straight-line code, so
    float t[10];                                // their DPP chains interleave instead of running one per basic block
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = wave_sum(v[i]);
    t[9] = wave_sum(v[NV - 1]);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) part[wave * NV + i] = t[i];
      part[wave * NV + NV - 1] = t[9];
    }
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < nv || i == NV - 1) {             // block-uniform
        const float t = wave_sum(v[i]);
        if (lane == 0) part[wave * NV + i] = t;
      }
    }
  }
  FIN_STAMP(14);
  __syncthreads();
  // cross-wave sums in parallel: thread i folds value i over the waves (one thread walking all of them was
  // ~70 dependent LDS reads, 3 us by the stamps), then hands the doubles to thread 0 through the same scratch
  double folded = 0;
  {
    const int nwv = blockDim.x >> 6;
    if ((int)threadIdx.x < NV && ((int)threadIdx.x < nv || (int)threadIdx.x == NV - 1))
      for (int w = 0; w < nwv; ++w) folded += (double)part[w * NV + threadIdx.x];
  }
  __syncthreads();                                          // every read of the per-wave partials is done
  if ((int)threadIdx.x < NV) reinterpret_cast<double*>(part)[threadIdx.x] = folded;
  __syncthreads();
  FIN_STAMP(15);
  if (threadIdx.x == 0) {
    auto red = [&](int i) { return reinterpret_cast<const double*>(part)[i]; };
    // per-layer fp32 rounding, then fp32 adds, as the reference sums l1 + l2 + l3
    // (networks.py:174-181)
    float a_tot = 0.f, b_tot = 0.f;
#pragma unroll                                   // static indices into the constants: their loads batch, the layers' fp64 chains overlap
    for (int l = 0; l < 8; ++l) {
      if (l >= p.n_layers) break;
      double r0, r1, r2;
      if (l == own_layer) {
        r0 = own0; r1 = own1; r2 = own2;
      } else {
        r0 = red(3 * l); r1 = red(3 * l + 1); r2 = red(3 * l + 2);
      }
      if (p.local_reparam) {
        a_tot += (float)(cst.kl_const[l] - r0 + (r1 + r2) * cst.inv2var);
      } else {
        const double lq = cst.cnt_c0[l] - r2 - 0.5 * r0;
        const double lp = (p.prior.kind == BNN_PRIOR_GAUSS) ? cst.lp_const[l] - r1 * cst.inv2var : r1;
        a_tot += (float)lp;
        b_tot += (float)lq;
      }
    }
    out_a = a_tot;
    out_b = b_tot;
    out_nll = (float)red(NV - 1);
  }
}

// The per-sample scalars of samples [0, S) folded in sample order into one (a, b, nll) triple per group of `g` consecutive
// samples, `emit(group, a, b, nll)` called as each group completes.  The loads of sixteen samples are all in flight before the
// first is consumed, whatever the group size: a `for` over `x += atomic_load(...)` compiles to one memory round trip per
// sample (s_waitcnt vmcnt(0) in the loop, ~0.6 us each on the one thread everybody else has already left to), and a branch
// around a load is closed with vmcnt(0) too -- hence the stand-in pointers for absent tensors.
template <typename Emit>
__device__ __forceinline__ void fin_fold_groups(const float* pa, const float* pb, const float* pn, int S, int g, Emit emit) {
  const float* fb = pa ? pa : pb ? pb : pn;                    // an absent tensor reads another's words (not added)
  double ta = 0, tb = 0, tn = 0;
  int left = g, group = 0;
  constexpr int CH = 16;
  for (int base = 0; base < S; base += CH) {
    float va[CH], vb[CH], vn[CH];
    if (fb) {
      const float* qa = pa ? pa : fb;
      const float* qb = pb ? pb : fb;
      const float* qn = pn ? pn : fb;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int i = min(base + j, S - 1);                    // clamped: a re-read, not added
        va[j] = __hip_atomic_load(qa + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vb[j] = __hip_atomic_load(qb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vn[j] = __hip_atomic_load(qn + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) va[j] = vb[j] = vn[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (base + j < S) {
        ta += pa ? va[j] : 0.f;
        tb += pb ? vb[j] : 0.f;
        tn += pn ? vn[j] : 0.f;
        if (--left == 0) {
          emit(group, ta, tb, tn);
          ta = tb = tn = 0;
          left = g;
          ++group;
        }
      }
    }
  }
}

// The 4-vector(s) of sums from the per-sample scalars, in sample order (one thread; the callers make the scalars
// of all samples visible first): one vector, or one per minibatch of `group` samples.
__device__ __forceinline__ void fin_fold_sums(const FinK& p, float* sums) {
  const int g = p.group > 0 ? p.group : p.S;
  const float* pa = p.local_reparam ? p.kl : p.log_prior;
  const float* pb = (!p.local_reparam && p.log_q) ? p.log_q : nullptr;
  fin_fold_groups(pa, pb, p.nll, p.S, g, [&](int m, double ta, double tb, double tn) {
    float* so = sums + 4 * m;
    so[0] = (float)ta; so[1] = (float)tb; so[2] = (float)tn; so[3] = (float)g;
  });
}

__device__ __forceinline__ void fin_store(const FinK& p, int s, float a, float b, float nll) {
  if (p.local_reparam) {
    if (p.kl) p.kl[s] = a;
  } else {
    if (p.log_prior) p.log_prior[s] = a;
    if (p.log_q) p.log_q[s] = b;
  }
  if (p.nll) p.nll[s] = nll;
}

// The training step's tail riding on the row-split final kernels K1r / K3r (bnn_loss_args): each row block
// differentiates its rows' NLL (elbo_loss_nll_bwd_kernel's arithmetic), the block that finishes the evaluation assembles
// the loss and the seeds (elbo_loss_block's arithmetic, fp64 sums in sample order).  out4 == nullptr: no tail.
struct FinLoss {
  const float* beta;
  float total, grad_scale, inv_var;
  float* out4;
  float* g_a;
  float* g_b;
  float* g_kl3;
  float* g_logits;
};

__device__ __forceinline__ void fin_loss_assemble(const FinK& fk, const FinLoss& tr) {
  const float* pa = fk.local_reparam ? fk.kl : fk.log_prior;
  double x = 0, y = 0, z = 0;
  fin_fold_groups(pa, fk.local_reparam ? nullptr : fk.log_q, fk.nll, fk.S, fk.S, [&](int, double ta, double tb, double tn) { x = ta; y = tb; z = tn; });
  const float beta = *tr.beta;
  const float inv = tr.grad_scale / tr.total;
  for (int i = 0; i < fk.S; ++i) {
    if (tr.g_a) tr.g_a[i] = fk.local_reparam ? 0.f : -beta * inv;
    if (tr.g_b) tr.g_b[i] = beta * inv;
  }
  const float am = (float)x / tr.total, bm = (float)y / tr.total, nm = (float)z / tr.total;
  tr.out4[0] = fk.local_reparam ? beta * am + nm : beta * bm - beta * am + nm;
  tr.out4[1] = am;
  tr.out4[2] = bm;
  tr.out4[3] = nm;
  if (tr.g_kl3) { tr.g_kl3[0] = beta * tr.grad_scale; tr.g_kl3[1] = 0.f; tr.g_kl3[2] = 0.f; }
}

// d (summed NLL of row `brow` of sample s) / d logits, scaled by grad_scale / total, from the block's logits in LDS
// (`lgrow`: the row's C logits)
__device__ __forceinline__ void fin_loss_row_grad(const FinK& fk, const FinLoss& tr, int s, int B, int brow, const float* lgrow) {
  const int C = fk.C;
  float* go = tr.g_logits + ((size_t)s * B + brow) * C;
  const float gs = tr.grad_scale / tr.total;
  if (fk.nll_mode == BNN_NLL_CLASSIFICATION) {
    const long long tc = reinterpret_cast<const long long*>(fk.target)[brow];
    float mx = lgrow[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, lgrow[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(lgrow[c] - mx);
    const float inv = (tc >= 0 && tc < C) ? 1.0f / se : __builtin_nanf("");   // bad label: NaN, as in nll_bwd_kernel
    for (int c = 0; c < C; ++c) go[c] = (expf(lgrow[c] - mx) * inv - (c == tc ? 1.f : 0.f)) * gs;
  } else {
    const float* tg = reinterpret_cast<const float*>(fk.target) + (size_t)brow * C;
    for (int c = 0; c < C; ++c) go[c] = (lgrow[c] - tg[c]) * tr.inv_var * gs;
  }
}

static inline FinLoss make_fin_loss(const bnn_finalize_args* f) {
  FinLoss tr{};
  if (f->loss) {
    const bnn_loss_args* t = f->loss;
    tr.beta = t->beta; tr.total = t->total_samples; tr.grad_scale = t->grad_scale;
    tr.inv_var = f->nll_mode == BNN_NLL_REGRESSION ? (float)(1.0 / ((double)f->nll_sigma * f->nll_sigma)) : 0.f;
    tr.out4 = t->out4; tr.g_a = t->g_a; tr.g_b = t->g_b; tr.g_kl3 = t->g_kl3; tr.g_logits = t->g_logits;
  }
  return tr;
}

// validation of f->loss for the launch functions that honour it
static inline int check_fin_loss(const bnn_finalize_args* f) {
  if (!f->loss) return BNN_OK;
  const bnn_loss_args* t = f->loss;
  if (!t->beta || !t->out4 || !t->g_logits || !f->logits || !f->target || !f->nll || !(t->total_samples > 0.f)) return BNN_ERR_NULL;
  if (f->group_samples > 0) return BNN_ERR_SHAPE;
  if (f->local_reparam ? !f->kl : (!f->log_prior || !f->log_q)) return BNN_ERR_SHAPE;
  return BNN_OK;
}

}  // namespace bnn
