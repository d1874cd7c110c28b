// Device-side building blocks shared by the gfx950 kernels: Philox4x32 + Box-Muller
// (the frozen epsilon map of include/bnn_hip.h), fp32 softplus/log on the hardware
// transcendental units, wave64 and block reductions, MFMA fragment types.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bnn_hip.h"

namespace bnn {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int kWave = 64;
constexpr float kC0 = -0.918938533204672742f;  // -log(sqrt(2 pi)), networks.py:46
constexpr float kLn2 = 0.693147180559945309f;
constexpr float kLog2e = 1.442695040888963407f;

// ---------------------------------------------------------------------------- Philox
// One round = two 32x32->64 multiplies (v_mad_u64_u32 / v_mul_hi_u32 + v_mul_lo_u32) and
// four xors; the key schedule is wave-uniform and lives on the scalar unit.
template <int ROUNDS = BNN_PHILOX_ROUNDS>
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    const uint64_t p0 = (uint64_t)c.x * 0xD2511F53u;
    const uint64_t p1 = (uint64_t)c.z * 0xCD9E8D57u;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// uint32 -> (0,1]: one v_cvt_f32_u32 + one v_fma_f32.
__device__ __forceinline__ float u01(uint32_t r) {
  return __builtin_fmaf((float)r, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

// Box-Muller on the transcendental unit: v_log_f32 is log2, v_sin/v_cos_f32 take their
// argument in revolutions, so 2*pi*u needs no multiply.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = u01(a), u2 = u01(b);
  const float rad = __builtin_amdgcn_sqrtf(-2.0f * kLn2 * __builtin_amdgcn_logf(u1));
  n0 = rad * __builtin_amdgcn_cosf(u2);
  n1 = rad * __builtin_amdgcn_sinf(u2);
}

// The four N(0,1) values of epsilon group `group` (see include/bnn_hip.h).
__device__ __forceinline__ void philox_normal4(uint32_t group, uint32_t gsample, uint32_t tensor_id,
                                               uint32_t k0, uint32_t k1, float out[4]) {
  const uint4 r = philox4x32<>(make_uint4(group, gsample, tensor_id, 0u), k0, k1);
  box_muller(r.x, r.y, out[0], out[1]);
  box_muller(r.z, r.w, out[2], out[3]);
}

// Two consecutive epsilon groups (`group`, `group + 1`: the 8 weights a lane of the block forms holds per k-step) with
// the two Philox calls in LOCK-STEP.  A call is a chain of ROUNDS dependent (multiply -> xor -> xor -> multiply ...) rounds
// with two independent multiplies each; written one call after the other the compiler keeps them apart (the k-step's
// critical path is then 2 x ROUNDS rounds deep, and three or four waves per SIMD do not cover it: the counters show the
// waves issue-stalled a third of the time).  Interleaved round by round the chain is ROUNDS deep with four independent
// multiplies per level.  Same arithmetic, same bits.
// (in two halves, so that a kernel can place other work -- an LDS-DMA piece -- between the integer and the transcendental part)
template <int ROUNDS = BNN_PHILOX_ROUNDS>
__device__ __forceinline__ void philox_pair(uint32_t group, uint32_t gsample, uint32_t tensor_id, uint32_t k0, uint32_t k1, uint4& a, uint4& b) {
  a = make_uint4(group, gsample, tensor_id, 0u);
  b = make_uint4(group + 1u, gsample, tensor_id, 0u);
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    const uint64_t pa0 = (uint64_t)a.x * 0xD2511F53u, pb0 = (uint64_t)b.x * 0xD2511F53u;
    const uint64_t pa1 = (uint64_t)a.z * 0xCD9E8D57u, pb1 = (uint64_t)b.z * 0xCD9E8D57u;
    a = make_uint4((uint32_t)(pa1 >> 32) ^ a.y ^ k0, (uint32_t)pa1, (uint32_t)(pa0 >> 32) ^ a.w ^ k1, (uint32_t)pa0);
    b = make_uint4((uint32_t)(pb1 >> 32) ^ b.y ^ k0, (uint32_t)pb1, (uint32_t)(pb0 >> 32) ^ b.w ^ k1, (uint32_t)pb0);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// four Box-Muller pairs, level by level: the uniforms, the four logarithms, the four square roots, the sines / cosines
__device__ __forceinline__ void box_muller8(const uint4& a, const uint4& b, float out[8]) {
  const float u[8] = {u01(a.x), u01(a.y), u01(a.z), u01(a.w), u01(b.x), u01(b.y), u01(b.z), u01(b.w)};
  float rad[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rad[j] = -2.0f * kLn2 * __builtin_amdgcn_logf(u[2 * j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) rad[j] = __builtin_amdgcn_sqrtf(rad[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    out[2 * j] = rad[j] * __builtin_amdgcn_cosf(u[2 * j + 1]);
    out[2 * j + 1] = rad[j] * __builtin_amdgcn_sinf(u[2 * j + 1]);
  }
}
template <int ROUNDS = BNN_PHILOX_ROUNDS>
__device__ __forceinline__ void philox_normal8(uint32_t group, uint32_t gsample, uint32_t tensor_id, uint32_t k0, uint32_t k1,
                                               float out[8]) {
#ifdef BNN_PHILOX_SEQ       // A/B build knob (tools/): the two calls one after the other, as rounds 1-3 wrote them
  philox_normal4(group, gsample, tensor_id, k0, k1, out);
  philox_normal4(group + 1u, gsample, tensor_id, k0, k1, out + 4);
  return;
#endif
  uint4 a, b;
  philox_pair<ROUNDS>(group, gsample, tensor_id, k0, k1, a, b);
  box_muller8(a, b, out);
}

// ---------------------------------------------------------------------------- math
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * kLn2; }
// acc + log(x) as ONE explicit fma.  The statistics sums must not depend on which kernel a piece of code is inlined
// into: with -ffp-contract=fast `acc += fast_log(x)` is an fma in one kernel and mul + add in another (the pipelined
// evaluator's results are compared bitwise with the plain sequence's), so every accumulation states its contraction.
__device__ __forceinline__ float add_log(float acc, float x) { return __builtin_fmaf(__builtin_amdgcn_logf(x), kLn2, acc); }

// sigma = log1p(exp(rho)) (networks.py:39), no threshold trick: +inf once exp overflows, 0 once
// it underflows, like the reference.  With u = fl(1 + e) and d = u - 1 (exact),
//   log1p(e) = log(u) + log1p((e - d)/u) = log(u) + (e - d) * (1 + O(d)),
// so adding back the rounding residue (e - d) restores the bits that forming 1 + e drops when
// e is small (rho ~ -5), without a division: 7 VALU ops + the overflow select.
__device__ __forceinline__ float softplus(float rho) {
  const float e = fast_exp(rho);
  const float u = 1.0f + e;
  const float d = u - 1.0f;
  const float s = __builtin_fmaf(__builtin_amdgcn_logf(u), kLn2, e - d);
  return (e > 3.0e38f) ? e : s;   // exp overflowed: +inf (e - d would be inf - inf)
}

// ---------------------------------------------------------------------------- reductions
// v + (v moved across lanes by a DPP control); lanes the control or the row mask leaves unwritten add 0.
// __shfl_xor is a ds_bpermute: an LDS-crossbar round trip of ~120 cycles per step, six dependent steps per sum
// (stamps: ten back-to-back sums were 7 k cycles of the finalize tail); DPP moves ride on the add itself.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(float, moved);
}

// Sum over the 64 lanes of a wave, the same value returned in every lane.  All lanes must be active.
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]: every lane holds its quad's sum
  v = dpp_add<0x141, 0xF>(v);   // row_half_mirror: ... its 8-lane half's
  v = dpp_add<0x140, 0xF>(v);   // row_mirror: ... its 16-lane row's
  v = dpp_add<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3: row1 = r0 + r1, row3 = r2 + r3
  v = dpp_add<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3: row3 = r0 + r1 + r2 + r3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Sum `v` over the block; result valid in thread 0.  `scratch` holds >= blockDim.x/64 Ts.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  T tot = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w) tot += scratch[w];
  return tot;
}

// XCD-aware block -> work mapping (speed only, never correctness).  Blocks are dealt round-robin
// over the 8 XCDs (block b and b+8 share an L2), so XCD x = b % 8 is given the contiguous
// range [x*chunk, (x+1)*chunk) of the (tile-major, sample-minor) work list: the few feature
// tiles an XCD touches keep their (mu, rho) resident in that XCD's 4 MiB L2 while the samples
// stream through.  Returns false for the padding blocks of the last range.
__device__ __forceinline__ bool xcd_work_item(int total, int& item) {
  const int b = blockIdx.x;
  const int chunk = (total + 7) >> 3;
  item = (b & 7) * chunk + (b >> 3);
  return (b >> 3) < chunk && item < total;
}

// 2-D XCD-aware work order of the block-GEMM forms (speed only, never correctness).  A work item is (feature group
// g < G, unit u < U, sub-item in < inner) with unit = (sample, 128-row batch block) and sub-items such as K slices.
// Through its XCD's L2 a block pulls its group's weights (shared by every unit) and its unit's x tiles (shared by every
// group).  Dealing the items out group-major (the 1-D order above) keeps the weights resident but fetches a unit's x
// from memory once per GROUP -- measured on the 1200 x 1200 LR layer: 1.8 GB per 256-sample launch, 10 x the x bytes,
// and the kernel's bound at 6.2 TB/s.  Here the groups are split into `classes` classes of (nearly) equal size, each
// small enough to stay L2-resident; the work list is class-major, then unit, then group within the class, then
// sub-item, and XCD x owns the contiguous range [x * chunk, (x + 1) * chunk) of it (block b runs on XCD b % 8, launch
// after launch: tools/xcc_probe.hip).  An XCD therefore works on one class at a time, the sibling blocks of a unit
// (same x, the class's groups) are dispatched back to back, run side by side and read the unit's x tiles within
// microseconds of each other: memory serves them once per class instead of once per group, and the XCDs hold equal
// item counts whatever the class sizes.  Grid: 8 * ceil(G * U * inner / 8) blocks; false for the padding blocks.
struct Xcd2D {
  int G, U, classes, inner;
};
__device__ __forceinline__ bool xcd2d_work_item(const Xcd2D& m, int& g, int& u, int& in) {
  const long total = (long)m.G * m.U * m.inner;
  const long chunk = (total + 7) >> 3;
  const int b = blockIdx.x;
  long f = (long)(b & 7) * chunk + (b >> 3);
  if ((b >> 3) >= chunk || f >= total) return false;
  const int base = m.G / m.classes, extra = m.G % m.classes;
  const long per_group = (long)m.U * m.inner;
  int h = 0, start = 0, size = base + (extra > 0 ? 1 : 0);
#pragma unroll 1
  while (f >= (long)size * per_group) {              // wave-uniform: at most `classes` rounds on the scalar unit
    f -= (long)size * per_group;
    start += size;
    ++h;
    size = base + (h < extra ? 1 : 0);
  }
  const int row = size * m.inner;
  u = (int)(f / row);
  const int rem = (int)(f - (long)u * row);
  g = start + rem / m.inner;
  in = rem % m.inner;
  return true;
}
// Host side: as few classes as keep a class's groups within the L2 budget.
inline Xcd2D xcd2d_make(int G, int U, int inner, size_t group_bytes, size_t l2_budget) {
  Xcd2D m;
  m.G = G; m.U = U; m.inner = inner < 1 ? 1 : inner;
  size_t per = group_bytes ? l2_budget / group_bytes : (size_t)G;
  if (per < 1) per = 1;
  int c = (int)((G + per - 1) / per);
  m.classes = c < 1 ? 1 : (c > G ? G : c);
  return m;
}

// The split-bf16 pair of BNN_MATH_BF16X3 (include/bnn_hip.h): hi = bf16(v), lo = bf16(v - hi), both RNE.
// (split_lo takes the already rounded hi: vector elements do not bind to references)
__device__ __forceinline__ __bf16 split_lo(float v, __bf16 hi) { return (__bf16)(v - (float)hi); }

// fp32 -> bf16 (RNE) cast of the input batch, once per ELBO evaluation, so that every layer of
// the throughput path reads 2-byte activations (the x tile is then LDS-DMA'd as is).  Thread `tid` of `nt`.
// dsq (optional): the squares; dlo (optional): the low plane of the split-bf16 pair.
// The x^2 operand of the LR variance product in bf16 math (include/bnn_hip.h, BNN_MATH_BF16): the square of the ROUNDED
// activation, rounded -- bf16(bf16(x)^2) -- whoever forms it: a producer's epilogue / the input cast (the `x_sq` stream K3b
// reads) or a consumer squaring the bf16 fragment it loaded (K3a, K3s, K3r).  One value per element, whatever the launch plans.
__device__ __forceinline__ __bf16 sq_bf16(float v) {
  const float r = (float)(__bf16)v;
  return (__bf16)(r * r);
}

__device__ __forceinline__ void cast_bf16_span(const float* __restrict__ src, __bf16* __restrict__ dst,
                                               __bf16* __restrict__ dsq, long n, int vec_ok, long tid, long nt,
                                               __bf16* __restrict__ dlo = nullptr) {
  if (vec_ok) {
    const long n8 = n >> 3;
    for (long i = tid; i < n8; i += nt) {
      const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
      const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      bf16x8 v, q, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = (__bf16)f[j];
        l[j] = split_lo(f[j], v[j]);
        q[j] = dlo ? (__bf16)(f[j] * f[j]) : sq_bf16(f[j]);
      }
      reinterpret_cast<bf16x8*>(dst)[i] = v;
      if (dsq) reinterpret_cast<bf16x8*>(dsq)[i] = q;
      if (dlo) reinterpret_cast<bf16x8*>(dlo)[i] = l;
    }
    for (long i = (n8 << 3) + tid; i < n; i += nt) {
      const __bf16 h = (__bf16)src[i], l = split_lo(src[i], h);
      dst[i] = h;
      if (dsq) dsq[i] = dlo ? (__bf16)(src[i] * src[i]) : sq_bf16(src[i]);
      if (dlo) dlo[i] = l;
    }
  } else {
    for (long i = tid; i < n; i += nt) {
      const __bf16 h = (__bf16)src[i], l = split_lo(src[i], h);
      dst[i] = h;
      if (dsq) dsq[i] = dlo ? (__bf16)(src[i] * src[i]) : sq_bf16(src[i]);
      if (dlo) dlo[i] = l;
    }
  }
}

}  // namespace bnn
