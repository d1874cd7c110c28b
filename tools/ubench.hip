// Diagnostic micro-benchmarks (not part of the product): cycles per call of the generator
// building blocks on one SIMD, at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../bayesian-neural-network_amd/csrc/bnn_device.h"
using namespace bnn;

// Philox4x32 with R rounds; MULHI: v_mul_hi_u32 + v_mul_lo_u32 instead of the 64-bit multiply-add
template <int R, bool MULHI>
__device__ __forceinline__ uint4 philox_r(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < R; ++i) {
    uint32_t hi0, lo0, hi1, lo1;
    if (MULHI) {
      hi0 = __umulhi(c.x, 0xD2511F53u); lo0 = c.x * 0xD2511F53u;
      hi1 = __umulhi(c.z, 0xCD9E8D57u); lo1 = c.z * 0xCD9E8D57u;
    } else {
      const uint64_t p0 = (uint64_t)c.x * 0xD2511F53u, p1 = (uint64_t)c.z * 0xCD9E8D57u;
      hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
    }
    c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
template <int R, bool MULHI>
__device__ __forceinline__ void normal4_r(uint32_t g, uint32_t s, float out[4]) {
  const uint4 r = philox_r<R, MULHI>(make_uint4(g, s, 3u, 0u), 1u, 2u);
  box_muller(r.x, r.y, out[0], out[1]);
  box_muller(r.z, r.w, out[2], out[3]);
}
// 8 normals from ONE Philox call: 16-bit uniforms (radius from the high halves, angle from the low halves)
__device__ __forceinline__ void bm16(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = __builtin_fmaf((float)a, 1.52587890625e-05f, 7.62939453125e-06f);   // (a + 0.5) / 65536
  const float u2 = (float)b * 1.52587890625e-05f;
  const float rad = __builtin_amdgcn_sqrtf(-2.0f * kLn2 * __builtin_amdgcn_logf(u1));
  n0 = rad * __builtin_amdgcn_cosf(u2);
  n1 = rad * __builtin_amdgcn_sinf(u2);
}
template <int R>
__device__ __forceinline__ void normal8_16bit(uint32_t g, uint32_t s, float out[8]) {
  const uint4 r = philox_r<R, false>(make_uint4(g, s, 3u, 0u), 1u, 2u);
  bm16(r.x >> 16, r.x & 0xFFFFu, out[0], out[1]);
  bm16(r.y >> 16, r.y & 0xFFFFu, out[2], out[3]);
  bm16(r.z >> 16, r.z & 0xFFFFu, out[4], out[5]);
  bm16(r.w >> 16, r.w & 0xFFFFu, out[6], out[7]);
}
// epsilon-map v3 candidate (round-3 verdict, item 2): 40 bits per Box-Muller pair -- a 24-bit radius uniform (tail sqrt(2 ln 2^25) =
// 5.89 sigma, what a 24-bit fp32 uniform gives) and a 16-bit angle (v_sin / v_cos take revolutions) -- so ONE Philox call yields
// three pairs = 6 normals instead of 4.  Word w of (x, y, z) carries a pair's radius in its upper 24 bits and the high byte of its
// angle in its low byte; the angle's low byte is byte w of the fourth word.
__device__ __forceinline__ void bm_v3(uint32_t word, uint32_t lowbyte, float& n0, float& n1) {
  const float u1 = __builtin_fmaf((float)(word >> 8), 5.9604644775390625e-08f, 2.98023223876953125e-08f);   // (r24 + 0.5) / 2^24
  const float u2 = (float)(((word & 0xFFu) << 8) | lowbyte) * 1.52587890625e-05f;                              // a16 / 2^16 revolutions
  const float rad = __builtin_amdgcn_sqrtf(-2.0f * kLn2 * __builtin_amdgcn_logf(u1));
  n0 = rad * __builtin_amdgcn_cosf(u2);
  n1 = rad * __builtin_amdgcn_sinf(u2);
}
template <int R>
__device__ __forceinline__ void normal6_v3(uint32_t g, uint32_t s, float out[6]) {
  const uint4 r = philox_r<R, false>(make_uint4(g, s, 3u, 0u), 1u, 2u);
  bm_v3(r.x, r.w & 0xFFu, out[0], out[1]);
  bm_v3(r.y, (r.w >> 8) & 0xFFu, out[2], out[3]);
  bm_v3(r.z, (r.w >> 16) & 0xFFu, out[4], out[5]);
}
// 24 weights (three k-steps of a lane) around a generator: GEN 0 = map v2 (6 Philox-7 calls, 4 normals each), 1 = map v3 candidate
// (4 calls, 6 normals each); the same w / statistics / pack work either way
template <int GEN>
__device__ __forceinline__ float k1b_body24(uint32_t x, int i, float f) {
  float e[24];
  if (GEN == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) normal4_r<7, false>(x + c, i, e + 4 * c);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) normal6_v3<7>(x + c, i, e + 6 * c);
  }
  float e2 = 0.f, a = 0.f, acc = 0.f;
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    bf16x8 wa;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w = __builtin_fmaf(f + j, e[h * 8 + j], f * 0.5f);
      e2 = __builtin_fmaf(e[h * 8 + j], e[h * 8 + j], e2);
      a = __builtin_fmaf(w, w, a);
      wa[j] = (__bf16)w;
    }
    const float4 pk = __builtin_bit_cast(float4, wa);
    acc += pk.x + pk.y + pk.z + pk.w;
  }
  return e2 + a + acc;
}
// the per-8-weights body of K1b around a generator G: w = mu + sigma * eps, sum eps^2, sum w^2, bf16 pack
template <int GEN>
__device__ __forceinline__ float k1b_body(uint32_t x, int i, float f) {
  float e[8];
  if (GEN == 0) { normal4_r<10, false>(x, i, e); normal4_r<10, false>(x + 1, i, e + 4); }
  if (GEN == 1) { normal4_r<7, false>(x, i, e); normal4_r<7, false>(x + 1, i, e + 4); }
  if (GEN == 2) { normal8_16bit<10>(x, i, e); }
  if (GEN == 3) { normal8_16bit<7>(x, i, e); }
  float e2 = 0.f, a = 0.f, acc = 0.f;
  bf16x8 wa;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float w = __builtin_fmaf(f + j, e[j], f * 0.5f);
    e2 = __builtin_fmaf(e[j], e[j], e2);
    a = __builtin_fmaf(w, w, a);
    wa[j] = (__bf16)w;
  }
  const float4 pk = __builtin_bit_cast(float4, wa);
  return e2 + a + pk.x + pk.y + pk.z + pk.w;
}

template <int WHAT>
__global__ void k(unsigned long long* out, float* sink, int iters) {
  unsigned x = threadIdx.x * 2654435761u + 12345u;
  float acc = 0.f;
  float f = (float)(threadIdx.x + 1) * 1e-3f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (WHAT == 0) {  // philox4x32_10 x1
      uint4 r = philox4x32<10>(make_uint4(x, i, 3, 0), 1, 2); x ^= r.x ^ r.y ^ r.z ^ r.w;
    } else if (WHAT == 1) {  // philox_normal4 (philox + 2 box-muller)
      float e[4]; normal4_r<10, false>(x, i, e); acc += e[0] + e[1] + e[2] + e[3]; x += 7;
    } else if (WHAT == 2) {  // softplus x8
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += softplus(f + j * 0.1f + acc * 1e-9f);
    } else if (WHAT == 3) {  // 16 independent v_mad_u64_u32
      unsigned long long a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = (unsigned long long)(x + j) * 0xD2511F53u;
#pragma unroll
      for (int j = 0; j < 16; ++j) x ^= (unsigned)(a[j] >> 32) ^ (unsigned)a[j];
    } else if (WHAT == 4) {  // 16 independent fma
      float a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_fmaf(f, (float)j, acc);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += a[j];
    } else if (WHAT == 5) {  // box_muller x2
      float a, b, c, d; box_muller(x, x * 3u, a, b); box_muller(x * 5u, x * 7u, c, d); acc += a + b + c + d; x += 11;
    } else if (WHAT == 6) {  // philox-10 normal4 with mul_hi + mul_lo
      float e[4]; normal4_r<10, true>(x, i, e); acc += e[0] + e[1] + e[2] + e[3]; x += 7;
    } else if (WHAT == 7) {  // philox-7 normal4
      float e[4]; normal4_r<7, false>(x, i, e); acc += e[0] + e[1] + e[2] + e[3]; x += 7;
    } else if (WHAT == 8) {  // 8 normals from one philox-10 call, 16-bit uniforms
      float e[8]; normal8_16bit<10>(x, i, e); acc += e[0] + e[1] + e[2] + e[3] + e[4] + e[5] + e[6] + e[7]; x += 7;
    } else if (WHAT == 9) {  // K1b body per 8 weights: philox-10 x2
      acc += k1b_body<0>(x, i, f); x += 2;
    } else if (WHAT == 10) { // K1b body: philox-7 x2
      acc += k1b_body<1>(x, i, f); x += 2;
    } else if (WHAT == 11) { // K1b body: one philox-10, 16-bit uniforms
      acc += k1b_body<2>(x, i, f); x += 2;
    } else if (WHAT == 12) { // K1b body: one philox-7, 16-bit uniforms
      acc += k1b_body<3>(x, i, f); x += 2;
    } else if (WHAT == 13) { // 24 weights, map v2: 6 philox-7 calls
      acc += k1b_body24<0>(x, i, f); x += 6;
    } else if (WHAT == 14) { // 24 weights, map v3 candidate: 4 philox-7 calls, 24-bit radius / 16-bit angle
      acc += k1b_body24<1>(x, i, f); x += 4;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)x;
}

int main() {
  const char* names[] = {"philox4x32_10 (4 u32)", "philox_normal4 (4 normals)", "softplus x8", "16 mad_u64_u32 (+32 xor)", "16 fma (+16 add)", "box_muller x2 (4 normals)",
                         "normal4, philox-10 mul_hi+mul_lo", "normal4, philox-7", "normal8, 1 philox-10, 16-bit u", "K1b body/8w: 2 philox-10",
                         "K1b body/8w: 2 philox-7", "K1b body/8w: 1 philox-10 16b", "K1b body/8w: 1 philox-7 16b",
                         "body/24w: 6 philox-7 (map v2)", "body/24w: 4 philox-7 (v3: 24b/16b)"};
  unsigned long long* d; float* s; hipMalloc(&d, 1 << 20); hipMalloc(&s, 64 << 20);
  const int iters = 2000;
  for (int what = (getenv("UBENCH_FROM") ? atoi(getenv("UBENCH_FROM")) : 0); what < 15; ++what) {
    for (int wps = 1; wps <= 4; ++wps) {   // waves per SIMD: block = wps*4 waves, one block per CU
      dim3 grid(256), block(wps * 256);
      for (int rep = 0; rep < 3; ++rep) {
        switch (what) {
          case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, s, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, s, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, s, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, s, iters); break;
          case 4: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, s, iters); break;
          case 5: hipLaunchKernelGGL(k<5>, grid, block, 0, 0, d, s, iters); break;
          case 6: hipLaunchKernelGGL(k<6>, grid, block, 0, 0, d, s, iters); break;
          case 7: hipLaunchKernelGGL(k<7>, grid, block, 0, 0, d, s, iters); break;
          case 8: hipLaunchKernelGGL(k<8>, grid, block, 0, 0, d, s, iters); break;
          case 9: hipLaunchKernelGGL(k<9>, grid, block, 0, 0, d, s, iters); break;
          case 10: hipLaunchKernelGGL(k<10>, grid, block, 0, 0, d, s, iters); break;
          case 11: hipLaunchKernelGGL(k<11>, grid, block, 0, 0, d, s, iters); break;
          case 12: hipLaunchKernelGGL(k<12>, grid, block, 0, 0, d, s, iters); break;
          case 13: hipLaunchKernelGGL(k<13>, grid, block, 0, 0, d, s, iters); break;
          case 14: hipLaunchKernelGGL(k<14>, grid, block, 0, 0, d, s, iters); break;
        }
      }
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256 * wps * 4);
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      double med = (double)h[h.size() / 2] / iters;
      printf("%-28s waves/SIMD %d: %8.1f cyc/iter per wave -> %7.1f cyc/iter per SIMD-wave-slot\n", names[what], wps, med, med / wps);
    }
  }
  return 0;
}
