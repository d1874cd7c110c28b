// Diagnostic micro-benchmarks (not part of the product): cycles per call of the generator
// building blocks on one SIMD, at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include "../bayesian-neural-network_amd/csrc/bnn_device.h"
using namespace bnn;

template <int WHAT>
__global__ void k(unsigned long long* out, float* sink, int iters) {
  unsigned x = threadIdx.x * 2654435761u + 12345u;
  float acc = 0.f;
  float f = (float)(threadIdx.x + 1) * 1e-3f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (WHAT == 0) {  // philox4x32_10 x1
      uint4 r = philox4x32_10(make_uint4(x, i, 3, 0), 1, 2); x ^= r.x ^ r.y ^ r.z ^ r.w;
    } else if (WHAT == 1) {  // philox_normal4 (philox + 2 box-muller)
      float e[4]; philox_normal4(x, i, 3, 1, 2, e); acc += e[0] + e[1] + e[2] + e[3]; x += 7;
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
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)x;
}

int main() {
  const char* names[] = {"philox4x32_10 (4 u32)", "philox_normal4 (4 normals)", "softplus x8", "16 mad_u64_u32 (+32 xor)", "16 fma (+16 add)", "box_muller x2 (4 normals)"};
  unsigned long long* d; float* s; hipMalloc(&d, 1 << 20); hipMalloc(&s, 64 << 20);
  const int iters = 2000;
  for (int what = 0; what < 6; ++what) {
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
