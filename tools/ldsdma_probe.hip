// Diagnostic (not part of the product): what one CU ingests by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) from
// an L2-resident source, by waves per CU and pieces kept in flight per wave.  Every kernel of this repository that stages
// operands through LDS (K1b2, K3b, K1g) sits near 40 GB/s per CU: is that the path's ceiling or a depth of prefetch?
//   usage: ldsdma_probe.out [source MiB per XCD-shared region = 2]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(1024) void probe(const char* __restrict__ src, size_t region, int iters, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  // every block walks the same region (L2-resident after the first pass), offset by block and wave so that requests differ
  size_t off = ((size_t)blockIdx.x * 131 + (size_t)wave * 17) * 1024 % region;
  const char* base = src + (size_t)lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // prime DEPTH pieces, then one new piece per retired piece
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                     (__attribute__((address_space(3))) void*)(lds + ((wave * DEPTH + d) << 10)), 16, 0, 0);
    off = (off + (size_t)nw * 1024) % region;
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                       (__attribute__((address_space(3))) void*)(lds + ((wave * DEPTH + d) << 10)), 16, 0, 0);
      off = (off + (size_t)nw * 1024) % region;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = reinterpret_cast<float*>(lds)[threadIdx.x];
}

template <int DEPTH>
static void run(const char* src, size_t region, int waves, unsigned long long* d_cyc, float* d_sink) {
  const int iters = 400;
  const size_t lds = (size_t)waves * DEPTH * 1024;
  if (lds > 160 * 1024) return;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<DEPTH>, dim3(256), dim3(waves * 64), lds, 0, src, region, iters, d_cyc, d_sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<DEPTH>, dim3(256), dim3(waves * 64), lds, 0, src, region, iters, d_cyc, d_sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * waves * (double)(iters + 1) * DEPTH * 1024.0;
  printf("waves/CU %2d  pieces in flight per wave %2d (%3d KiB per CU): %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n", waves, DEPTH,
         waves * DEPTH, ms * 1e3, bytes / 256.0 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const size_t region = (size_t)(argc > 1 ? atoi(argv[1]) : 2) << 20;
  char* src; unsigned long long* d_cyc; float* d_sink;
  hipMalloc(&src, region + (1 << 20)); hipMemset(src, 1, region + (1 << 20));
  hipMalloc(&d_cyc, 256 * 8); hipMalloc(&d_sink, 256 * 1024 * 4);
  printf("source region %zu MiB (shared by all blocks: L2-resident)\n", region >> 20);
  for (int waves : {4, 8, 16}) {
    run<1>(src, region, waves, d_cyc, d_sink);
    run<2>(src, region, waves, d_cyc, d_sink);
    run<4>(src, region, waves, d_cyc, d_sink);
    run<8>(src, region, waves, d_cyc, d_sink);
    if (waves <= 8) run<16>(src, region, waves, d_cyc, d_sink);
  }
  return 0;
}
