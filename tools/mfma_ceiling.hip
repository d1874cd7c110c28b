// What the chip sustains on bare bf16 MFMA streams (operands in registers, random data, every CU busy): the ceiling
// K1g's schedule can be compared with.  Variants: MFMA shape (16x16x32 / 32x32x16), waves per SIMD (1, 2), s_nop padding
// between MFMAs, an s_barrier every 32 MFMAs.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o tools/mfma_ceiling.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// SHAPE 0: 16x16x32 (32 accumulators of 4 regs = the 128 x 64 wave tile of K1g); 1: 32x32x16 (8 accumulators of 16 regs)
template <int SHAPE, int PAD, bool BAR>
__global__ __launch_bounds__(512) void k(const float4* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(bf16x8, in[(tid * 12 + i) & 0xFFFFF]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(bf16x8, in[(tid * 12 + 8 + i) & 0xFFFFF]);
  float acc_out = 0.f;
  if (SHAPE == 0) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            if (PAD == 1) asm volatile("s_nop 0");
            if (PAD == 2) asm volatile("s_nop 3");
            if (BAR && ((i * 4 + j) & 31) == 31) __builtin_amdgcn_s_barrier();
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc_out += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {                 // 4 k-steps of 16 = K 64
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j + 2 * (h & 1)], a[i + 4 * (h >> 1)], acc[i][j], 0, 0, 0);
            if (PAD == 1) asm volatile("s_nop 0");
            if (PAD == 2) asm volatile("s_nop 3");
            if (BAR && ((h * 8 + i * 2 + j) & 15) == 15) __builtin_amdgcn_s_barrier();
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc_out += acc[i][j][0] + acc[i][j][15];
  }
  out[tid] = acc_out;
}

// Schedule-shaped streams of 16x16x32 (64 per iteration, as one K-tile of K1g):
//   MODE 0: s_setprio 1 / 0 around each 32-MFMA segment, no barriers
//   MODE 1: K1g's ping-pong: [s_barrier; 32 MFMAs; s_barrier; nothing] with waves 4-7 one barrier behind (one issuer per SIMD at a time)
//   MODE 2: 24 ds_read_b128 spread between the MFMAs (one behind 3 of every 8), lgkmcnt(0) once per iteration, no barriers
//   MODE 3: MODE 2 + one s_barrier per iteration (the plain software-pipelined form: both waves of a SIMD issue MFMAs)
template <int MODE>
__global__ __launch_bounds__(512) void k2(const float4* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ float4 lds[2048];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = in[i];
  __syncthreads();
  bf16x8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(bf16x8, in[(tid * 12 + i) & 0xFFFFF]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(bf16x8, in[(tid * 12 + 8 + i) & 0xFFFFF]);
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t la = (uint32_t)(size_t)(&lds[0]) + (threadIdx.x & 63) * 16;
  const bool late = (threadIdx.x >> 8) != 0;
  if (MODE == 1 && late) __builtin_amdgcn_s_barrier();
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (MODE == 1) __builtin_amdgcn_s_barrier();
      if (MODE <= 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          if (MODE >= 2 && ((i * 4 + j) & 7) < 3) {
            f32x4 t;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(t) : "v"(la), "n"(((i * 4 + j) & 31) * 1024));
            asm volatile("" :: "v"(t));
          }
        }
      if (MODE <= 1) __builtin_amdgcn_s_setprio(0);
      if (MODE == 1) __builtin_amdgcn_s_barrier();
    }
    if (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 3) __builtin_amdgcn_s_barrier();
  }
  if (MODE == 1 && !late) __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sink += acc[i][j][0] + acc[i][j][3];
  out[tid] = sink;
}

// K1g's own MFMA order inside the ping-pong skeleton: A = one of 8 weight fragments, B = one of 8 x fragments, four quadrants of
// 16 MFMAs per K-tile, every accumulator touched again 8-16 MFMAs later (MODE 0), or the same with the two k halves of a quadrant
// pair separated by the other pair (distance 32, MODE 1)
template <int MODE>
__global__ __launch_bounds__(512) void k3(const float4* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 wf[2][2][2], xf[4][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) wf[i >> 2][(i >> 1) & 1][i & 1] = __builtin_bit_cast(bf16x8, in[(tid * 16 + i) & 0x7FFFF]);
#pragma unroll
  for (int i = 0; i < 8; ++i) xf[i >> 1][i & 1] = __builtin_bit_cast(bf16x8, in[0x80000 + ((tid * 16 + i) & 0x7FFFF)]);
  f32x4 acc[2][4][2][2];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i >> 4][(i >> 2) & 3][(i >> 1) & 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool late = (threadIdx.x >> 8) != 0;
  if (late) __builtin_amdgcn_s_barrier();
  auto quad = [&](int mh, int nh, int kh) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mh][mi][nh][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nh][ni][kh], xf[mi][kh], acc[mh][mi][nh][ni], 0, 0, 0);
  };
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if (MODE == 0) {
        quad(h, h, 0); quad(h, 1 - h, 0);
        __builtin_amdgcn_sched_barrier(0);
        quad(h, h, 1); quad(h, 1 - h, 1);
      } else {
        quad(h, h, 0); quad(h, h, 1);
        __builtin_amdgcn_sched_barrier(0);
        quad(h, 1 - h, 0); quad(h, 1 - h, 1);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();
  float sink = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) sink += acc[i >> 4][(i >> 2) & 3][(i >> 1) & 1][i & 1][0] + acc[i >> 4][(i >> 2) & 3][(i >> 1) & 1][i & 1][3];
  out[tid] = sink;
}

template <int MODE>
static void run3(const char* name, const float4* in, float* out, int iters, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k3<MODE>), dim3(256), dim3(512), 0, 0, in, out, iters);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k3<MODE>), dim3(256), dim3(512), 0, 0, in, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)reps * iters * 64.0 * 16384.0 * (256.0 * 512 / 64.0);
  printf("%-52s  512 threads/CU, %5d iterations: %7.1f TFLOP/s (%.3f of 2.5 PF), %.3f ms per launch\n", name, iters,
         flops / (ms * 1e-3) * 1e-12, flops / (ms * 1e-3) * 1e-12 / 2500.0, ms / reps);
}

// The same launches with the memory system kept awake: a device memset of `bytes` before every launch, the kernel alone timed
template <int MODE>
static void run3_mem(const char* name, const float4* in, float* out, int iters, int reps, void* scratch, size_t bytes) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double total = 0;
  for (int r = 0; r < reps + 2; ++r) {
    (void)hipMemsetAsync(scratch, r, bytes, 0);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k3<MODE>), dim3(256), dim3(512), 0, 0, in, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) total += ms;
  }
  const double flops = (double)reps * iters * 64.0 * 16384.0 * (256.0 * 512 / 64.0);
  printf("%-52s  512 threads/CU, %5d iterations: %7.1f TFLOP/s (%.3f of 2.5 PF), %.3f ms per launch\n", name, iters,
         flops / (total * 1e-3) * 1e-12, flops / (total * 1e-3) * 1e-12 / 2500.0, total / reps);
}

template <int MODE>
static void run2(const char* name, int threads, const float4* in, float* out, int iters = 4000, int reps = 6, int dyn_lds = 0) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  if (dyn_lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k2<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn_lds);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k2<MODE>), dim3(256), dim3(threads), dyn_lds, 0, in, out, iters);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k2<MODE>), dim3(256), dim3(threads), dyn_lds, 0, in, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)reps * iters * 64.0 * 16384.0 * (256.0 * threads / 64.0);
  printf("%-52s %4d threads/CU, %5d iterations, %3d KiB LDS: %7.1f TFLOP/s (%.3f of 2.5 PF), %.3f ms per launch\n", name, threads, iters,
         (dyn_lds + 32768) >> 10, flops / (ms * 1e-3) * 1e-12, flops / (ms * 1e-3) * 1e-12 / 2500.0, ms / reps);
}

template <int SHAPE, int PAD, bool BAR>
static void run(const char* name, int threads, const float4* in, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<SHAPE, PAD, BAR>), dim3(256), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e0, 0);
  const int reps = 6;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<SHAPE, PAD, BAR>), dim3(256), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per wave and iteration: 64 MFMAs of 16x16x32 (or 32 of 32x32x16) = 64 * 16384 flop
  const double flops = (double)reps * iters * 64.0 * 16384.0 * (256.0 * threads / 64.0);
  printf("%-52s %4d threads/CU: %7.1f TFLOP/s (%.3f of 2.5 PF), %.2f ms per launch\n", name, threads, flops / (ms * 1e-3) * 1e-12,
         flops / (ms * 1e-3) * 1e-12 / 2500.0, ms / reps);
}

int main(int argc, char** argv) {
  const int data_mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: all operands in (-1, 1); 1: first half of the buffer (weights) in (-0.2, 0.2), second half (x) in (0, 1)
  std::vector<float> h(1 << 22);
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  auto bf = [&](float f) {                             // round-to-nearest-even bf16 bits of f
    uint32_t b;
    __builtin_memcpy(&b, &f, 4);
    return (uint32_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
  };
  size_t idx = 0;
  for (auto& v : h) {                                  // two finite bf16 values per word: realistic operand toggling,
    auto draw = [&]() {                                // accumulators stay finite (an Inf / NaN stream would draw less power)
      const float f = u(rng);
      if (!data_mode) return f;
      return idx < h.size() / 2 ? 0.2f * f : 0.5f + 0.5f * f;
    };
    const uint32_t bits = (bf(draw()) << 16) | bf(draw());
    __builtin_memcpy(&v, &bits, 4);
    ++idx;
  }
  printf("operand data mode %d\n", data_mode);
  float4* in; float* out;
  hipMalloc(&in, h.size() * 4); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<0, 0, false>("16x16x32 back to back", threads, in, out);
    run<0, 1, false>("16x16x32 + s_nop 0 after each", threads, in, out);
    run<0, 2, false>("16x16x32 + s_nop 3 after each", threads, in, out);
    run<0, 0, true>("16x16x32, s_barrier every 32", threads, in, out);
    run<1, 0, false>("32x32x16 back to back", threads, in, out);
    run<1, 1, false>("32x32x16 + s_nop 0 after each", threads, in, out);
    run<1, 0, true>("32x32x16, s_barrier every 16", threads, in, out);
  }
  run2<0>("setprio around 32-MFMA segments", 512, in, out);
  run2<1>("ping-pong: one issuer per SIMD, 2 barriers/segment", 512, in, out);
  run2<1>("ping-pong, 128 KiB of LDS per block", 512, in, out, 4000, 6, 96 * 1024);
  run2<1>("ping-pong, launches of one 64-K-tile block", 512, in, out, 64, 20);
  run2<1>("ping-pong, launches of 256 K-tiles", 512, in, out, 256, 20);
  run2<1>("ping-pong, 64 K-tiles, 128 KiB LDS", 512, in, out, 64, 20, 96 * 1024);
  run3<0>("ping-pong, K1g's MFMA order (distance 16)", in, out, 4000, 6);
  run3<0>("ping-pong, K1g's MFMA order (distance 16)", in, out, 64, 20);
  {
    void* scratch; (void)hipMalloc(&scratch, 256u << 20);
    run3_mem<0>("  same, event pair per launch, no memset", in, out, 64, 20, scratch, 4);
    run3_mem<0>("  same, 64 MiB memset before each launch", in, out, 64, 20, scratch, 64u << 20);
    run3_mem<0>("  same, 256 MiB memset before each launch", in, out, 64, 20, scratch, 256u << 20);
    run3_mem<0>("  256 iterations, 256 MiB memset before each", in, out, 256, 20, scratch, 256u << 20);
    (void)hipFree(scratch);
  }
  run3<1>("ping-pong, K1g's quadrants, k halves adjacent (8)", in, out, 4000, 6);
  run3<1>("ping-pong, K1g's quadrants, k halves adjacent (8)", in, out, 64, 20);
  run2<2>("24 ds_read_b128 between the MFMAs", 512, in, out);
  run2<2>("24 ds_read_b128 between the MFMAs", 256, in, out);
  run2<3>("24 ds_read_b128 between + 1 barrier per 64", 512, in, out);
  run2<3>("24 ds_read_b128 between + 1 barrier per 64", 256, in, out);
  return 0;
}
