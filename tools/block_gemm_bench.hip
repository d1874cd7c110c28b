// Development harness of K1g (csrc/bbb_block_gemm.h): random bf16 operands, spot check against a double-precision
// host sum, HIP-event timing.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/block_gemm_bench.hip -o tools/block_gemm_bench.out
// Run:   tools/block_gemm_bench.out S M N K [iters] [ybf16] [xshared]
#include "../bayesian-neural-network_amd/csrc/bbb_block_gemm.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>

#define CK(e)                                                                         \
  do {                                                                                \
    hipError_t _e = (e);                                                              \
    if (_e != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e));       \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 4, M = argc > 2 ? atoi(argv[2]) : 4096, N = argc > 3 ? atoi(argv[3]) : 4096,
            K = argc > 4 ? atoi(argv[4]) : 4096, iters = argc > 5 ? atoi(argv[5]) : 20, ybf = argc > 6 ? atoi(argv[6]) : 1,
            xshared = argc > 7 ? atoi(argv[7]) : 0;
  const int XS = xshared ? 1 : S;
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> ux(0.f, 1.f), uw(-0.2f, 0.2f);
  std::vector<uint16_t> hx((size_t)XS * M * K), hw((size_t)S * N * K);
  std::vector<float> hb((size_t)S * N);
  for (auto& v : hx) v = f2bf(ux(rng));
  for (auto& v : hw) v = f2bf(uw(rng));
  for (auto& v : hb) v = uw(rng);
  void *dx, *dw, *db, *dy;
  const size_t ybytes = (size_t)S * M * N * (ybf ? 2 : 4);
  CK(hipMalloc(&dx, hx.size() * 2));
  CK(hipMalloc(&dw, hw.size() * 2));
  CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dy, ybytes));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0xFF, ybytes));
  bnn::BlockGemmK k{};
  k.x = (const __bf16*)dx; k.x_sstride = xshared ? 0 : (long)M * K; k.xg = 1;
  k.w = (const __bf16*)dw; k.bias = (const float*)db; k.y = dy; k.y_bf16 = ybf; k.relu = 1;
  k.S = S; k.M = M; k.N = N; k.K = K;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  CK(bnn::launch_block_gemm(k, st));
  CK(hipStreamSynchronize(st));
  // spot check
  std::vector<uint8_t> hy(ybytes);
  CK(hipMemcpy(hy.data(), dy, ybytes, hipMemcpyDeviceToHost));
  std::mt19937 pick(99);
  double max_err = 0, max_ref = 0;
  int bad = 0;
  const int checks = 4000;
  for (int c = 0; c < checks; ++c) {
    int s = pick() % S, m, n;
    if (c % 4 == 0) { m = M - 1 - (pick() % std::min(M, 3)); n = N - 1 - (pick() % std::min(N, 5)); }   // edges
    else { m = pick() % M; n = pick() % N; }
    double ref = hb[(size_t)s * N + n];
    const uint16_t* xr = &hx[((size_t)(xshared ? 0 : s) * M + m) * K];
    const uint16_t* wr_ = &hw[((size_t)s * N + n) * K];
    for (int kk = 0; kk < K; ++kk) ref += (double)bf2f(xr[kk]) * (double)bf2f(wr_[kk]);
    if (ref < 0) ref = 0;
    const size_t o = ((size_t)s * M + m) * N + n;
    const double got = ybf ? bf2f(((uint16_t*)hy.data())[o]) : ((float*)hy.data())[o];
    const double err = fabs(got - ref), tol = (ybf ? 8e-3 : 2e-5) * std::max(1.0, fabs(ref)) + 1e-4 * sqrt((double)K) * 0.01;
    if (!(err <= tol)) {
      if (bad < 10) printf("MISMATCH s=%d m=%d n=%d got=%g ref=%g\n", s, m, n, got, ref);
      ++bad;
    }
    max_err = std::max(max_err, err);
    max_ref = std::max(max_ref, fabs(ref));
  }
  printf("check: %d/%d bad, max abs err %.3g (max |ref| %.3g)\n", bad, checks, max_err, max_ref);
  // timing
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(bnn::launch_block_gemm(k, st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) CK(bnn::launch_block_gemm(k, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, flops = 2.0 * S * M * (double)N * K;
  printf("S=%d M=%d N=%d K=%d ybf16=%d xshared=%d: %.1f us per launch, %.1f TFLOP/s (%.3f of 2.5 PF)\n", S, M, N, K, ybf, xshared, us,
         flops / us * 1e-6, flops / us * 1e-6 / 2500.0);
  return bad ? 1 : 0;
}
