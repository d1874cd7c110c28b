// K1s  bnn_bbb_sample_weights — the sampling half of BayesianLinear.forward for every hidden layer of a network
// in ONE launch (reference networks.py:73-86: w = mu + softplus(rho) * eps, b likewise, and the sums that
// log_prior / log_variational_posterior are made of), writing the sampled weights as bf16 [S, out, in] and the
// sampled biases as fp32 [S, out].  The matmul half then runs as a plain bf16 MFMA GEMM over the sampled
// weights (bnn_bbb_linear_fwd with w_sampled).
//
// Why split at all: a one-sample evaluation is a chain of dependent layer launches, and with sampling fused into
// each of them (K1a) the chain carries the whole Philox/Box-Muller/softplus cost of the network, one layer after
// the other, at a tile decomposition chosen for the matmul.  Sampling depends on no activation: taken out of the
// chain it is a streaming pass (8 B read + 2 B written per weight, ~65 VALU issue slots per weight) that fills
// every SIMD of the chip evenly, and the chain behind it is three short matmul launches.  Costs 2 B/weight of
// extra HBM write and read; pays for few MC samples (the per-sample weights of many samples would not fit the
// caches, K1b keeps them in registers instead).
//
// Layout: 1-D grid, blocks ordered [layer][group of 4 samples][chunk] (the parameters are read once per group); a block of
// 256 threads owns 512 consecutive octets
// (8 consecutive k of one output feature: one 16-byte bf16 store, two Philox groups) of the flattened [out, in]
// matrix, two per thread, all eight 16-byte parameter loads of a thread issued before any arithmetic.  Needs
// in_features % 8 == 0 and 16-byte aligned bases.  Statistics: one float4 {sum eps^2, sum w^2 | sum log p_mix,
// sum log sigma (sample 0 only), 0} per block, in the layer's K1 workspace format (entry count in word 0), so
// bnn_elbo_finalize / bnn_bbb_final_fwd read them exactly as they read K1a's tile partials.
#include "bbb_sample_body.h"

namespace bnn {

__global__ __launch_bounds__(kSampleThreads) void bbb_sample_kernel(const SampleK p) {
  __shared__ float red[kSampleRedFloats];
  sample_block(p, (int)blockIdx.x, red);
}

}  // namespace bnn

using namespace bnn;

extern "C" size_t bnn_bbb_sample_workspace_bytes(int32_t n_samples, int32_t in_features, int32_t out_features) {
  if (n_samples <= 0 || in_features <= 0 || out_features <= 0 || (in_features & 7)) return 0;
  // at least what bnn_bbb_linear_fwd asks for the same layer, so one workspace serves either form
  const size_t k1 = bnn_bbb_linear_fwd_workspace_bytes(n_samples, out_features);
  const size_t ks = (1 + (size_t)n_samples * (size_t)sample_chunks(in_features, out_features)) * 16;
  return k1 > ks ? k1 : ks;
}

extern "C" int bnn_bbb_sample_weights(const bnn_bbb_sample_args* a, void* stream_) {
  SampleK k;
  long blocks = 0;
  const int rc = fill_sample(a, k, blocks);
  if (rc != BNN_OK) return rc;
  hipLaunchKernelGGL(bbb_sample_kernel, dim3((unsigned)blocks), dim3(kSampleThreads), 0, reinterpret_cast<hipStream_t>(stream_),
                     k);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? BNN_OK : (int)err;
}
