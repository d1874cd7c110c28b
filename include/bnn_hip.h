/*
 * bnn_hip.h — C ABI of libbnn_hip.so: the MI355X (gfx950) kernels behind the
 * Bayes-by-backprop hot path of tennisonliu/bayesian-neural-network.
 *
 * The reference has no FFI of its own: its boundary is the Python nn.Module surface of
 * networks.py.  This header is the build-side boundary underneath that surface; every
 * entry point names the reference lines (paths relative to the reference tree) whose
 * arithmetic it replaces.  Callers: bayesian-neural-network_amd/bnn_hip/_lib.py (ctypes).
 *
 * Conventions (all entry points)
 *   - extern "C", plain pointers and sizes, no C++/torch types, no exceptions.
 *   - Return int: 0 = BNN_OK, <0 = argument error (enum below), >0 = a hipError_t.
 *   - Every pointer is a caller-owned DEVICE pointer (contiguous, row-major, fp32 unless a
 *     dtype field says otherwise).  The library never allocates, frees, copies, synchronises
 *     or retains a pointer: launch functions only enqueue kernels on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream), so they are hipGraph-capturable.
 *   - Workspaces are caller-allocated; sizes come from the *_workspace_bytes queries.
 *   - Re-entrant for distinct streams/buffers; no global mutable state.
 *   - Structs start with `struct_bytes` = sizeof(the struct) as the caller compiled it;
 *     a mismatch returns BNN_ERR_ABI.
 *
 * Epsilon generator (map version BNN_EPS_MAP_VERSION; results must not depend on tiling, launch shape or #GPUs)
 *   Philox4x32 with BNN_PHILOX_ROUNDS = 7 rounds (Salmon et al., SC'11: the round function and key schedule of
 *   rocRAND's PHILOX4_32_10; 7 rounds is the variant the paper reports as passing BigCrush, 10 its default safety
 *   margin) keyed by the 64-bit seed.  Map version 1 (rounds 1-2 of this repository) ran 10 rounds: the kernels that
 *   draw eps per weight are bound by this integer arithmetic (20 -> 14 v_mad_u64_u32 per 4 normals; measured
 *   238 -> 197 cycles per 4 normals per SIMD, the per-8-weights body of K1b 533 -> 424: DESIGN.md 4), so version 2
 *   trades the margin for 20 % of the generator.  tests/test_oracle_golden.py checks the 10-round form of the SAME code
 *   against the Random123 known-answer vectors and the 7-round outputs for avalanche, bit balance, moments, a
 *   Kolmogorov-Smirnov distance and cross-stream correlation.
 *   For an epsilon tensor of logical shape [rows, cols] belonging to
 *   GLOBAL MC sample index g (= sample_offset + local sample), tensor id
 *   t = 4*layer_id + kind  (kind 0: BBB weight eps [out,in]; 1: bias eps [1,out];
 *   2: LR activation eps [batch,out]):
 *       group  = row * ceil(cols/4) + (col >> 2)           (uint32)
 *       counter = (group, g, t, 0), key = (seed_lo, seed_hi)
 *       (r0,r1,r2,r3) = Philox4x32-R(counter, key), R = BNN_PHILOX_ROUNDS
 *       u(r) = fma((float)r, 2^-32, 2^-33)                  in (0,1]
 *       slot 0,1 = sqrt(-2 ln u(r0)) * {cos, sin}(2 pi u(r1)); slot 2,3 likewise from r2,r3
 *       eps[row, col] = slot (col & 3)
 *   oracle/bnn_oracle.py:philox_normal restates this on the CPU.
 */
#ifndef BNN_HIP_H_
#define BNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BNN_HIP_ABI_VERSION 7
#define BNN_EPS_MAP_VERSION 2     /* 1: Philox4x32-10 (rounds 1-2); 2: Philox4x32-7 */
#ifndef BNN_PHILOX_ROUNDS          /* build-time choice (csrc/Makefile: make PHILOX_ROUNDS=10): 7 = map version 2 (the product), */
#define BNN_PHILOX_ROUNDS 7        /* 10 = rocRAND's PHILOX4_32_10 / map version 1.  bnn_philox_rounds() tells what a library runs */
#endif

enum bnn_status {
  BNN_OK = 0,
  BNN_ERR_NULL = -1,        /* required pointer is NULL */
  BNN_ERR_SHAPE = -2,       /* non-positive or unsupported dimension */
  BNN_ERR_ENUM = -3,        /* unknown dtype / mode / prior kind */
  BNN_ERR_WORKSPACE = -4,   /* workspace missing or too small */
  BNN_ERR_ABI = -5,         /* struct_bytes mismatch */
  BNN_ERR_ALIGN = -6        /* pointer not aligned to its element type */
};

enum bnn_dtype { BNN_F32 = 0, BNN_BF16 = 1 };

/* Arithmetic of the matmul.  BNN_MATH_F32: exact fp32 MFMA (v_mfma_f32_16x16x4_f32, a
 * k-ordered fmaf chain) — the parity mode.  BNN_MATH_BF16: operands rounded to bf16 (RNE),
 * fp32 accumulate (v_mfma_f32_16x16x32_bf16) — the throughput mode.  Statistics (log-probs,
 * KL) are always formed in fp32 from the un-rounded fp32 weights.  The x^2 operand of the LR variance product
 * (networks.py:121) in BNN_MATH_BF16 is bf16(bf16(x)^2) — the square of the rounded activation, rounded — for every kernel
 * form: the x_sq / y_sq / cast_dst_sq streams hold exactly what a kernel squaring the bf16 x it loaded computes, so which
 * forms a launch plan picks never changes a result bit. */
enum bnn_math { BNN_MATH_F32 = 0, BNN_MATH_BF16 = 1, BNN_MATH_BF16X3 = 2 };
/* BNN_MATH_BF16X3 (BBB forward): split-bf16 operands on the bf16 matrix core.  Every matmul operand v is carried as the pair
 *     hi = bf16(v),   lo = bf16(v - hi)            (both round-to-nearest-even; v - hi is exact in fp32)
 * and a product a . b is accumulated in fp32 as  a_hi b_hi + a_lo b_hi + a_hi b_lo  (three v_mfma_f32_16x16x32_bf16 per tile
 * and k-step; the lo.lo term, 2^-16 of the product, is dropped): |a b - (...)| <= ~2^-15 |a b| per product against 2^-8 for
 * BNN_MATH_BF16, i.e. the reference's fp32 F.linear (networks.py:88) to ~1e-5 of the output scale -- ELBO rtol 1e-4 at every
 * beta of classification/class_task.py:70 -- at 3/16 of the matrix-core time of the exact-fp32 MFMA.  bf16 activations of
 * this mode are PAIRS of planes: x / x_lo and y / y_lo below (fp32 activations are split in registers).  Statistics are
 * fp32 from the un-rounded weights as in every mode. */

enum bnn_eps_mode {
  BNN_EPS_PHILOX = 0,   /* generated on chip (map above); never touches HBM */
  BNN_EPS_MEMORY = 1,   /* read from eps_* buffers: identical-eps parity with the reference */
  BNN_EPS_ZERO = 2      /* eps = 0: w = mu (networks.py:78-79, eval & sample=False) */
};

enum bnn_prior_kind {
  BNN_PRIOR_GAUSS = 0,    /* Normal(0, sigma_p)                 networks.py:67-68 */
  BNN_PRIOR_MIXTURE = 1   /* pi N(0,sigma1) + (1-pi) N(0,sigma2) networks.py:14-27 */
};

enum bnn_nll_mode { BNN_NLL_REGRESSION = 0, BNN_NLL_CLASSIFICATION = 1 };

/* Kernel forms of a layer launch.  TILE: one block per (feature tile, sample, batch block), the block's
 * waves split the reduction and meet in LDS -- few samples in flight.  GEMM: block GEMM, the x tile shared
 * through LDS by LDS-DMA, a wave owns 16 features for all of K -- many samples.  GEMM_KSLICE (BBB): the GEMM
 * form with the reduction cut into slices whose fp32 partial tiles a second kernel sums in slice order.
 * BLOCK256 (BBB, matmul half over w_sampled only): one 8-wave block per 256 x 256 output tile, both operands
 * through LDS by LDS-DMA, ping-pong MFMA / load segments -- layers fed >= 512 batch rows, where the bf16
 * matrix cores, not the sampling, bound the layer (K1g, csrc/bbb_block_gemm.h). */
enum bnn_form { BNN_FORM_AUTO = 0, BNN_FORM_TILE = 1, BNN_FORM_GEMM = 2, BNN_FORM_GEMM_KSLICE = 3, BNN_FORM_BLOCK256 = 4 };

typedef struct bnn_prior {
  int32_t kind;      /* bnn_prior_kind */
  float sigma_p;     /* Gaussian prior scale (prior_init[0])            */
  float pi;          /* mixture weight       (prior_init[0] if mixture) */
  float sigma1;      /* exp(prior_init[1])                              */
  float sigma2;      /* exp(prior_init[2])                              */
} bnn_prior;

/* ------------------------------------------------------------------------------------
 * Launch plans.  The geometry of a forward layer launch is a pure function of the shape and of what the
 * arguments allow -- never of pointers' values, streams or the environment -- so one shape always runs one
 * summation order (bitwise reproducible), and the function is testable without a device.
 * bnn_bbb_plan / bnn_lr_plan return what bnn_bbb_linear_fwd / bnn_lr_linear_fwd will launch for `a`
 * (only the shape, dtype, math, form, alignment of the pointers and presence of the optional buffers are read).
 * ---------------------------------------------------------------------------------- */
typedef struct bnn_plan {
  int32_t form;          /* bnn_form taken */
  int32_t k_classes;     /* TILE: R k-range classes of the 16 MFMA rows; a tile holds 16 / R features */
  int32_t waves;         /* waves per block */
  int32_t batch_rows;    /* batch rows per block */
  int32_t k_slices;      /* GEMM_KSLICE: reduction slices (1 otherwise) */
  int32_t blocks;        /* work items (the grid is padded to a multiple of 8) */
  int32_t lds_bytes;     /* LDS per block */
  int32_t features_per_block;
} bnn_plan;

/* ------------------------------------------------------------------------------------
 * K1  bnn_bbb_linear_fwd — BayesianLinear.forward for n_samples MC samples in ONE launch.
 * Replaces networks.py:73-88 (+ :39-46 GaussianNode, :14-27 / :67-68 priors) and the
 * serial MC loop over it (networks.py:199-200) for one layer:
 *     w_s = mu + softplus(rho) * eps_s       (per weight and bias, never stored)
 *     y_s = x_s . w_s^T + b_s  [-> ReLU]     (networks.py:88, :169-171)
 *     stats partials for log p(w_s), log q(w_s)   (networks.py:82-83)
 * Shapes: x [x_rows, batch, in] (x_rows = 1 if x_per_sample == 0, else ceil(n_samples / x_per_sample));
 * w_mu,w_rho [out,in]; b_mu,b_rho [out]; eps_w [n_samples,out,in]; eps_b [n_samples,out];
 * y [n_samples, batch, out].
 *
 * Sample groups.  The n_samples of a launch may be G independent minibatches x S MC samples each
 * (sample s = minibatch s / S, MC sample s % S): the reference evaluates minibatches one after the
 * other (classification/class_task.py:66-79 trains, :89-103 evaluates), but given the parameters
 * their forward passes are independent, so a stream of minibatches is batched into one launch per
 * layer exactly like the MC samples of one minibatch.  x_per_sample = S then gives the first layer
 * one x per minibatch, and (sample_group = S_local, sample_group_stride = S_global) keeps the Philox
 * index of (minibatch m, global MC sample j) at sample_offset + m * S_global + j whatever share of
 * the S_global samples this rank owns (results independent of the number of GPUs).
 *
 * Stats workspace (want_stats != 0): opaque to the caller, produced here and consumed only
 * by this library (the optional per-layer reduction below and bnn_elbo_finalize).  It holds,
 * per (sample, feature tile), the fp32 partial sums {sum eps^2, sum w^2 (Gaussian prior) or
 * sum log p_mix(w) (mixture prior), sum log sigma} over the tile's weights and biases, plus a
 * one-entry header with the tile count, so the launch geometry may change between library
 * versions without touching callers.  One partial per block, no atomics.
 * If log_prior/log_q are non-NULL a second tiny kernel reduces the partials to the
 * layer's per-sample scalars float[n_samples] (the values BayesianLinear stores in
 * self.log_prior / self.log_variational_posterior).
 * ---------------------------------------------------------------------------------- */
struct bnn_bbb_sample_args;

typedef struct bnn_bbb_fwd_args {
  uint32_t struct_bytes;
  int32_t n_samples, batch, in_features, out_features;
  const void* x;
  int32_t x_dtype;          /* bnn_dtype */
  int32_t x_per_sample;     /* 0: one x for all samples; g >= 1: sample s reads x[s / g] (1: one x per sample) */
  const float* w_mu;
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  int32_t eps_mode;         /* bnn_eps_mode */
  int32_t math;             /* bnn_math */
  const float* eps_w;       /* BNN_EPS_MEMORY only */
  const float* eps_b;
  uint64_t seed;            /* BNN_EPS_PHILOX */
  uint32_t layer_id;
  uint32_t sample_offset;   /* global MC index of local sample 0 (multi-GPU shards) */
  const uint32_t* sample_counter; /* optional DEVICE word added to sample_offset at run time, so a
                               captured hipGraph draws fresh eps on every replay (see K4) */
  uint32_t sample_group;    /* 0: global index of local sample s = sample_offset + counter + s.  g > 0: */
  uint32_t sample_group_stride; /* sample_offset + counter + (s / g) * sample_group_stride + s % g      */
  float* eps_w_dump;        /* optional: the eps actually used, [n_samples,out,in] */
  float* eps_b_dump;        /* optional: [n_samples,out] */
  bnn_prior prior;
  int32_t want_stats;
  int32_t relu;             /* fuse ReLU (networks.py:161,163) */
  void* workspace;          /* stats partials, see above */
  size_t workspace_bytes;
  float* log_prior;         /* optional [n_samples] */
  float* log_q;             /* optional [n_samples] */
  void* y;
  int32_t y_dtype;          /* bnn_dtype */
  int32_t form;             /* bnn_form preference: BNN_FORM_AUTO lets the plan choose; another value is taken when
                               the arguments allow that form (bnn_bbb_plan tells), else the plan's own choice.
                               Tests compare the forms with each other through it; bnn_bbb_final_fwd fuses the
                               finalize only under BNN_FORM_AUTO */
  void* split_scratch;      /* optional, 16-byte aligned, >= bnn_bbb_split_scratch_bytes(n_samples, batch, out_features),
                               its first bnn_bbb_split_scratch_zero_bytes(...) bytes zero before the first launch that
                               uses it (launches leave them zero): lets a mid-sized launch (4 .. ~100 samples) split its
                               K range over several blocks; the block of a slice that finishes last adds the slices'
                               fp32 partial tiles up in slice order (bitwise reproducible), in the same launch.  One
                               scratch serves one launch at a time */
  size_t split_scratch_bytes;
  const float* w_sigma;     /* optional [out,in]: softplus(w_rho) from bnn_softplus, computed once per
                               evaluation; the throughput kernel then skips the per-sample softplus */
  const void* w_sampled;    /* optional bf16 [n_samples,out,in] from bnn_bbb_sample_weights: the launch is then the
                               matmul half only, y = act(x . w_sampled^T + b_sampled) -- no sampling, no statistics
                               (want_stats must be 0; w_mu .. eps_* are ignored and may be NULL).  bf16 math,
                               in_features % 8 == 0, 16-byte aligned x and w_sampled */
  const float* b_sampled;   /* with w_sampled: fp32 [n_samples,out] */
  const struct bnn_bbb_sample_args* rider; /* optional: an INDEPENDENT bnn_bbb_sample_weights(rider) job, carried by this
                               launch as extra blocks when it takes the tile form in bf16 math (launched on its own
                               ahead of the layer otherwise: same results).  Sampling depends on no activation: the
                               output layer's weights are drawn beside the layer before it, and the output layer of a
                               few-sample evaluation becomes a matmul-only launch (bnn_bbb_final_fwd, w_sampled) */
  void* y_bf16_copy;        /* optional bf16 [n_samples,batch,out], with y_dtype == BNN_F32: y also in bf16 (for the next
                               layer's forward of a training step, whose backward reads the fp32 y).  Tile form only:
                               selects it */
  void* w_sampled_t_out;    /* optional, with w_sampled and bf16 x: bf16 [n_samples,in,out], the same weights TRANSPOSED,
                               written by this launch.  The layer's backward (bnn_bbb_bwd_args.w_sampled_t) then computes its input
                               gradient as this same matmul-only launch instead of gathering along the reduction */
  const void* x_lo;         /* BNN_MATH_BF16X3 with x_dtype == BNN_BF16: the low plane of x (shaped and strided like x), i.e.
                               bf16(x_fp32 - x) as bnn_eval_prepare (cast_dst_lo) or a previous layer's y_lo left it */
  void* y_lo;               /* BNN_MATH_BF16X3 with y_dtype == BNN_BF16: the low plane of y, bf16(y_fp32 - y) after bias / ReLU */
} bnn_bbb_fwd_args;

size_t bnn_bbb_linear_fwd_workspace_bytes(int32_t n_samples, int32_t out_features);
size_t bnn_bbb_split_scratch_bytes(int32_t n_samples, int32_t batch, int32_t out_features);
size_t bnn_bbb_split_scratch_zero_bytes(int32_t n_samples, int32_t batch, int32_t out_features);
int bnn_bbb_linear_fwd(const bnn_bbb_fwd_args* args, void* stream);
int bnn_bbb_plan(const bnn_bbb_fwd_args* args, bnn_plan* plan);

/* ------------------------------------------------------------------------------------
 * K1s  bnn_bbb_sample_weights — the sampling half of BayesianLinear.forward (networks.py:73-86) for up to
 * BNN_SAMPLE_MAX_LAYERS layers and n_samples MC samples in ONE launch:
 *     w_out[s] = bf16(mu + softplus(rho) * eps_s)  [n_samples,out,in],   b_out[s] likewise, fp32 [n_samples,out]
 * with eps from the Philox map at the top (the same elements K1 would draw for these layer_ids / sample
 * indices), and the per-sample statistics {sum eps^2, sum w^2 | sum log p_mix, sum log sigma} (taken from the
 * fp32 w, as K1 does) in each layer's workspace in K1's format, so bnn_elbo_finalize / bnn_bbb_final_fwd consume
 * them unchanged.  The matmuls then run as bnn_bbb_linear_fwd(w_sampled, b_sampled).  Sampling depends on no
 * activation: out of the layer-after-layer chain of a few-sample evaluation it is one streaming pass over the
 * parameters (8 B read per weight and group of four samples -- a block serves the group from one read and one softplus --
 * + 2 B written per weight and sample).  in_features % 8 == 0; 16-byte aligned pointers.
 * ---------------------------------------------------------------------------------- */
#define BNN_SAMPLE_MAX_LAYERS 8
typedef struct bnn_bbb_sample_layer {
  int32_t in_features, out_features;
  uint32_t layer_id;
  int32_t reserved;
  const float* w_mu;        /* [out,in] */
  const float* w_rho;
  const float* b_mu;        /* [out] */
  const float* b_rho;
  void* w_out;              /* bf16 [n_samples,out,in] */
  float* b_out;             /* [n_samples,out] */
  void* workspace;          /* >= bnn_bbb_sample_workspace_bytes(n_samples, in, out) */
  size_t workspace_bytes;
  bnn_prior prior;
  int32_t reserved2;
} bnn_bbb_sample_layer;

typedef struct bnn_bbb_sample_args {
  uint32_t struct_bytes;
  int32_t n_layers;
  int32_t n_samples;
  uint32_t sample_offset;
  uint64_t seed;
  const uint32_t* sample_counter;   /* optional device word, as in bnn_bbb_fwd_args */
  uint32_t sample_group;            /* sample groups, as in bnn_bbb_fwd_args; 0 = none */
  uint32_t sample_group_stride;
  bnn_bbb_sample_layer layer[BNN_SAMPLE_MAX_LAYERS];
  const float* cast_src;            /* optional rider on the same launch: cast_dst[i] = bf16(cast_src[i]), */
  void* cast_dst;                   /* i < cast_n -- the evaluation's input batch for the bf16 matmuls      */
  int64_t cast_n;                   /* (16-byte aligned pointers); 0 = none                                  */
} bnn_bbb_sample_args;

/* >= bnn_bbb_linear_fwd_workspace_bytes(n_samples, out_features): one workspace serves either form; 0 when
 * in_features % 8 != 0 (the split form does not apply) */
size_t bnn_bbb_sample_workspace_bytes(int32_t n_samples, int32_t in_features, int32_t out_features);
int bnn_bbb_sample_weights(const bnn_bbb_sample_args* args, void* stream);

/* ------------------------------------------------------------------------------------
 * K3  bnn_lr_linear_fwd — BayesianLinearLR.forward (networks.py:116-138) for n_samples
 * MC samples in one launch: two MFMA GEMMs sharing the x tile,
 *     m = x . M,   v = x^2 . softplus(rho)^2          (networks.py:120-121)
 *     y = m + sqrt(v) * eps_act + (b_mu + softplus(b_rho) * eps_b)   (:123-128)
 * with the closed-form KL(q||p) partial sums (networks.py:109-114, :134-136) taken from
 * the same pass over (M, rho).  Weights are [in, out] (networks.py:95-96).
 * eps_act [n_samples,batch,out], eps_b [n_samples,out].
 * KL workspace (want_kl != 0): opaque, as for K1; per feature tile the fp32 partial sums
 * {sum log sigma, sum sigma^2, sum mu^2} over the tile's weights and biases.
 * kl_out (optional, float[3]) = {weight_kl + bias_kl, weight_kl, bias_kl}.
 * ---------------------------------------------------------------------------------- */
/* bnn_lr_rider — what bnn_lr_prepare does for one NARROW layer (out_features <= 16: the output layer of the reference's
 * networks, networks.py:160-164), carried by another layer's launch: the K-sliced form of bnn_lr_linear_fwd (K3s) runs it as a
 * few extra blocks on CUs its own blocks leave idle; any other form launches bnn_lr_prepare ahead of itself.  Either way,
 * after bnn_lr_linear_fwd returns (in stream order) `w_frag` holds the bf16 (M, sigma^2) operands in fragment order and
 * `kl_workspace` the layer's closed-form KL sums -- what bnn_lr_final_fwd takes as bnn_lr_fwd_args.w_frag / .workspace, so
 * that its row blocks park nothing (the one-evaluation chain: 12.4 -> 8 us for the 1200 x 10 layer). */
typedef struct bnn_lr_rider {
  uint32_t struct_bytes;
  int32_t in_features, out_features;
  const float* w_mu;        /* [in, out] */
  const float* w_rho;
  const float* b_mu;        /* [out] */
  const float* b_rho;
  void* w_frag;             /* >= bnn_lr_prepare_bytes(in, out), 16-byte aligned */
  size_t w_frag_bytes;
  void* kl_workspace;       /* >= bnn_lr_linear_fwd_workspace_bytes(out), 16-byte aligned */
  size_t kl_workspace_bytes;
} bnn_lr_rider;

typedef struct bnn_lr_fwd_args {
  uint32_t struct_bytes;
  int32_t n_samples, batch, in_features, out_features;
  const void* x;
  int32_t x_dtype;
  int32_t x_per_sample;     /* as in bnn_bbb_fwd_args */
  const float* w_mu;        /* [in,out] */
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  int32_t eps_mode;
  int32_t math;
  const float* eps_act;
  const float* eps_b;
  uint64_t seed;
  uint32_t layer_id;
  uint32_t sample_offset;
  const uint32_t* sample_counter; /* optional device word, as in bnn_bbb_fwd_args */
  uint32_t sample_group;    /* as in bnn_bbb_fwd_args */
  uint32_t sample_group_stride;
  float* eps_act_dump;
  float* eps_b_dump;
  float sigma_p;            /* prior_init[0]; prior mean is 0 (networks.py:103-104) */
  int32_t want_kl;
  int32_t relu;
  int32_t form;             /* bnn_form, as in bnn_bbb_fwd_args */
  void* workspace;
  size_t workspace_bytes;
  float* kl_out;
  void* y;
  int32_t y_dtype;
  int32_t reserved2;
  const void* x_sq;         /* optional bf16 [x_samples,batch,in]: the squares of x as a previous layer's y_sq or
                               bnn_cast_bf16 wrote them (BNN_MATH_BF16: bf16(bf16(x)^2), see bnn_math); lets the throughput kernel
                               stream both GEMM operands of networks.py:120-121 by LDS-DMA */
  void* y_sq;               /* optional bf16 [n_samples,batch,out]: also write the squares of y (after ReLU) for the next layer's x_sq */
  const void* w_frag;       /* optional output of bnn_lr_prepare for these weights: the throughput
                               kernel then streams ready bf16 (M, sigma^2) fragments and the KL
                               workspace is the one bnn_lr_prepare filled (want_kl still set) */
  float* v_out;             /* optional fp32 [n_samples,batch,out]: the pre-activation variance v
                               (networks.py:121) as the kernel computed it; bnn_lr_linear_bwd needs
                               it.  Selects the latency form of the kernel. */
  float* hfac_out;          /* optional fp32 [n_samples,batch,out]: eps_act / (2 sqrt(v)) (0 where v == 0), the factor
                               the backward multiplies the upstream gradient with (h = gz * hfac, see bnn_lr_bwd_args):
                               a hidden layer of a training step saves THIS instead of v and its backward needs no
                               preparation launch.  Selects the latency form of the kernel. */
  void* y_bf16_copy;        /* optional bf16 [n_samples,batch,out], with y_dtype == BNN_F32: y also in bf16.  A training
                               step keeps fp32 activations for the backward and feeds the next layer's forward
                               the bf16 ones (half the bytes through the CU, no conversion in its k loop).
                               Selects the latency form of the kernel. */
  const struct bnn_lr_rider* rider; /* optional: prepare ANOTHER (narrow) LR layer's operands beside this launch -- see bnn_lr_rider */
  void* split_scratch;      /* optional, 16-byte aligned, >= bnn_lr_split_scratch_bytes(n_samples, batch, out_features), its
                               first bnn_lr_split_scratch_zero_bytes(...) bytes zero before the first launch that uses it
                               (the kernel leaves them zero): lets a launch of 1-3 samples on a wide layer split the K
                               range of a 32-feature group over several blocks that meet through it (BNN_FORM_GEMM_KSLICE;
                               bf16 math on bf16 x, in_features % 8 == 0, out_features % 4 == 0).  One launch at a time
                               per scratch.
                               With x_per_sample == 0 and 2 .. 64 samples (no sample groups) -- the first layer of
                               sample_elbo_lr / predict, where the reference runs forward(x) per sample on the same x
                               (networks.py:211-225) -- the two products x M and x^2 sigma^2 are made ONCE per launch and
                               only the bias, the activation noise and the stores run per sample; the scratch then only
                               needs the sizes of n_samples = 1. */
  size_t split_scratch_bytes;
  const void* x_lo;         /* BNN_MATH_BF16X3: the low plane of x (as in bnn_bbb_fwd_args).  In that mode the layer runs the
                               block-GEMM form only: bf16 x with x_lo AND x_sq (in this mode bf16 of the fp32 squares), w_frag from
                               bnn_lr_prepare_x3; the mean product x . M (networks.py:120) in split-bf16 (three MFMAs), the
                               variance product x^2 . sigma^2 (:121) on bf16 operands as in BNN_MATH_BF16 -- the activation
                               noise is a few per cent of the output, its 2^-9 error below the mean's 2^-15 */
  void* y_lo;               /* BNN_MATH_BF16X3 with y_dtype == BNN_BF16: the low plane of y */
} bnn_lr_fwd_args;

size_t bnn_lr_linear_fwd_workspace_bytes(int32_t out_features);
size_t bnn_lr_split_scratch_bytes(int32_t n_samples, int32_t batch, int32_t out_features);
size_t bnn_lr_split_scratch_zero_bytes(int32_t n_samples, int32_t batch, int32_t out_features);
int bnn_lr_linear_fwd(const bnn_lr_fwd_args* args, void* stream);
int bnn_lr_plan(const bnn_lr_fwd_args* args, bnn_plan* plan);


/* bnn_lr_prepare — the eps-independent half of BayesianLinearLR.forward, once per ELBO
 * evaluation instead of once per MC sample (the reference recomputes it inside its sample loop,
 * networks.py:118-119, :134-136): sigma^2 = softplus(rho)^2, both GEMM operands rounded to bf16
 * and stored in MFMA fragment order, plus the closed-form KL sums into kl_workspace (optional). */
size_t bnn_lr_prepare_bytes(int32_t in_features, int32_t out_features);
int bnn_lr_prepare(const float* w_mu, const float* w_rho, const float* b_mu, const float* b_rho,
                   int32_t in_features, int32_t out_features, void* w_frag, size_t w_frag_bytes,
                   void* kl_workspace, size_t kl_workspace_bytes, void* stream);
/* The same for BNN_MATH_BF16X3: the fragments additionally carry the low part bf16(M - bf16(M)) of the mean operand
 * ([mean hi | variance | mean lo] per feature tile and k-step: 1.5 x the bytes). */
size_t bnn_lr_prepare_x3_bytes(int32_t in_features, int32_t out_features);
/* (ABI 7) The prepared operands of several layers in ONE launch: an evaluation's prepare launches depend on no activation, and
 * one launch per layer put ~4 us of launch boundary per layer ahead of the first layer's kernel.  Bitwise the fragments and KL
 * entries of n_jobs bnn_lr_prepare (x3 != 0: bnn_lr_prepare_x3) calls; n_jobs <= 8. */
typedef struct {
  const float* w_mu;
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  int32_t in_features, out_features;
  void* w_frag;
  size_t w_frag_bytes;
  void* kl_workspace;        /* optional, as in bnn_lr_prepare */
  size_t kl_workspace_bytes;
} bnn_lr_prepare_job;
int bnn_lr_prepare_many(const bnn_lr_prepare_job* jobs, int32_t n_jobs, int32_t x3, void* stream);
int bnn_lr_prepare_x3(const float* w_mu, const float* w_rho, const float* b_mu, const float* b_rho,
                      int32_t in_features, int32_t out_features, void* w_frag, size_t w_frag_bytes,
                      void* kl_workspace, size_t kl_workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * K2  bnn_gauss_kl — one streaming pass over (mu, rho)[n]: the eps-independent sums of
 * log q (networks.py:46, the -log sigma term) and the closed-form KL against
 * Normal(0, sigma_p) (networks.py:113).  out = float[4]:
 *   [0] KL = 0.5*sum(2 log(sigma_p/sigma) - 1 + (sigma/sigma_p)^2 + (mu/sigma_p)^2)
 *   [1] sum log sigma   [2] sum sigma^2   [3] sum mu^2
 * 8 algorithmic bytes per element; wavefront-shuffle + LDS block reduction, one partial
 * per block, fp64 final sum by a second one-block kernel (no float atomics: bitwise
 * reproducible).
 * ---------------------------------------------------------------------------------- */
size_t bnn_gauss_kl_workspace_bytes(int64_t n);
int bnn_gauss_kl(const float* mu, const float* rho, int64_t n, float sigma_p, void* workspace,
                 size_t workspace_bytes, float* out4, void* stream);

/* ------------------------------------------------------------------------------------
 * K4  bnn_elbo_finalize — per-sample scalars of one ELBO evaluation in one launch:
 * sums the stats partials of up to 8 BBB layers into log p / log q
 * (networks.py:174-178), or the KL partials of LR layers (networks.py:180-181), and the
 * NLL of the logits (networks.py:183-190: CrossEntropyLoss(reduction='sum') or the
 * Gaussian NLL with scale nll_sigma).  Outputs float[n_samples] each; any may be NULL.
 * For LR layers kl[] receives the same (eps-independent) value for every sample.
 * target: int64[batch] (classification) or float[batch,classes] (regression).
 * sample_counter: the launch functions bake their arguments into a captured hipGraph; the
 * Philox sample index therefore has a device-resident part that the layer kernels read and
 * this kernel (the last of an evaluation, stream-ordered after them) advances.
 * ---------------------------------------------------------------------------------- */
/* The tail of a training step's forward, optional in bnn_finalize_args: what bnn_elbo_loss_nll_bwd computes (loss
 * assembly networks.py:205-208, the backward seeds, d nll / d logits), done by the launch that finalizes when that
 * launch is bnn_bbb_final_fwd's row-split form (each row block differentiates its own rows' NLL, the last block to
 * arrive assembles the loss), else by a follow-up launch.  Arguments as for bnn_elbo_loss_nll_bwd. */
typedef struct bnn_loss_args {
  const float* beta;                /* device scalar */
  float total_samples;
  float grad_scale;
  float* out4;
  float* g_a;                       /* [n_samples]: d loss / d log p[s] (zeros for LR) */
  float* g_b;                       /* [n_samples]: d loss / d log q[s] */
  float* g_kl3;                     /* float[3] or NULL */
  float* g_logits;                  /* [n_samples,batch,classes] */
} bnn_loss_args;

typedef struct bnn_finalize_args {
  uint32_t struct_bytes;
  int32_t n_layers;                 /* 0..8 */
  int32_t local_reparam;            /* 0: BBB stats workspaces, 1: LR KL workspaces */
  int32_t n_samples, batch, classes;
  const void* layer_workspace[8];
  int32_t layer_in[8];
  int32_t layer_out[8];
  bnn_prior prior;
  const float* logits;              /* [n_samples,batch,classes] fp32, or NULL */
  const void* target;
  int32_t nll_mode;                 /* bnn_nll_mode */
  float nll_sigma;
  float* log_prior;                 /* BBB */
  float* log_q;                     /* BBB */
  float* kl;                        /* LR  */
  float* nll;
  uint32_t* sample_counter;         /* optional device word: += sample_counter_inc when done */
  uint32_t sample_counter_inc;      /* normally the GLOBAL number of MC samples of the evaluation */
  uint32_t reserved;
  float* sums;                      /* optional float[4] (float[G][4] with group_samples): sums over the local samples
                                       (of each minibatch) of {log p | KL, log q | 0, nll, sample count}: the vector
                                       a sharded job all-reduces (fixed summation order) */
  uint32_t* ticket;                 /* optional zero-initialised device word (left at zero again): lets 2..64 samples be
                                       finalized by one block each IN ONE launch, the last arriver folding `sums`
                                       (bnn_elbo_finalize and the fused last-layer form bnn_bbb_final_fwd) */
  void* scratch;                    /* optional, bnn_bbb_final_scratch_bytes(n_samples) bytes, 16-byte
                                       aligned, ZEROED ONCE by the caller: lets the fused last layer
                                       split its K range over several blocks per sample */
  size_t scratch_bytes;
  int32_t group_samples;            /* 0: the n_samples are one evaluation.  g > 0: n_samples = G * g, G independent
                                       minibatches of g MC samples each (see bnn_bbb_fwd_args): `sums` is then
                                       float[G][4], one 4-vector per minibatch */
  int32_t target_per_group;         /* with group_samples: 0 = one target for all, 1 = target[G][batch(,classes)] */
  const bnn_loss_args* loss;        /* optional (host pointer, read during the call): see bnn_loss_args.  Honoured by
                                       bnn_bbb_final_fwd; one evaluation only (group_samples == 0) */
} bnn_finalize_args;

int bnn_elbo_finalize(const bnn_finalize_args* args, void* stream);

/* bnn_lr_final_fwd — the LAST BayesianLinearLR layer of an evaluation together with its finalize
 * (networks.py:116-138 + :179-190): the same results as bnn_lr_linear_fwd(layer) followed by bnn_elbo_finalize(fin)
 * with fin->logits == layer->y, in ONE launch when the layer is narrow (<= 16 outputs, batch <= 128, bf16 math and x, on-chip
 * eps; <= 16 samples -- or, over prepared operands (layer->w_frag from bnn_lr_prepare[_many] / a rider), up to 4096 (minibatch,
 * sample) pairs, a block then taking two 16-row tiles and, beyond 64 pairs, the sums of the per-sample scalars made by a one-block
 * follow-up launch; fin->scratch as for bnn_bbb_final_fwd and, from 2 to 64 samples, fin->ticket): row blocks
 * compute the logits and the rows' NLL, one more block per sample the KL of all layers (this layer's from its
 * parameters: fin->layer_workspace[n_layers-1] is not read), the last block to arrive folds them.  Otherwise the two
 * launches. */
struct bnn_finalize_args;
int bnn_lr_final_fwd(const bnn_lr_fwd_args* layer, const struct bnn_finalize_args* fin, void* stream);

/* bnn_bbb_final_fwd — the LAST BBB layer of an evaluation together with its finalize: the
 * same results as bnn_bbb_linear_fwd(layer) followed by bnn_elbo_finalize(fin) with
 * fin->logits == layer->y and fin->layer_workspace[n_layers-1] == layer->workspace, but in ONE
 * launch when the layer is a single feature tile (out_features <= 16, batch <= 128): the block
 * that produced a sample's logits also forms its NLL (networks.py:183-190) and log p / log q
 * (networks.py:174-178).  Falls back to the two launches otherwise.
 * With layer->w_sampled / b_sampled (the layer's weights drawn earlier by bnn_bbb_sample_weights, its statistics in
 * fin->layer_workspace[n_layers-1]; bf16 math, <= 4096 samples (beyond 64 the sums come from a one-block follow-up launch),
 * fin->scratch given): the row-split form -- every
 * 16-row batch block of a sample is a block of its own (plain bf16 matmul + the rows' NLL), one more block sums the
 * layers' statistics, and the last block of the sample to finish folds the handful of scalars (write-through
 * stores + one arrival counter: no block waits, no fence). */
size_t bnn_bbb_final_scratch_bytes(int32_t n_samples);
int bnn_bbb_final_fwd(const bnn_bbb_fwd_args* layer, const bnn_finalize_args* fin, void* stream);

/* ------------------------------------------------------------------------------------
 * bnn_bbb_linear_bwd — backward of BayesianLinear for n_samples MC samples (what autograd
 * derives from networks.py:73-88 under classification/class_task.py:78 `loss.backward()`;
 * closed forms in SURVEY Appendix A.5).  eps is REGENERATED from the Philox map (or re-read
 * in BNN_EPS_MEMORY mode): no eps- or weight-sized tensor is kept from the forward pass.
 *   gz = gy * (y > 0) if relu;  gW_s = gz_s^T x_s;  t_s = gW_s + g_log_prior[s] * dlogp/dw(w_s)
 *   g_w_mu = sum_s t_s;  g_w_rho = (sum_s t_s eps_s - (sum_s g_log_q[s]) / sigma) * sigmoid(rho)
 *   (bias likewise with column sums of gz);  g_x[s] = gz_s . w_s  (optional).
 * All tensors fp32.  x [x_samples,batch,in]; gy, y [n_samples,batch,out]; g_x
 * [n_samples,batch,in].  workspace: bnn_bbb_linear_bwd_workspace_bytes (holds gz).
 * Weight gradients use the exact-fp32 matrix core; `math` selects the arithmetic of g_x only.
 * ---------------------------------------------------------------------------------- */
typedef struct bnn_bbb_bwd_args {
  uint32_t struct_bytes;
  int32_t n_samples, batch, in_features, out_features;
  const float* x;
  int32_t x_per_sample;
  int32_t relu;
  const float* gy;
  const float* y;             /* forward output (after ReLU); required when relu != 0 */
  const float* w_mu;
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  int32_t eps_mode;
  int32_t math;
  const float* eps_w;
  const float* eps_b;
  uint64_t seed;
  uint32_t layer_id;
  uint32_t sample_offset;
  bnn_prior prior;
  int32_t gx_relu_mask;       /* != 0: g_x is multiplied by (x > 0).  x is then the output of the ReLU layer below,
                                 whose own backward takes this g_x as gy with relu = 0 (no separate mask pass) */
  const float* g_log_prior;   /* [n_samples] or NULL (zeros) */
  const float* g_log_q;       /* [n_samples] or NULL (zeros) */
  float* g_w_mu;
  float* g_w_rho;
  float* g_b_mu;
  float* g_b_rho;
  float* g_x;                 /* optional */
  void* workspace;
  size_t workspace_bytes;
  const uint32_t* sample_counter; /* optional device word added to sample_offset at run time (the value
                                     the forward of the same step read), as in bnn_bbb_fwd_args */
  const void* w_sampled;      /* optional bf16 [n_samples,out,in]: the weights the forward of this step sampled
                                 (bnn_bbb_sample_weights).  g_x = gz . w_sampled is then a plain matmul instead of
                                 regenerating w through the transposed generator (bf16 math only); the weight
                                 gradients still regenerate eps */
  const void* w_sampled_t;    /* optional bf16 [n_samples,in,out] (the forward's w_sampled_t_out), with g_x: the input
                                 gradient is then the forward's matmul-only launch over it (bf16 math) */
  const void* gy_bf16;        /* optional bf16 copy of gy (the layer above's g_x_bf16): read by that launch when relu == 0 */
  void* g_x_bf16;             /* optional bf16 [n_samples,batch,in]: g_x also in bf16, for the layer below's gy_bf16 */
} bnn_bbb_bwd_args;

size_t bnn_bbb_linear_bwd_workspace_bytes(int32_t n_samples, int32_t batch, int32_t out_features);
int bnn_bbb_linear_bwd(const bnn_bbb_bwd_args* args, void* stream);

/* ------------------------------------------------------------------------------------
 * bnn_lr_linear_bwd — backward of BayesianLinearLR for n_samples MC samples (what autograd
 * derives from networks.py:116-138 under class_task.py:78 `loss.backward()`; closed forms in
 * SURVEY Appendix A.5).  eps_act / eps_b are REGENERATED from the Philox map.
 *   gz = gy * (y > 0) if relu;   h = gz * eps_act / (2 sqrt(v))
 *   g_w_mu  = sum_s x_s^T gz_s + c_w M / sigma_p^2
 *   g_w_rho = (2 sigma sum_s (x_s^2)^T h_s + c_w (sigma / sigma_p^2 - 1 / sigma)) * sigmoid(rho)
 *   g_b_mu  = sum_s colsum(gz_s) + c_b b_mu / sigma_p^2;   g_b_rho likewise with eps_b
 *   g_x[s]  = gz_s M^T + 2 x_s * (h_s (sigma^2)^T)          (optional)
 * with c_w = g_kl[0] + g_kl[1], c_b = g_kl[0] + g_kl[2]: the upstream gradients of the layer's
 * kl_out triple {kl, weight_kl, bias_kl} (device float[3]; NULL = zeros).
 * All tensors fp32.  x [x_samples,batch,in]; gy, y, v [n_samples,batch,out] (v = the v_out of
 * the forward call); weights [in,out]; g_x [n_samples,batch,in].  Exact-fp32 matrix core.
 * ---------------------------------------------------------------------------------- */
typedef struct bnn_lr_bwd_args {
  uint32_t struct_bytes;
  int32_t n_samples, batch, in_features, out_features;
  const float* x;
  int32_t x_per_sample;
  int32_t relu;
  const float* gy;
  const float* y;             /* forward output (after ReLU); required when relu != 0 */
  const float* v;             /* pre-activation variance saved by bnn_lr_linear_fwd (v_out) */
  const float* w_mu;
  const float* w_rho;
  const float* b_mu;
  const float* b_rho;
  int32_t eps_mode;
  int32_t math;               /* bnn_math.  BNN_MATH_BF16: the input gradient's two products take bf16-rounded operands
                                 (fp32 accumulation), as the forward of that mode does (out_features % 8 == 0); the
                                 weight gradients stay on the exact-fp32 matrix core.  A narrow output layer
                                 (<= 16 outputs, batch <= 128, on-chip eps, g_x wanted) is one launch of plain fp32
                                 FMAs in either mode */
  const float* eps_act;       /* BNN_EPS_MEMORY */
  const float* eps_b;
  uint64_t seed;
  uint32_t layer_id;
  uint32_t sample_offset;
  float sigma_p;
  int32_t gx_relu_mask;       /* as in bnn_bbb_bwd_args */
  const float* g_kl;          /* device float[3] or NULL */
  float* g_w_mu;
  float* g_w_rho;
  float* g_b_mu;
  float* g_b_rho;
  float* g_x;                 /* optional */
  void* workspace;
  size_t workspace_bytes;
  const uint32_t* sample_counter; /* optional device word, as in bnn_bbb_bwd_args */
  const float* hfac;              /* optional, instead of v (which may then be NULL), with relu == 0: the forward's
                                     hfac_out.  gz = gy and h = gy * hfac are formed as the kernels load them: no
                                     preparation launch, eps_act is not regenerated, the workspace is not used */
} bnn_lr_bwd_args;

size_t bnn_lr_linear_bwd_workspace_bytes(int32_t n_samples, int32_t batch, int32_t in_features,
                                         int32_t out_features, int32_t want_gx);
int bnn_lr_linear_bwd(const bnn_lr_bwd_args* args, void* stream);

/* ------------------------------------------------------------------------------------
 * F2  bnn_adam_step — torch.optim.Adam.step() (the optimiser the reference's trainers build,
 * classification/class_task.py:60, :79; regression/reg_task.py:53) over up to
 * BNN_ADAM_MAX_TENSORS parameter tensors in ONE launch:
 *     g = grad + weight_decay * p;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2
 *     p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * t = step (host) or, when step_device != NULL, the device word *step_device, advanced first if
 * step_advance != 0 (so a captured hipGraph of the training step counts by itself).  lr_device (optional device float)
 * overrides lr, so StepLR (class_task.py:61) can change the rate of a captured graph.
 * All tensors fp32 (grad: fp32 or bf16, grad_dtype), contiguous, 16-byte aligned.
 * ---------------------------------------------------------------------------------- */
#define BNN_ADAM_MAX_TENSORS 16
typedef struct bnn_adam_args {
  uint32_t struct_bytes;
  int32_t n_tensors;
  float* param[BNN_ADAM_MAX_TENSORS];
  const float* grad[BNN_ADAM_MAX_TENSORS];
  float* exp_avg[BNN_ADAM_MAX_TENSORS];
  float* exp_avg_sq[BNN_ADAM_MAX_TENSORS];
  int64_t numel[BNN_ADAM_MAX_TENSORS];
  double lr, beta1, beta2, eps, weight_decay;   /* doubles, as torch holds them: 1 - beta2 must not be
                                                   formed from a float-rounded beta2 */
  uint32_t step;                 /* 1-based step number when step_device == NULL */
  uint32_t bump_by;              /* see bump_counter */
  const float* lr_device;
  uint32_t* step_device;
  int32_t step_advance;          /* with step_device: 1 = ++(*step_device) before the update, 0 = use it
                                    as it is (second and later launches of one optimiser step) */
  int32_t grad_dtype;            /* bnn_dtype of grad[]: BNN_F32, or BNN_BF16 -- grad[t] then points at bf16 values (the
                                    gradient bucket a data-parallel step all-reduces in 2-byte elements) */
  uint32_t* ticket;              /* optional zero-initialised device array of 16 words.  With step_advance: the update uses
                                    *step_device + 1 and the block that finishes last stores it (and re-zeroes the
                                    ticket) -- the step counts inside the one launch, no separate tick launch */
  uint32_t* bump_counter;        /* optional (needs ticket): *bump_counter += bump_by by that same last block, e.g. the
                                    MC-sample counter of a captured training step, advanced after the backward read it */
} bnn_adam_args;

int bnn_adam_step(const bnn_adam_args* args, void* stream);

/* bnn_elbo_loss — the ELBO of networks.py:205-208 (BBB: a = log p, b = log q per sample) or
 * :222-224 (LR: a = KL per sample, b = NULL) from per-sample scalars and a DEVICE beta, with
 * total_samples = S of the whole job:  out4 = {loss, mean a, mean b, mean nll}, and the seeds of the
 * backward chain g_a[s] = -beta/S (0 for LR), g_b[s] = beta/S, g_nll[s] = 1/S, g_kl3 = {beta, 0, 0}
 * (what bnn_*_linear_bwd / bnn_nll_bwd take), all times grad_scale (1, or 1 / ranks when the
 * gradients of data-parallel minibatches are then SUM-all-reduced).  Any seed pointer may be NULL. */
int bnn_elbo_loss(const float* a, const float* b, const float* nll, const float* beta, int32_t n_samples,
                  float total_samples, float grad_scale, int32_t local_reparam, float* out4, float* g_a, float* g_b,
                  float* g_nll, float* g_kl3, void* stream);

/* bnn_nll_bwd — gradient of the summed NLL (networks.py:183-190) w.r.t. the logits of every MC
 * sample, scaled by g_nll[s] (device float[n_samples]):
 *   classification: (softmax(logits_s) - onehot(target)) * g_nll[s]
 *   regression:     (logits_s - target) / nll_sigma^2 * g_nll[s]
 * logits, g_logits fp32 [n_samples,batch,classes]; target as for bnn_elbo_finalize. */
int bnn_nll_bwd(const float* logits, const void* target, const float* g_nll, float* g_logits, int32_t n_samples,
                int32_t batch, int32_t classes, int32_t nll_mode, float nll_sigma, void* stream);

/* bnn_elbo_loss_nll_bwd — bnn_elbo_loss and bnn_nll_bwd of one training step in ONE launch: the NLL seed is the
 * constant grad_scale / total_samples, so the logits' gradient does not wait for the loss arithmetic.
 * Arguments as for the two functions (no g_nll: it is not materialised). */
int bnn_elbo_loss_nll_bwd(const float* a, const float* b, const float* nll, const float* beta, int32_t n_samples,
                          float total_samples, float grad_scale, int32_t local_reparam, float* out4, float* g_a, float* g_b,
                          float* g_kl3, const float* logits, const void* target, float* g_logits, int32_t batch,
                          int32_t classes, int32_t nll_mode, float nll_sigma, void* stream);

/* bnn_stage_inputs — the per-step inputs of a captured training step (the minibatch the reference's loop hands
 * to sample_elbo, class_task.py:72-77, and its KL weight beta) copied into the graph's static device buffers
 * in one launch: dst0 <- src0 (bytes0), dst1 <- src1 (bytes1), *word = value.  Device pointers; any part may be
 * empty (bytes = 0 / word = NULL). */
int bnn_stage_inputs(const void* src0, void* dst0, size_t bytes0, const void* src1, void* dst1, size_t bytes1,
                     float* word, float value, void* stream);
/* The same with src0 an fp32 tensor whose bf16 copy (bytes0 / 2 bytes) is also written to cast0_bf16: the first layer's
 * forward of a bf16-math step reads that one.  16-byte aligned pointers and sizes. */
int bnn_stage_inputs_cast(const void* src0, void* dst0, size_t bytes0, const void* src1, void* dst1, size_t bytes1,
                          float* word, float value, void* cast0_bf16, void* stream);

/* ------------------------------------------------------------------------------------
 * F3  bnn_mc_softmax_mean — the MC-averaged prediction of classification/class_task.py:81-87:
 *     probs[b, :] = scale * sum_s softmax(logits[s, b, :]),   preds[b] = argmax_c probs[b, c]
 * (scale = 1 / test_samples; a rank of a sharded job passes its local samples and the global
 * scale, then sum-all-reduces probs).  logits fp32 [n_samples,batch,classes]; probs fp32
 * [batch,classes]; preds int64[batch] or NULL.
 * ---------------------------------------------------------------------------------- */
int bnn_mc_softmax_mean(const float* logits, int32_t n_samples, int32_t batch, int32_t classes, float scale,
                        float* probs, long long* preds, void* stream);

/* ------------------------------------------------------------------------------------
 * F3  bnn_ece — ECELoss.forward of compute_ece.py:14-57 over the MC-averaged class probabilities the predict path
 * produces (classification/class_task.py:81-87 -> compute_ece.py:62-78): every one of the n x classes probabilities
 * is binned (np.digitize(p, bin_edges, right=True) - 1), a probability is "correct" when it is its row's argmax and
 * that argmax is the label; ece = sum_b |mean confidence_b - accuracy_b| * count_b / sum_b count_b.
 * probs fp32 [n, classes]; labels int64[n]; bin_edges: HOST float64 array of n_edges <= 65 increasing values
 * (np.arange(0, 1.1, bin_step) for the reference's bins);  out fp32 [1 + 3 * (n_edges - 1)] = {ece, then per bin
 * count, corrects, confidence sum}.  Bins without data contribute nothing (the reference's loop breaks there).
 * Integer counts by LDS atomics, fp64 confidence sums in a fixed order: bitwise reproducible.
 * ---------------------------------------------------------------------------------- */
size_t bnn_ece_workspace_bytes(void);
int bnn_ece(const float* probs, const long long* labels, int64_t n, int32_t classes, const double* bin_edges,
            int32_t n_edges, void* workspace, size_t workspace_bytes, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * F4  signal-to-noise pruning of weight_pruning.py:85-115 over n contiguous fp32 (mu, rho) pairs:
 *   bnn_snr_db     out[i] = 10 log10(|mu[i]| / log1p(exp(rho[i])))           (:85-87 compute_snr, :100, :109)
 *   bnn_snr_prune  in place mu[i] *= m, rho[i] *= m with m = (snr_db > threshold)   (:101-105, :110-114);
 *                  a pruned weight is left at rho = 0 (sigma = log 2), as the reference leaves it.
 *                  kept (optional device uint64, caller-zeroed): += number of survivors.
 * The threshold is the caller's np.percentile(snrs, 100 * drop_percentage) (:92).
 * ---------------------------------------------------------------------------------- */
int bnn_snr_db(const float* mu, const float* rho, int64_t n, float* out, void* stream);
int bnn_snr_prune(float* mu, float* rho, int64_t n, float threshold, unsigned long long* kept, void* stream);

/* ------------------------------------------------------------------------------------
 * bnn_philox_normal — materialise the on-chip epsilon stream (map at the top) into
 * eps[n_samples, rows, cols]: used by the backward pass to regenerate eps instead of
 * storing it, and by tests to check the frozen counter->element map.
 * ---------------------------------------------------------------------------------- */
int bnn_philox_normal(float* eps, uint64_t seed, uint32_t tensor_id, uint32_t sample_offset,
                      int32_t n_samples, int32_t rows, int32_t cols, void* stream);

/* ------------------------------------------------------------------------------------
 * bnn_cast_bf16 — fp32 -> bf16 (round to nearest even) of n contiguous elements: the input
 * batch is cast once per ELBO evaluation when bf16 math runs many MC samples, so every
 * layer streams 2-byte activations.  (The reference keeps x in fp32, main.py / class_task.py:71.)
 * ---------------------------------------------------------------------------------- */
/* sigma[i] = log1p(exp(rho[i])) (networks.py:39), n contiguous fp32 elements. */
int bnn_softplus(const float* rho, float* sigma, int64_t n, void* stream);

int bnn_cast_bf16(const float* src, void* dst_bf16, void* dst_sq_bf16 /* optional: bf16(bf16(x)^2) */, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * bnn_eval_prepare — the two passes above for a whole evaluation in ONE launch: everything that depends on no
 * activation.  sigma[i] = softplus(rho[i]) for up to BNN_PREPARE_MAX parameter tensors (networks.py:37-39 evaluates
 * it three times per node and forward; here once per evaluation, then shared by all MC samples through
 * bnn_bbb_fwd_args.w_sigma) and, optionally, the bf16 cast (+ squares) of the input batch.  Elementwise: the same
 * values as bnn_softplus / bnn_cast_bf16.
 * ---------------------------------------------------------------------------------- */
#define BNN_PREPARE_MAX 8
typedef struct bnn_prepare_args {
  uint32_t struct_bytes;
  int32_t n_softplus;                     /* 0 .. BNN_PREPARE_MAX */
  const float* rho[BNN_PREPARE_MAX];
  float* sigma[BNN_PREPARE_MAX];
  int64_t n[BNN_PREPARE_MAX];             /* elements of each tensor */
  const float* cast_src;                  /* optional (cast_n > 0) */
  void* cast_dst;                         /* bf16 */
  void* cast_dst_sq;                      /* optional bf16 squares: bf16(bf16(src)^2); with cast_dst_lo (BNN_MATH_BF16X3) bf16(src^2) */
  int64_t cast_n;
  void* cast_dst_lo;                      /* optional bf16: the low plane bf16(src - cast_dst) of BNN_MATH_BF16X3 */
} bnn_prepare_args;
int bnn_eval_prepare(const bnn_prepare_args* args, void* stream);

int bnn_version(void);                    /* BNN_HIP_ABI_VERSION the library was built with */
int bnn_philox_rounds(void);              /* BNN_PHILOX_ROUNDS the library was built with (7, or 10 for rocRAND's generator) */
const char* bnn_status_string(int status); /* static string for a negative status */

#ifdef __cplusplus
}
#endif
#endif /* BNN_HIP_H_ */
