// K1g -- the matrix-core-bound form of the BayesianLinear matmul (networks.py:88, F.linear(input, weight, bias)) over
// weights drawn once per launch by bnn_bbb_sample_weights:   y[s] = act(x[s] . w[s]^T + b[s])
// for layers fed >= 512 batch rows, where 2 * batch flops per sampled weight make the bf16 matrix cores the bound
// (the C5 "MFMA-bound roofline point").  Both operands are K-contiguous (x [rows, K], w [N, K]): an NT GEMM.
//
// Decomposition.  One 512-thread block (8 waves, 2 per SIMD) per 256 x 256 output tile (batch rows x features), one
// block per CU, the whole K range in the block.  K is walked in tiles of 64; a K-tile of an operand is two HALF-TILES of
// 128 rows x 128 B (16 KiB): X0/X1 (batch rows 0-127 / 128-255 of the block), W0/W1 (features).  LDS holds two K-tiles
// (2 x 64 KiB); every half-tile is brought by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers), two
// wave-instructions per wave, each fetching 8 whole 128-byte lines and filling 1 KiB of LDS.
//   LDS image of a half-tile: row-major 128-byte rows, the eight 16-byte chunks of row r stored at chunk slot
//   c ^ ((r >> 1) & 7).  LDS-DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE address; the
//   fragment reads apply the same XOR.  With it the 16 lanes of every ds_read_b128 service group (MI355X LDS: 4 groups
//   of 16 lanes, 64 banks of 4 B) hit 16 different 16-byte bank slots: conflict-free (the algebra is in DESIGN.md 4).
// Wave (wr = wave >> 2, wc = wave & 3) owns batch rows {64 wr .. 64 wr + 63} of BOTH X half-tiles and features
// {32 wc .. 32 wc + 31} of BOTH W half-tiles: 128 x 64 outputs as four QUADRANTS (X half mh, W half nh) of 64 x 32 =
// 4 x 2 accumulator tiles of v_mfma_f32_16x16x32_bf16 (A operand = 16 features x 32 k of w, B operand = 32 k x 16 rows
// of x, so a lane ends up with 4 consecutive FEATURES of one batch row: 8 / 16-byte y stores).
//
// Schedule.  A K-tile is a sequence of PHASES, each
//             LOAD segment  { ds_read_b128 fragments ; issue LDS-DMA half-tiles ; counted s_waitcnt vmcnt }
//             s_barrier
//             MFMA segment  { s_waitcnt lgkmcnt(0) ; one or two quadrants of 16 MFMAs }
//             s_barrier
// Waves 4-7 run one barrier behind waves 0-3 (they pass one extra s_barrier before the loop), so on every SIMD one
// wave is in its MFMA segment while its partner is in its LOAD segment: the matrix pipe is never shared and the LDS /
// DMA issue of one wave hides behind the other's MFMAs.  Group g in {0,1} runs LOAD(p) in barrier interval 2p + g and
// MFMA(p) in interval 2p + g + 1, so a half-tile may be re-filled from two phases after its last read on (the later
// group's reads retire -- lgkmcnt(0) at the top of its MFMA segment -- before the barrier that precedes the earlier
// group's issue), and LDS-DMA data may be read one phase after the phase whose LOAD segment waited for it (the wait sits
// before that segment's barrier; nothing else orders a ds_read behind another wave's DMA).  vmcnt never drains to zero
// inside the loop.
//   Two-phase form (BG_PHASES 2, the product build: 104.5 against 109.2 us at 4 x 1024 x 4096 x 4096 -- half the barriers,
//   32-MFMA segments), tile u in buffer u & 1:
//     A: reads W0 W1 X0 (16), issues X0 W0 W1 of tile u+1, vmcnt(6) = "X1 of tile u has landed";  quadrants (0,0) (0,1)
//     B: reads X1 (8),        issues X1 of tile u+1,       vmcnt(2) = "the three of phase A have"; quadrants (1,1) (1,0)
//   Four-phase form (BG_PHASES 4: 16-MFMA segments, finer-grained buffer release, four half-tiles = 64 KiB per CU in flight):
//     ph0 reads W0 X0 (12) -> (0,0); ph1 reads W1 (4) -> (0,1); ph2 reads X1 (8) -> (1,1); ph3 reads nothing -> (1,0);
//     ph0 of tile u issues W1 of tile u+1, ph1: X1 of u+1, ph2: X0 of u+2, ph3: W0 of u+2; every phase waits vmcnt(8)
//     after its issue = "my two pieces issued four phases ago have landed"; the last two K-tiles issue less, so their
//     waits count down (8, 8, 6, 4 | 2, 0).
// What bounds it (tools/block_gemm_bench.hip with BG_ABL builds, tools/mfma_ceiling.hip; 4 x 1024 x 4096 x 4096, one block per
// CU): the full kernel 104-109 us (0.50-0.53 of 2.5 PF), matrix pipe 63-65 % busy at 2.0-2.1 GHz; without the y stores 98
// (every block reaches its epilogue at the same time: 33.5 MB at 3.3 TB/s with all matrix pipes idle); without LDS reads,
// DMA and barriers 90; with nothing but the MFMAs left 81 (0.68).  A bare kernel issuing the same 64 MFMAs per K-tile in
// the same ping-pong skeleton takes 69-73 us for 64 K-tiles (0.76-0.80; 0.82-0.85 in long launches, pipe 94-97 % busy at
// 2.1-2.2 GHz): the remaining 12-15 % of this kernel's MFMA-only loop is not explained (DESIGN.md 4 lists what was
// ruled out).  The vendor BLAS runs the shape at 0.59-0.65 on the same device (tools/block_gemm_vs_library.py).
// Edges: rows past the batch / feature count are clamped in the source address (computed, never stored); K % 64 != 0:
// the lanes whose chunk lies past K get an out-of-range buffer offset and the hardware writes zeros (K % 8 == 0 is
// required, so a chunk is in or out as a whole).
#pragma once
#include "bnn_device.h"
#include <type_traits>

namespace bnn {

struct BlockGemmK {
  const __bf16* x;      // [rows of S / xg][M, K] bf16
  long x_sstride;       // elements between x row blocks (0: one x for all samples)
  int xg;               // samples sharing one x row block
  const __bf16* w;      // [S, N, K] bf16 sampled weights
  const float* bias;    // [S, N] sampled biases or nullptr
  void* y;              // [S, M, N] fp32 or bf16
  int y_bf16, relu;
  int S, M, N, K;
  int MT, NT;           // 256-row / 256-feature tiles per sample
};

#ifndef BG_PHASES
#define BG_PHASES 2   // 2: two 32-MFMA segments per K-tile (measured faster: 104.5 vs 109.2 us at 4 x 1024 x 4096 x 4096); 4: four 16-MFMA segments
#endif
#ifndef BG_ABL
#define BG_ABL 0   // development ablations (tools/block_gemm_bench.hip only): 1 = no LDS reads after the first K-tile, 2 = no LDS-DMA after the prologue, 4 = no barriers, 8 = no y stores, 16 = fragments read once before the loop (no read code in it), 32 = no s_waitcnt in the loop, 64 = no s_setprio
#endif
constexpr int kBgThreads = 512;
constexpr int kBgLds = 128 * 1024;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bg_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

#define BG_LDSP(off) ((__attribute__((address_space(3))) void*)(bg_lds + (off)))

template <bool KTAIL, int PH = 4>
__global__ __launch_bounds__(kBgThreads, 2) void bbb_block_gemm_kernel(const BlockGemmK p) {
  extern __shared__ __attribute__((aligned(1024))) char bg_lds[];   // [buffer 2][X0 X1 W0 W1][16 KiB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q = lane >> 4;

  // ---- work item: sample-major, then bands of 4 batch tiles, then feature tile, then batch tile in the band; XCD x
  // (blocks b = x mod 8) owns a contiguous range of that list, so the 32 blocks an XCD runs together are 8 feature
  // tiles x 4 batch tiles sharing 12 operand panels through its L2 (speed only).
  const int per_sample = p.MT * p.NT;
  const long total = (long)p.S * per_sample;
  const long chunk = (total + 7) >> 3;
  const long item = (long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if ((long)(blockIdx.x >> 3) >= chunk || item >= total) return;
  const int s = (int)(item / per_sample);
  int rem = (int)(item - (long)s * per_sample);
  const int band = rem / (4 * p.NT);
  rem -= band * 4 * p.NT;
  const int band_rows = min(4, p.MT - band * 4);
  const int nt = rem / band_rows, mt = band * 4 + rem % band_rows;
  const int m0 = mt * 256, n0 = nt * 256;
  const int K = p.K, M = p.M, N = p.N;
  const int nkt = (K + 63) >> 6;

  const __bf16* xs = p.x + (size_t)(s / p.xg) * (size_t)p.x_sstride;
  const __bf16* wsm = p.w + (size_t)s * (size_t)N * K;
  const __amdgpu_buffer_rsrc_t rx = bg_rsrc(xs, (uint32_t)((size_t)M * K * 2));
  const __amdgpu_buffer_rsrc_t rw = bg_rsrc(wsm, (uint32_t)((size_t)N * K * 2));

  // ---- staging: piece i (0,1) of a half-tile = its 8-row group 2 * wave + i; lane l -> row l >> 3, chunk slot l & 7
  // holding logical chunk (l & 7) ^ ((row >> 1) & 7) = (l & 7) ^ (4 i + (l >> 4))
  const int c0 = (lane & 7) ^ (lane >> 4);                 // i = 0; i = 1: c0 ^ 4
  uint32_t vx[2][2], vw[2][2];                             // [half][piece] byte offsets of the lane's 16 bytes at K-tile 0
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = h * 128 + (wave * 2 + i) * 8 + (lane >> 3);
      const int c = c0 ^ (4 * i);
      vx[h][i] = (uint32_t)min(m0 + row, M - 1) * (uint32_t)(K * 2) + (uint32_t)(c * 16);
      vw[h][i] = (uint32_t)min(n0 + row, N - 1) * (uint32_t)(K * 2) + (uint32_t)(c * 16);
    }
  const int kc0 = c0 * 8, kc1 = (c0 ^ 4) * 8;              // first k of the lane's chunk (pieces 0 / 1)
  // op: 0 = X, 1 = W; h: half; kt: K-tile; into buffer kt & 1
  auto stage = [&](int op, int h, int kt) __attribute__((always_inline)) {
    if ((BG_ABL & 2) && kt >= 2) return;
    const uint32_t ldso = (uint32_t)((kt & 1) * 65536 + op * 32768 + h * 16384 + wave * 2048);
    const int kb = kt * 128;
    uint32_t v0 = op ? vw[h][0] : vx[h][0], v1 = op ? vw[h][1] : vx[h][1];
    if (KTAIL) {
      v0 = (kt * 64 + kc0 < K) ? v0 : 0x80000000u;
      v1 = (kt * 64 + kc1 < K) ? v1 : 0x80000000u;
    }
    if (op) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, BG_LDSP(ldso), 16, v0, kb, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, BG_LDSP(ldso + 1024), 16, v1, kb, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, BG_LDSP(ldso), 16, v0, kb, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, BG_LDSP(ldso + 1024), 16, v1, kb, 0, 0);
    }
  };

  // ---- fragment read addresses (bytes within a half-tile): row * 128 + ((4 kh + q) ^ ((r >> 1) & 7)) * 16
  const int sw = (r >> 1) & 7;
  uint32_t xa[2], wa[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    xa[kh] = (uint32_t)((wr * 64 + r) * 128 + (((4 * kh + q) ^ sw) << 4));
    wa[kh] = (uint32_t)((wc * 32 + r) * 128 + (((4 * kh + q) ^ sw) << 4)) + 32768u;
  }

  f32x4 acc[2][4][2][2];                                   // [X half][batch tile][W half][feature tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xf[4][2];                                         // [batch tile][k half] of the current X half
  bf16x8 wf[2][2][2];                                      // [W half][feature tile][k half]

  auto rd = [&](uint32_t a) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(bg_lds + a);
  };
  bool abl_skip = false;
  bool abl_pre = false;
  auto read_x = [&](int buf, int mh) __attribute__((always_inline)) {
    if ((BG_ABL & 16) && !abl_pre) return;
    if ((BG_ABL & 1) && abl_skip) return;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) xf[mi][kh] = rd(xa[kh] + (uint32_t)(buf * 65536 + mh * 16384 + mi * 2048));
  };
  auto read_w = [&](int buf, int nh) __attribute__((always_inline)) {
    if ((BG_ABL & 16) && !abl_pre) return;
    if ((BG_ABL & 1) && abl_skip) return;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) wf[nh][ni][kh] = rd(wa[kh] + (uint32_t)(buf * 65536 + nh * 16384 + ni * 2048));
  };
  auto quad = [&](int mh, int nh) __attribute__((always_inline)) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mh][mi][nh][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nh][ni][kh], xf[mi][kh], acc[mh][mi][nh][ni], 0, 0, 0);
  };
#define BG_BAR()                                \
  do {                                          \
    __builtin_amdgcn_sched_barrier(0);          \
    if (!(BG_ABL & 4)) __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)
#define BG_VM(N) do { if (!(BG_ABL & 32)) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); } while (0)
#define BG_LGKM0() do { if (!(BG_ABL & 32)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#define BG_PRIO(P) do { if (!(BG_ABL & 64)) __builtin_amdgcn_s_setprio(P); } while (0)
  // MFMA segment of a phase
#define BG_MFMA(MH, NH)                                     \
  do {                                                      \
    BG_LGKM0();                                             \
    __builtin_amdgcn_sched_barrier(0);                      \
    BG_PRIO(1);                                             \
    quad(MH, NH);                                           \
    BG_PRIO(0);                                             \
  } while (0)

  // ---- prologue: K-tile 0 whole, X0 / W0 of K-tile 1
  stage(0, 0, 0);
  stage(1, 0, 0);
  stage(1, 1, 0);
  stage(0, 1, 0);
  if (PH == 4 && nkt > 1) {
    stage(0, 0, 1);
    stage(1, 0, 1);
    BG_VM(4);
  } else {
    BG_VM(0);
  }
  BG_BAR();
  if (wr == 1) BG_BAR();                                   // waves 4-7: one barrier behind
  if (BG_ABL & 16) {
    abl_pre = true;
    read_w(0, 0);
    read_w(0, 1);
    read_x(0, 0);
    abl_pre = false;
  }

  // MODE 0: steady (every phase issues); 1: second-to-last K-tile (ph0, ph1 issue); 2: last K-tile (no issue)
  auto ktile = [&](int u, auto mode_) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_)::value;
    const int buf = u & 1;
    // ph0
    read_w(buf, 0);
    read_x(buf, 0);
    if (MODE <= 1) stage(1, 1, u + 1);
    if (MODE <= 1) BG_VM(8); else BG_VM(2);
    BG_BAR();
    BG_MFMA(0, 0);
    BG_BAR();
    // ph1
    read_w(buf, 1);
    if (MODE <= 1) stage(0, 1, u + 1);
    if (MODE <= 1) BG_VM(8); else BG_VM(0);
    BG_BAR();
    BG_MFMA(0, 1);
    BG_BAR();
    // ph2
    read_x(buf, 1);
    if (MODE == 0) stage(0, 0, u + 2);
    if (MODE == 0) BG_VM(8); else if (MODE == 1) BG_VM(6);
    BG_BAR();
    BG_MFMA(1, 1);
    BG_BAR();
    // ph3
    if (MODE == 0) stage(1, 0, u + 2);
    if (MODE == 0) BG_VM(8); else if (MODE == 1) BG_VM(4);
    BG_BAR();
    BG_MFMA(1, 0);
    BG_BAR();
  };
  // Two-phase form of a K-tile (32 MFMAs per segment, half the barriers; one K-tile of DMA run-ahead):
  //   A: reads W0 W1 X0 (16), issues X0 W0 W1 of tile u+1, waits for X1 of tile u;  quadrants (0,0) (0,1)
  //   B: reads X1 (8), issues X1 of tile u+1, waits for the three of phase A;        quadrants (1,1) (1,0)
  auto ktile2 = [&](int u, auto last_) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_)::value;
    const int buf = u & 1;
    read_w(buf, 0);
    read_w(buf, 1);
    read_x(buf, 0);
    if (!LAST) {
      stage(0, 0, u + 1);
      stage(1, 0, u + 1);
      stage(1, 1, u + 1);
      BG_VM(6);
    } else {
      BG_VM(0);
    }
    BG_BAR();
    BG_LGKM0();
    __builtin_amdgcn_sched_barrier(0);
    BG_PRIO(1);
    quad(0, 0);
    quad(0, 1);
    BG_PRIO(0);
    BG_BAR();
    read_x(buf, 1);
    if (!LAST) {
      stage(0, 1, u + 1);
      BG_VM(2);
    }
    BG_BAR();
    BG_LGKM0();
    __builtin_amdgcn_sched_barrier(0);
    BG_PRIO(1);
    quad(1, 1);
    quad(1, 0);
    BG_PRIO(0);
    BG_BAR();
  };
  int u = 0;
  if (PH == 4) {
#pragma nounroll
    for (; u < nkt - 2; ++u) {
      ktile(u, std::integral_constant<int, 0>{});
      abl_skip = true;
    }
    if (nkt >= 2) {
      ktile(u, std::integral_constant<int, 1>{});
      ++u;
    }
    ktile(u, std::integral_constant<int, 2>{});
  } else {
#pragma nounroll
    for (; u < nkt - 1; ++u) {
      ktile2(u, std::false_type{});
      abl_skip = true;
    }
    ktile2(u, std::true_type{});
  }
  if (wr == 0) BG_BAR();                                   // waves 0-3 meet the extra barrier of waves 4-7

  // ---- epilogue: bias, ReLU, conversion, stores.  Lane (r, q) of accumulator tile (mh, mi, nh, ni) holds batch row
  // m0 + 128 mh + 64 wr + 16 mi + r, features n0 + 128 nh + 32 wc + 16 ni + 4 q + {0..3}
  const bool vec = (N & 3) == 0;
  const float* bs = p.bias ? p.bias + (size_t)s * N : nullptr;
#ifndef BG_NO_WIDE_STORES
  if (p.y_bf16 && (N & 7) == 0 && !(reinterpret_cast<uintptr_t>(p.y) & 15) && !(BG_ABL & 8)) {
    // bf16 output, 16-byte stores (the store tail of this kernel is issue-bound: every block reaches its epilogue at once,
    // 32 eight-byte stores per lane).  Lanes (r, q) and (r, q ^ 1) hold neighbouring 4-feature groups of the SAME batch row in
    // each of the two feature tiles ni = 0, 1 of a W half; one v_permlane16_swap per packed dword hands the even row of 16 lanes
    // its partner's group of tile 0 and the odd row its partner's group of tile 1: every lane then owns 8 consecutive
    // features -- [own | partner's] of tile 0 in the even rows, [partner's | own] of tile 1 in the odd rows -- and stores
    // them with one instruction: 16 stores per lane instead of 32, the same bytes at the same addresses.
    const int qe = q & ~1;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      float bv[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + nh * 128 + wc * 32 + ni * 16 + q * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[ni][j] = 0.f;
        if (bs && n < N) {
          const float4 b4 = *reinterpret_cast<const float4*>(bs + n);
          bv[ni][0] = b4.x; bv[ni][1] = b4.y; bv[ni][2] = b4.z; bv[ni][3] = b4.w;
        }
      }
      const int n8 = n0 + nh * 128 + wc * 32 + (q & 1) * 16 + qe * 4;     // first of this lane's 8 features after the exchange
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          uint32_t pk[2][2];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = acc[mh][mi][nh][ni][j] + bv[ni][j];
              if (p.relu) v[j] = fmaxf(v[j], 0.f);
            }
            typedef __attribute__((ext_vector_type(2))) float f32x2_t;
            pk[ni][0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v[0], v[1]}, bf16x2_t));
            pk[ni][1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v[2], v[3]}, bf16x2_t));
          }
          const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          const int m = m0 + mh * 128 + wr * 64 + mi * 16 + r;
          if (m < M && n8 < N) {
            const size_t o = ((size_t)s * M + m) * (size_t)N + n8;
            *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(p.y) + o) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
        }
    }
    return;
  }
#endif
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + nh * 128 + wc * 32 + ni * 16 + q * 4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bs) {
        if (vec && n < N) {
          const float4 b4 = *reinterpret_cast<const float4*>(bs + n);
          bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[j] = (n + j < N) ? bs[n + j] : 0.f;
        }
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int m = m0 + mh * 128 + wr * 64 + mi * 16 + r;
          if (m >= M || n >= N) continue;
          if ((BG_ABL & 8) && acc[mh][mi][nh][ni][0] != 12345.678f) continue;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = acc[mh][mi][nh][ni][j] + bv[j];
            if (p.relu) v[j] = fmaxf(v[j], 0.f);
          }
          const size_t o = ((size_t)s * M + m) * (size_t)N + n;
          if (p.y_bf16) {
            __bf16* yp = reinterpret_cast<__bf16*>(p.y) + o;
            if (vec) {
              bf16x4 pk;
#pragma unroll
              for (int j = 0; j < 4; ++j) pk[j] = (__bf16)v[j];
              *reinterpret_cast<bf16x4*>(yp) = pk;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (n + j < N) yp[j] = (__bf16)v[j];
            }
          } else {
            float* yp = reinterpret_cast<float*>(p.y) + o;
            if (vec) {
              *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (n + j < N) yp[j] = v[j];
            }
          }
        }
    }
#undef BG_BAR
#undef BG_VM
#undef BG_MFMA
#undef BG_LGKM0
#undef BG_PRIO
}

// Host side.  Preconditions (checked by the caller, bnn_bbb_linear_fwd): K % 8 == 0, 16-byte aligned x / w, M * K and
// N * K bf16 elements of one sample below 2 GiB (32-bit buffer offsets).
inline hipError_t launch_block_gemm(BlockGemmK k, hipStream_t stream) {
  k.MT = (k.M + 255) / 256;
  k.NT = (k.N + 255) / 256;
  const long total = (long)k.S * k.MT * k.NT;
  const dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(kBgThreads);
  hipError_t err;
  if (k.K % 64) {
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(bbb_block_gemm_kernel<true, BG_PHASES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kBgLds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL((bbb_block_gemm_kernel<true, BG_PHASES>), grid, block, kBgLds, stream, k);
  } else {
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(bbb_block_gemm_kernel<false, BG_PHASES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kBgLds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL((bbb_block_gemm_kernel<false, BG_PHASES>), grid, block, kBgLds, stream, k);
  }
  return hipGetLastError();
}

}  // namespace bnn
