#!/bin/bash
# Development binaries behind profiles/rNN_block_gemm_ablation.log, rNN_mfma_ceiling.log and rNN_ubench_generator.log (built in the
# container, they travel to the GPU box with the snapshot; *.out is git-ignored).
set -e
cd "$(dirname "$0")/.."
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value"
for a in 0 1 2 3 7 8 15 27 31; do $H -DBG_ABL=$a -DBG_PHASES=2 tools/block_gemm_bench.hip -o tools/bg_ph2_abl$a.out; done
$H -DBG_PHASES=4 tools/block_gemm_bench.hip -o tools/block_gemm_bench_ph4.out
$H -DBG_PHASES=2 tools/block_gemm_bench.hip -o tools/block_gemm_bench.out
$H tools/mfma_ceiling.hip -o tools/mfma_ceiling.out
$H tools/ubench.hip -o tools/ubench.out
