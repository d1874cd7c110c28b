#!/bin/bash
# The round's judged profiles in one GPU-box call: kernel-trace summaries of the headline, LR, 8-sample and training-step
# runs, then the FETCH_SIZE / WRITE_SIZE passes behind profiles/traffic.json (stamped with the kernel source hash).
# usage: tools/profile_final.sh <round tag>      (copy gpurun_out/profiles/* into profiles/ afterwards)
TAG=${1:-rXX}
P="bash tools/profile_round.sh $TAG"
$P stats bbb_g256 &&
$P stats lr_g256_l2 --variant lr &&
$P stats bbb_S8 --samples 8 --group 1 &&
$P statspy train_step tools/train_step_bench.py 2 graph &&
$P traffic bbb256 bbb_1200_n256_b128_bf16 bbb_fwd_gemm_kernel &&
$P traffic lr256 lr_1200_n256_b128_bf16 lr_fwd_gemm_kernel --variant lr &&
$P traffic bbbS8 bbb_1200_n8_b128_bf16 bbb_fwd_gemm --samples 8 --group 1
