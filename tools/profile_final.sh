#!/bin/bash
# The round's judged profiles in one GPU-box call: kernel-trace summaries of the driver's command (--steps 20 --warmup 5),
# of the isolated dominant kernels (headline K1b, LR K3b, 8-sample K-sliced form, the block GEMM K1g at batch 4096) and of
# the training step, the one-evaluation LR layer (K3s); the FETCH_SIZE / WRITE_SIZE passes behind profiles/traffic.json and the SQ / GRBM passes behind
# profiles/pmc.json (both stamped with the hash of the kernel sources they were measured on; pmc.json is rebuilt from
# scratch so that it holds current kernels only).
# usage: tools/profile_final.sh <round tag>      (copy gpurun_out/profiles/* into profiles/ afterwards)
TAG=${1:-rXX}
P="bash tools/profile_round.sh $TAG"
PMC="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
mkdir -p gpurun_out/profiles && echo '{}' > gpurun_out/profiles/pmc.json
$P stats bbb_g256 &&
$P stats bbb_g256_x3 --math bf16x3 &&
$P stats lr_g256_l2 --variant lr &&
$P stats lr_g256_l2_x3 --variant lr --math bf16x3 &&
$P stats bbb_S8 --samples 8 --group 1 &&
$P stats lr_S1 --variant lr --samples 1 --group 1 &&
$P stats wide_B4096_S4 --net wide --batch 4096 --samples 4 --group 1 &&
$P stats wide_B1024_S4 --net wide --batch 1024 --samples 4 --group 1 &&
$P statspy train_step tools/train_step_bench.py 2 graph &&
$P traffic bbb256 bbb_1200_n256_b128_bf16 bbb_fwd_gemm &&
$P traffic bbb256x3 bbb_1200_n256_b128_bf16x3 bbb_fwd_gemm --math bf16x3 &&
$P traffic lr256 lr_1200_n256_b128_bf16 lr_fwd_gemm_kernel --variant lr &&
$P traffic bbbS8 bbb_1200_n8_b128_bf16 bbb_fwd_gemm --samples 8 --group 1 &&
$P traffic lrS1 lr_1200_n1_b128_bf16 lr_fwd_kslice_kernel --variant lr --samples 1 --group 1 &&
$P traffic wide4096 block_gemm_4096_n4_b4096_bf16 bbb_block_gemm_kernel --net wide --batch 4096 --samples 4 --group 1 &&
$P pmc bbb256 bbb_g256 "$PMC" &&
$P pmc bbb256x3 bbb_g256_x3 "$PMC" --math bf16x3 &&
$P pmc lr256 lr_g256 "$PMC" --variant lr &&
$P pmc wide4096 wide_B4096 "$PMC" --net wide --batch 4096 --samples 4 --group 1 &&
$P statsfull steps20 --steps 20 --warmup 5 --no-extras --no-cpu-baseline    # last: its bench line carries the traffic / counters collected above
# harness logs behind DESIGN.md 4 (development binaries are built in the container and travel with the snapshot)
PROF=gpurun_out/profiles
( tools/ubench.out > $PROF/${TAG}_ubench_generator.log 2>&1 ) || true
( for pz in 1 0; do echo "== BNN_TUNE_PAIRS=$pz (1: K1b2, 0: K1b)"; BNN_TUNE_PAIRS=$pz BNN_HIP_LIB=$PWD/bayesian-neural-network_amd/bnn_hip/libbnn_hip_tune.so python3 tools/k1b_ablate.py 2>&1 | grep -v amdgpu.ids; done > $PROF/${TAG}_k1b_ablate.log 2>&1 ) || true
( python3 tools/make_valu_floor.py $PROF/${TAG}_k1b_ablate.log > /dev/null 2>&1 && cp profiles/valu_floor.json $PROF/valu_floor.json ) || true
( python3 tools/block_gemm_vs_library.py 2>&1 | grep -v amdgpu.ids > $PROF/${TAG}_block_gemm_vs_library.log ) || true
( python3 tools/single_eval_bench.py 2>&1 | grep -v amdgpu.ids > $PROF/${TAG}_single_eval.log ) || true
( python3 tools/predict_bench.py 10 2>&1 | grep "S=10" > $PROF/${TAG}_predict.log ) || true
( bash tools/gpu_k1b2_floor.sh > /dev/null 2>&1 && cp gpurun_out/k1b2_floor/summary.txt $PROF/${TAG}_k1b2_floor_pmc.log ) || true
( python3 tools/stamps_k1b2.py 2>&1 | grep -v amdgpu.ids > $PROF/${TAG}_k1b2_stamps.log; python3 tools/stamps_k1b2.py bf16x3 2>&1 | grep -v amdgpu.ids >> $PROF/${TAG}_k1b2_stamps.log ) || true
( tools/ldsdma_probe.out 2 > $PROF/${TAG}_ldsdma_probe.log 2>&1; tools/ldsdma_probe.out 64 >> $PROF/${TAG}_ldsdma_probe.log 2>&1 ) || true
