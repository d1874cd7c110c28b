#!/bin/bash
# PMC passes over tools/k1b2_floor_probe.py (tuning build): what a k-step of the headline kernel and of its vector-only skeleton
# really cost.  usage: tools/gpu_k1b2_floor.sh   -> gpurun_out/k1b2_floor/summary.txt
ROOT=$(pwd); export TMPDIR=/tmp
export BNN_HIP_LIB=$ROOT/bayesian-neural-network_amd/bnn_hip/libbnn_hip_tune.so
OUT=$ROOT/gpurun_out/k1b2_floor; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout -k 10 200 python3 $ROOT/tools/k1b2_floor_probe.py > $OUT/plain.log 2> $OUT/plain.err) || exit 1
(cd /tmp && timeout -k 10 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/a -o a -- python3 $ROOT/tools/k1b2_floor_probe.py > $OUT/a.log 2> $OUT/a.err) || exit 1
cat $OUT/plain.log > $OUT/summary.txt
python3 $ROOT/tools/k1b2_floor_summary.py $OUT/a $OUT/a.log >> $OUT/summary.txt
cat $OUT/summary.txt
