#!/bin/bash
# The shader clock (GRBM_GUI_ACTIVE / 8 / duration) and the VALU occupancy of the headline launch for build variants of K1b2
# (tools/build_k1b2_variants.sh): does a variant that overlaps more of the step run at a lower clock?
# usage: tools/gpu_k1b2_clock.sh <variant tag ...>   (libbnn_hip_<tag>.so; "product" = the product library)
ROOT=$(pwd); export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/k1b2_clock; rm -rf $OUT; mkdir -p $OUT
for v in product "$@"; do
  if [ "$v" = product ]; then unset BNN_HIP_LIB; else export BNN_HIP_LIB=$ROOT/bayesian-neural-network_amd/bnn_hip/libbnn_hip_$v.so; fi
  (cd /tmp && PROBE_REPS=10 timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/$v -o a -- python3 $ROOT/tools/k1b2_floor_probe.py > $OUT/$v.log 2> $OUT/$v.err) || exit 1
  echo "== $v" >> $OUT/summary.txt
  python3 $ROOT/tools/k1b2_floor_summary.py $OUT/$v $OUT/$v.log >> $OUT/summary.txt
done
cat $OUT/summary.txt
