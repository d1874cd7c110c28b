#!/bin/bash
# One GPU-box call for the training rows: the training tests, then the graph step timing (BBB + LR, bf16 + fp32).
# usage: tools/gpu_train_check.sh <tag>
TAG=${1:-x}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_training.py -m gpu -x -q > gpurun_out/tt_$TAG.log 2>&1; rc=$?
tail -12 gpurun_out/tt_$TAG.log; echo pytest_rc=$rc
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/train_step_bench.py 2 graph > gpurun_out/ts_$TAG.log 2>&1 || { tail -20 gpurun_out/ts_$TAG.log; exit 1; }
grep "ms/step" gpurun_out/ts_$TAG.log
