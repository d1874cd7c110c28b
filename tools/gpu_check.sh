#!/bin/bash
# One GPU-box call of the development loop: the GPU test suite, then (if green) a bench run summarised in a few lines.
# usage: tools/gpu_check.sh <tag> [bench args...]   (outputs under gpurun_out/)
TAG=${1:-x}; shift
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/t_$TAG.log 2>&1; rc=$?
tail -12 gpurun_out/t_$TAG.log; echo pytest_rc=$rc
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py "$@" > gpurun_out/b_$TAG.json 2> gpurun_out/b_$TAG.err || { tail -20 gpurun_out/b_$TAG.err; exit 1; }
python tools/bench_summary.py gpurun_out/b_$TAG.json
