#!/bin/bash
# K1b2 build variants (pairs per block x LDS staging buffers) as separate libraries for A/B runs through BNN_HIP_LIB:
#   tools/build_k1b2_variants.sh "2 2" "4 3" ...  ->  bnn_hip/libbnn_hip_p<P>r<R>.so
set -e
cd "$(dirname "$0")/../bayesian-neural-network_amd/csrc"
make -j4 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-function -ffp-contract=fast"
for v in "$@"; do
  set -- $v
  /opt/rocm/bin/hipcc $FLAGS -DBNN_GEMM_PAIRS=$1 -DBNN_GEMM_RING=$2 ${EXTRA_DEFS} -c bbb_linear.hip -o /tmp/bbb_linear_p$1r$2.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../bnn_hip/libbnn_hip_p$1r$2.so /tmp/bbb_linear_p$1r$2.o bbb_sample.o bbb_bwd.o lr_linear.o lr_bwd.o reduce.o optim.o posthoc.o
  echo built p$1r$2
done
