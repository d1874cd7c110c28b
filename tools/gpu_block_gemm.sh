#!/bin/bash
# K1g development runs: correctness over edge shapes, then timing (tools/block_gemm_bench.hip); variants = binaries
O=gpurun_out/block_gemm.log
: > $O
for B in "$@"; do
  echo "== $B" >> $O
  for a in "1 256 256 64 2 0" "2 300 520 192 2 0" "2 512 1200 784 5 1 1" "1 100 40 8 2 0" "4 1024 4096 4096 20 1" "4 4096 4096 4096 10 1" "1 4096 4096 4096 20 1"; do
    timeout -k 10 120 $B $a >> $O 2>&1 || echo "FAILED: $a (rc $?)" >> $O
  done
done
grep -v "^check: 0/" $O
