#!/bin/bash
# timing only: usage tools/gpu_bg_time.sh "<args>" bin...
A=$1; shift
for B in "$@"; do echo "== $B"; timeout -k 10 120 $B $A | tail -1; done
