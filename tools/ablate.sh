#!/bin/bash
# Runs on the GPU box: additive sensitivity runs of the lane-per-tree kernel (lane_kernel.cuh: SYN_ABLATE). Each run executes one
# component twice; its slowdown against mask 0 is that component's marginal cost in the contended kernel.
#   usage: tools/ablate.sh <out file> <lane_sweep args...>
OUT=$1; shift
export SYN_DEBUG=1 SYN_PROFILE=1
: > $OUT
for m in 128 1 2 4 8 16 32 64 128 0; do   # 128 = nothing duplicated, stamps off: the reference of the other runs
  echo "== SYN_ABLATE=$m" >> $OUT
  SYN_ABLATE=$m python3 tools/lane_sweep.py "$@" 2>&1 | grep -E "games/s|cycles per round: A" | grep -v "grid=1 " >> $OUT
done
# the stamps themselves (no ablation): what the profile build costs against the production kernel
echo "== production kernel" >> $OUT
env -u SYN_PROFILE python3 tools/lane_sweep.py "$@" 2>&1 | grep -E "games/s" >> $OUT
