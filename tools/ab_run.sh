#!/bin/bash
# Runs on the GPU box: the same lane_sweep arguments against several builds of the library (synthesis_amd/ab/lib<name>.so; "HEAD" =
# the in-tree library), interleaved and repeated so that box and thermal state average out.   usage: tools/ab_run.sh "<names>" <reps> <lane_sweep args...>
NAMES=$1; REPS=$2; shift 2
export SYN_DEBUG=1
for r in $(seq 1 $REPS); do
  for n in $NAMES; do
    if [ "$n" = "HEAD" ]; then unset SYNTHESIS_AMD_LIB; else export SYNTHESIS_AMD_LIB=$PWD/synthesis_amd/ab/lib$n.so; fi
    python3 tools/lane_sweep.py "$@" 2>&1 | grep games/s | sed "s/^/[$n] /"
  done
done
