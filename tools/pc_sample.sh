#!/bin/bash
# Runs on the GPU box (via gpurun): PC sampling of one fused self-play launch (rocprofv3 beta feature; ATT needs a decoder
# library this image does not ship). Tries the hardware (stochastic) method first, the host-trap method second. The raw sample
# file is large, so it is reduced here to counts per (code object offset / instruction / stall reason) and only the reduction plus a head
# of the raw file travel back in gpurun_out/pcs_<tag>/.
#   usage: tools/pc_sample.sh <tag> <lane_sweep args...>
set -u
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/pcs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS=("$@")
try() {  # name, extra rocprof args
  local name=$1; shift
  rm -rf /tmp/pcs_$name; mkdir -p /tmp/pcs_$name
  timeout -k 5 600 rocprofv3 --pc-sampling-beta-enabled "$@" --kernel-trace --output-format csv -d /tmp/pcs_$name -- python3 tools/lane_sweep.py "${ARGS[@]}" > $OUT/log_$name.txt 2>&1
  echo "rc=$?" >> $OUT/log_$name.txt
  find /tmp/pcs_$name -type f | xargs ls -la >> $OUT/log_$name.txt 2>&1
}
try stoch --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576
if ! find /tmp/pcs_stoch -name '*pc_sampling*' | grep -q .; then
  try trap --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100
fi
python3 tools/pc_reduce.py /tmp/pcs_stoch /tmp/pcs_trap $OUT > $OUT/reduce_log.txt 2>&1
ls -la $OUT
