#!/bin/bash
# Runs on the GPU box (via gpurun): the SQ / TA / L2 counter passes of tools/collect_profiles.sh on the PolicyWithCache leg
# (`bench.py --only-policy-cache`) — what bounds the kernel when two thirds of the matrix work disappears.
#   usage: tools/collect_cache_sq.sh r02      -> gpurun_out/profiles_<tag>/<tag>_pmc_c{sq1,sq2,ta,l2}.csv
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CACHE="bench.py --only-policy-cache"
run() {
  local name=$1; shift
  rm -rf $R/gpurun_out/prof_$name; mkdir -p $R/gpurun_out/prof_$name
  timeout -k 5 600 rocprofv3 "$@" > $OUT/log_$name.txt 2>&1
}
run csq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/prof_csq1 -- python3 $CACHE
run csq2 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/prof_csq2 -- python3 $CACHE
run cta --pmc TA_TA_BUSY_sum TA_BUSY_avr --output-format csv -d $R/gpurun_out/prof_cta -- python3 $CACHE
run cl2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/prof_cl2 -- python3 $CACHE
for c in csq1 csq2 cta cl2; do
  f=$(ls -t $R/gpurun_out/prof_$c/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && continue
  head -1 $f > $OUT/${TAG}_pmc_$c.csv
  grep selfplay_kernel $f >> $OUT/${TAG}_pmc_$c.csv
done
ls -la $OUT | grep "_pmc_c"
