#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of the bench command + the PMC passes the MI355X guide prescribes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, they need 3 + 2) on the same launches, the same two passes on
# the PolicyWithCache / reference-configuration legs (bench.py --only-policy-cache) and the SQ / TA / L2 counter passes the design
# discussion quotes; distilled into gpurun_out/profiles_<tag>/ for copying into profiles/ (tracked).
#   usage: tools/collect_profiles.sh r03 [fast]      ("fast": skip the SQ / TA / L2 passes)
set -u
TAG=${1:-r05}
MODE=${2:-full}     # full | fast (no SQ / TA / L2 passes) | trace (only the kernel trace: refreshes <tag>_kernel_stats.csv)
R=$PWD
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the kernel trace runs the full default line (with its roofline objects); the counter passes only need the launches
TRACE="bench.py --steps 2 --warmup 1 --full-warmup --no-cpu-baseline --no-4096 --no-policy-cache --no-extras --no-learner-loop"
BENCH="bench.py --steps 1 --warmup 1 --skip-counted --no-learner-loop"
CACHE="bench.py --only-extra-legs"
run() {  # name, rocprof args..., -- program
  local name=$1; shift
  rm -rf $R/gpurun_out/prof_$name; mkdir -p $R/gpurun_out/prof_$name
  timeout -k 5 900 rocprofv3 "$@" > $OUT/log_$name.txt 2>&1
}
run trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $TRACE
if [ "$MODE" != "trace" ]; then
run fetch --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $BENCH
run write --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $BENCH
run cfetch --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_cfetch -- python3 $CACHE
run cwrite --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_cwrite -- python3 $CACHE
fi
if [ "$MODE" = "full" ]; then
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/prof_sq1 -- python3 $BENCH
run sq2 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/prof_sq2 -- python3 $BENCH
run ta --pmc TA_TA_BUSY_sum TA_BUSY_avr --output-format csv -d $R/gpurun_out/prof_ta -- python3 $BENCH
run l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/prof_l2 -- python3 $BENCH
run csq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/prof_csq1 -- python3 $CACHE
run csq2 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/prof_csq2 -- python3 $CACHE
run cl2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/prof_cl2 -- python3 $CACHE
fi
cp $(ls -t $R/gpurun_out/prof_trace/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
head -1 $(ls -t $R/gpurun_out/prof_trace/*/*_kernel_trace.csv | head -1) > $OUT/${TAG}_kernel_trace_selfplay.csv
grep selfplay_kernel $(ls -t $R/gpurun_out/prof_trace/*/*_kernel_trace.csv | head -1) >> $OUT/${TAG}_kernel_trace_selfplay.csv
for c in fetch write cfetch cwrite sq1 sq2 ta l2 csq1 csq2 cl2; do
  f=$(ls -t $R/gpurun_out/prof_$c/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && continue
  head -1 $f > $OUT/${TAG}_pmc_$c.csv
  grep selfplay_kernel $f >> $OUT/${TAG}_pmc_$c.csv
done
if [ "$MODE" != "trace" ]; then
grep -h '"metric"' $OUT/log_trace.txt $OUT/log_fetch.txt $OUT/log_write.txt > $OUT/${TAG}_bench_lines_under_profiler.jsonl
grep -h 'with_policy_cache' $OUT/log_cfetch.txt $OUT/log_cwrite.txt > $OUT/${TAG}_cache_lines_under_profiler.jsonl
fi
python3 tools/summarize_profiles.py "$OUT" "$TAG"
ls -la $OUT
