#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + the two PMC passes the MI355X guide prescribes (FETCH_SIZE
# and WRITE_SIZE cannot share a pass: TCC has 4 slots, they need 3 + 2), all on the SAME bench.py command, and distils
# them into gpurun_out/profiles_<tag>/ for copying into profiles/ (tracked).
#   usage: tools/collect_profiles.sh r01
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT $R/gpurun_out/prof_trace $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
export TMPDIR=/tmp
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-4096 --no-policy-cache"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $BENCH > $OUT/bench_under_trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $BENCH > $OUT/bench_under_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $BENCH > $OUT/bench_under_write.log 2>&1
cp $(ls -t $R/gpurun_out/prof_trace/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
head -1 $(ls -t $R/gpurun_out/prof_trace/*/*_kernel_trace.csv | head -1) > $OUT/${TAG}_kernel_trace_selfplay.csv
grep selfplay_kernel $(ls -t $R/gpurun_out/prof_trace/*/*_kernel_trace.csv | head -1) >> $OUT/${TAG}_kernel_trace_selfplay.csv
for c in fetch write; do
  f=$(ls -t $R/gpurun_out/prof_$c/*/*_counter_collection.csv | head -1)
  head -1 $f > $OUT/${TAG}_pmc_$c.csv
  grep selfplay_kernel $f >> $OUT/${TAG}_pmc_$c.csv
done
grep -h '"metric"' $OUT/bench_under_*.log > $OUT/${TAG}_bench_lines_under_profiler.jsonl
python3 - "$OUT" "$TAG" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
def rows(path):
    return list(csv.DictReader(open(path)))
def timed(rs):  # the un-instrumented (COUNT = false) launches of the headline kernel (timed steps + warm-up)
    return [r for r in rs if "selfplay_kernel_lanes<0, false" in r["Kernel_Name"]]
f = timed(rows(f"{out}/{tag}_pmc_fetch.csv")); w = timed(rows(f"{out}/{tag}_pmc_write.csv"))
fetch_kb = sum(float(r["Counter_Value"]) for r in f) / max(1, len(f))
write_kb = sum(float(r["Counter_Value"]) for r in w) / max(1, len(w))
ks = [r for r in rows(f"{out}/{tag}_kernel_stats.csv") if "selfplay_kernel_lanes<0, false" in r["Name"]]
summary = {
    "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-4096 --no-policy-cache",
    "kernel": ks[0]["Name"] if ks else None,
    "kernel_calls": int(ks[0]["Calls"]) if ks else None,
    "kernel_avg_ms": float(ks[0]["AverageNs"]) / 1e6 if ks else None,
    "FETCH_SIZE_kb_per_launch": fetch_kb, "WRITE_SIZE_kb_per_launch": write_kb,
    # MI355X guide §HBM: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) loads -> doubled; WRITE_SIZE exact
    "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "traffic_bytes_per_launch_uncorrected": (fetch_kb + write_kb) * 1024.0,
}
json.dump(summary, open(f"{out}/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(summary))
PY
ls -la $OUT
