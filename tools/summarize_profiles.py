#!/usr/bin/env python3
"""Distils the rocprofv3 outputs of tools/collect_profiles.sh into <tag>_pmc.json (HBM traffic per launch of the headline
kernel and of the PolicyWithCache leg, guide-corrected) and <tag>_sq.json (SQ / TA / L2 counters per launch)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

out, tag = sys.argv[1], sys.argv[2]


def rows(name):
    path = f"{out}/{tag}_{name}.csv"
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def per_launch(name, pick):
    """counter -> value per launch (summed over the dispatch's counter rows, averaged over the picked dispatches)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows(name):
        if pick(r["Kernel_Name"]):
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    res = collections.defaultdict(list)
    for d in acc.values():
        for k, v in d.items():
            res[k].append(v)
    return {k: sum(v) / len(v) for k, v in res.items()}, len(acc)


timed = lambda n: "selfplay_kernel" in n and "<0, false" in n   # the un-instrumented (COUNT = false) self-play launches
line = None
for l in open(f"{out}/{tag}_bench_lines_under_profiler.jsonl"):
    line = json.loads(l)
cfgv = [line["config"]["concurrent_games_per_gpu"], line["config"]["games_per_step_per_gpu"], line["config"]["explores_per_move"]] if line else None
f, nf = per_launch("pmc_fetch", timed)
w, nw = per_launch("pmc_write", timed)
cf, _ = per_launch("pmc_cfetch", timed)
cw, _ = per_launch("pmc_cwrite", timed)
ks = [r for r in csv.DictReader(open(f"{out}/{tag}_kernel_stats.csv")) if timed(r["Name"])]
ks.sort(key=lambda r: -float(r["TotalDurationNs"]))
summary = {
    "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-4096 --no-policy-cache --no-extras",
    "bench_config": cfgv, "csrc_sha16": kernel_source_hash(),
    "kernel": ks[0]["Name"] if ks else None, "kernel_calls": int(ks[0]["Calls"]) if ks else None,
    "kernel_avg_ms": float(ks[0]["AverageNs"]) / 1e6 if ks else None,
    "FETCH_SIZE_kb_per_launch": f.get("FETCH_SIZE"), "WRITE_SIZE_kb_per_launch": w.get("WRITE_SIZE"),
    # MI355X guide §HBM: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) loads -> doubled; WRITE_SIZE exact
    "traffic_bytes_per_launch": (2.0 * f.get("FETCH_SIZE", 0) + w.get("WRITE_SIZE", 0)) * 1024.0 if f and w else None,
    "cache_command": "python3 bench.py --only-policy-cache",
    "cache_FETCH_SIZE_kb_per_launch": cf.get("FETCH_SIZE"), "cache_WRITE_SIZE_kb_per_launch": cw.get("WRITE_SIZE"),
    # the cache leg launches twice at full size (timed + counted): the average over its launches of >= 1 M games
    "cache_traffic_bytes_per_launch": (2.0 * cf.get("FETCH_SIZE", 0) + cw.get("WRITE_SIZE", 0)) * 1024.0 if cf and cw else None,
}
json.dump(summary, open(f"{out}/{tag}_pmc.json", "w"), indent=1)
sq = {}
for name in ("pmc_sq1", "pmc_sq2", "pmc_ta", "pmc_l2", "pmc_lat"):
    v, n = per_launch(name, timed)
    sq.update(v)
if sq:
    wave = sq.get("SQ_WAVE_CYCLES")
    d = {"counters_per_launch": sq, "csrc_sha16": kernel_source_hash(), "bench_config": cfgv}
    if wave and ks:
        waves_per_simd = 4 if "1024" in (ks[0]["Name"] or "") or True else 3
        simd_cycles = wave * 4.0 / waves_per_simd       # SQ_WAVE_CYCLES counts 4-cycle units per resident wave
        d["derived"] = {
            "note": "fractions of SIMD time (1024 SIMDs); SQ_WAVE_CYCLES x 4 / resident waves per SIMD = SIMD cycles",
            "mfma_pipe_busy": sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles,
            "valu_issue_non_mfma": (sq.get("SQ_INSTS_VALU", 0) - sq.get("SQ_INSTS_MFMA", 0)) * 4.0 / simd_cycles,
            "l2_hit_rate": sq.get("TCC_HIT_sum", 0) / max(1.0, sq.get("TCC_REQ_sum", 1.0)),
            "l2_read_latency_cycles": sq.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(1.0, sq.get("TCP_TCC_READ_REQ_sum", 1.0)),
        }
    json.dump(d, open(f"{out}/{tag}_sq.json", "w"), indent=1)
print(json.dumps(summary))
