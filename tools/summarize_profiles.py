#!/usr/bin/env python3
"""Distils the rocprofv3 outputs of tools/collect_profiles.sh into <tag>_pmc.json (HBM traffic per launch of the headline kernel,
of the PolicyWithCache leg and of the reference-configuration leg, guide-corrected) and <tag>_sq.json (SQ / TA / L2 counters per
launch of the headline kernel and of the PolicyWithCache leg)."""
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


def per_launch(name, pick, largest=True):
    """counter -> value of ONE launch: the counter rows of a dispatch summed, and among the dispatches `pick` accepts the one with
    the largest sums (the timed launch: the warm-up launches of the same kernel play fewer games) — or their average."""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows(name):
        if pick(r["Kernel_Name"]):
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if not acc:
        return {}, 0
    if largest:
        best = max(acc.values(), key=lambda d: sum(d.values()))
        return dict(best), len(acc)
    res = collections.defaultdict(list)
    for d in acc.values():
        for k, v in d.items():
            res[k].append(v)
    return {k: sum(v) / len(v) for k, v in res.items()}, len(acc)


def nth_timed_launch(name, pick, nth):
    """counter -> value of the nth (in dispatch order) TIMED launch among the dispatches `pick` accepts: a leg's timed launch plays four
    times the games of its warm-up launch, so the timed ones are the dispatches whose counters reach half of the largest one's."""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows(name):
        if pick(r["Kernel_Name"]):
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    if not acc:
        return {}
    top = max(sum(d.values()) for d in acc.values())
    big = [dict(acc[k]) for k in sorted(acc) if sum(acc[k].values()) >= 0.5 * top]
    return big[nth] if nth < len(big) else {}


# the un-instrumented (COUNT = false) self-play launches: the parity configuration takes the FAST instantiation, the reference's
# Fpu::Func configuration the general one (template arguments <MODE, COUNT, FAST, ...>)
import re  # noqa: E402
timed = lambda n: "selfplay_kernel" in n and re.search(r"<0, false, (true|1),", n) is not None
# the extra legs of `bench.py --only-extra-legs`, told apart by their instantiation <MODE, COUNT, FAST, waves, PROF, POLICY>: PolicyWithCache
# = the parity family on 12 waves, the trained network = the parity family on 16 waves, the conv network = POLICY 2, the reference's
# Fpu::Func configuration = family 2 (the runtime-switched instantiation before round 4)
# (round 5: Connect4Net's parity family runs 12 waves in every leg, so the PolicyWithCache leg and the trained-network leg are the FIRST and
#  the SECOND timed launch of that instantiation in the command's order; POLICY 3 = the f16x2 arithmetic: random-init, then trained)
leg_cache = lambda n: "selfplay_kernel_lanes" in n and re.search(r"<0, false, (true|1), 12, (false|0), 0>", n) is not None
leg_trained = leg_cache
leg_f16 = lambda n: "selfplay_kernel_lanes" in n and re.search(r"<0, false, (true|1), 12, (false|0), 3>", n) is not None
leg_conv = lambda n: "selfplay_kernel_lanes" in n and re.search(r"<0, false, (true|1), 16, (false|0), 2>", n) is not None
# (POLICY 0 only: the reference configuration's f16x2 variant — family 2, POLICY 3 — plays the same number of games right behind it)
timed_general = lambda n: "selfplay_kernel" in n and re.search(r"<0, false, (false|0|2), \d+, (false|0), 0>", n) is not None
line = None
for l in open(f"{out}/{tag}_bench_lines_under_profiler.jsonl"):
    line = json.loads(l)
cfgv = [line["config"]["concurrent_games_per_gpu"], line["config"]["games_per_step_per_gpu"], line["config"]["explores_per_move"]] if line else None
cache_line = None
if os.path.exists(f"{out}/{tag}_cache_lines_under_profiler.jsonl"):
    for l in open(f"{out}/{tag}_cache_lines_under_profiler.jsonl"):
        cache_line = json.loads(l)
f, nf = per_launch("pmc_fetch", timed)
w, nw = per_launch("pmc_write", timed)
cf = nth_timed_launch("pmc_cfetch", leg_cache, 0)
cw = nth_timed_launch("pmc_cwrite", leg_cache, 0)
tf = nth_timed_launch("pmc_cfetch", leg_trained, 1)
tw = nth_timed_launch("pmc_cwrite", leg_trained, 1)
hf, hw = nth_timed_launch("pmc_cfetch", leg_f16, 0), nth_timed_launch("pmc_cwrite", leg_f16, 0)
htf, htw = nth_timed_launch("pmc_cfetch", leg_f16, 1), nth_timed_launch("pmc_cwrite", leg_f16, 1)
vf, _ = per_launch("pmc_cfetch", leg_conv)
vw, _ = per_launch("pmc_cwrite", leg_conv)
rf, _ = per_launch("pmc_cfetch", timed_general)
rw, _ = per_launch("pmc_cwrite", timed_general)
ks = [r for r in csv.DictReader(open(f"{out}/{tag}_kernel_stats.csv")) if timed(r["Name"])]
ks.sort(key=lambda r: -float(r["TotalDurationNs"]))


def traffic(fe, wr):
    # MI355X guide §HBM: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) loads -> doubled; WRITE_SIZE exact; KB units
    return (2.0 * fe.get("FETCH_SIZE", 0) + wr.get("WRITE_SIZE", 0)) * 1024.0 if fe and wr else None


summary = {
    "command": "python3 bench.py --steps 1 --warmup 1 --skip-counted --no-learner-loop   (the timed launch of each pass)",
    "trace_command": "python3 bench.py --steps 2 --warmup 1 --full-warmup --no-cpu-baseline --no-4096 --no-policy-cache --no-extras --no-learner-loop",
    "bench_config": cfgv, "csrc_sha16": kernel_source_hash(),
    "kernel": ks[0]["Name"] if ks else None, "kernel_calls": int(ks[0]["Calls"]) if ks else None,
    "kernel_avg_ms": float(ks[0]["AverageNs"]) / 1e6 if ks else None,
    "kernel_max_ms": float(ks[0]["MaxNs"]) / 1e6 if ks and "MaxNs" in ks[0] else None,
    "FETCH_SIZE_kb_per_launch": f.get("FETCH_SIZE"), "WRITE_SIZE_kb_per_launch": w.get("WRITE_SIZE"),
    "traffic_bytes_per_launch": traffic(f, w),
    "cache_command": "python3 bench.py --only-extra-legs   (four legs; the largest dispatch of each kernel instantiation)",
    "cache_FETCH_SIZE_kb_per_launch": cf.get("FETCH_SIZE"), "cache_WRITE_SIZE_kb_per_launch": cw.get("WRITE_SIZE"),
    "cache_traffic_bytes_per_launch": traffic(cf, cw),
    "reference_FETCH_SIZE_kb_per_launch": rf.get("FETCH_SIZE"), "reference_WRITE_SIZE_kb_per_launch": rw.get("WRITE_SIZE"),
    "reference_traffic_bytes_per_launch": traffic(rf, rw),
    "trained_FETCH_SIZE_kb_per_launch": tf.get("FETCH_SIZE"), "trained_WRITE_SIZE_kb_per_launch": tw.get("WRITE_SIZE"),
    "trained_traffic_bytes_per_launch": traffic(tf, tw),
    "conv_FETCH_SIZE_kb_per_launch": vf.get("FETCH_SIZE"), "conv_WRITE_SIZE_kb_per_launch": vw.get("WRITE_SIZE"),
    "conv_traffic_bytes_per_launch": traffic(vf, vw),
    "f16x2_FETCH_SIZE_kb_per_launch": hf.get("FETCH_SIZE"), "f16x2_WRITE_SIZE_kb_per_launch": hw.get("WRITE_SIZE"),
    "f16x2_traffic_bytes_per_launch": traffic(hf, hw),
    "f16x2_trained_FETCH_SIZE_kb_per_launch": htf.get("FETCH_SIZE"), "f16x2_trained_WRITE_SIZE_kb_per_launch": htw.get("WRITE_SIZE"),
    "f16x2_trained_traffic_bytes_per_launch": traffic(htf, htw),
    "extra_leg_games": {
        "cache_traffic_bytes_per_launch": (cache_line or {}).get("with_policy_cache", {}).get("games"),
        "reference_traffic_bytes_per_launch": (cache_line or {}).get("reference_selfplay_config", {}).get("games"),
        "trained_traffic_bytes_per_launch": (cache_line or {}).get("with_trained_weights", {}).get("games"),
        "conv_traffic_bytes_per_launch": (cache_line or {}).get("with_conv_policy", {}).get("games"),
        "f16x2_traffic_bytes_per_launch": (cache_line or {}).get("with_f16x2_network", {}).get("games"),
        "f16x2_trained_traffic_bytes_per_launch": (cache_line or {}).get("with_f16x2_network", {}).get("trained_checkpoint", {}).get("games"),
    },
}
json.dump(summary, open(f"{out}/{tag}_pmc.json", "w"), indent=1)


def derive(sq, waves_per_simd):
    wave = sq.get("SQ_WAVE_CYCLES")
    if not wave:
        return None
    simd_cycles = wave * 4.0 / waves_per_simd       # SQ_WAVE_CYCLES counts 4-cycle units per resident wave
    return {
        "note": "fractions of SIMD time (1024 SIMDs); SQ_WAVE_CYCLES x 4 / resident waves per SIMD = SIMD cycles",
        "mfma_pipe_busy": sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles,
        "valu_issue_non_mfma_if_INSTS_VALU_includes_mfma": (sq.get("SQ_INSTS_VALU", 0) - sq.get("SQ_INSTS_MFMA", 0)) * 4.0 / simd_cycles,
        "valu_issue_if_INSTS_VALU_excludes_mfma": sq.get("SQ_INSTS_VALU", 0) * 4.0 / simd_cycles,
        "l2_hit_rate": sq.get("TCC_HIT_sum", 0) / max(1.0, sq.get("TCC_REQ_sum", 1.0)),
        "valu_mfma_coexec_share_of_simd_cycles": sq.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / simd_cycles if "SQ_VALU_MFMA_COEXEC_CYCLES" in sq else None,
        "wait_any_share_of_wave_cycles": sq.get("SQ_WAIT_ANY", 0) / wave if "SQ_WAIT_ANY" in sq else None,
        "wait_inst_any_share_of_wave_cycles": sq.get("SQ_WAIT_INST_ANY", 0) / wave,
        "active_inst_any_share_of_wave_cycles": sq.get("SQ_ACTIVE_INST_ANY", 0) / wave,
    }


d = {"csrc_sha16": kernel_source_hash(), "bench_config": cfgv}
sq = {}
for name in ("pmc_sq1", "pmc_sq2", "pmc_ta", "pmc_l2"):
    sq.update(per_launch(name, timed)[0])
if sq:
    d["counters_per_launch"] = sq
    d["derived"] = derive(sq, 3)   # (round 5: 12 waves per CU = 3 per SIMD)
csq, hsq = {}, {}
for name in ("pmc_csq1", "pmc_csq2", "pmc_cl2"):
    csq.update(nth_timed_launch(name, leg_cache, 0))
    hsq.update(nth_timed_launch(name, leg_f16, 0))
if hsq:
    d["f16x2_leg"] = {"command": "python3 bench.py --only-extra-legs (the first timed launch of the 12-wave POLICY-3 instantiation: random-init network)",
                      "counters_per_launch": hsq, "derived": derive(hsq, 3)}
if csq:
    d["policy_cache_leg"] = {"command": "python3 bench.py --only-extra-legs (the largest 12-wave parity-family dispatch of each pass)",
                             "counters_per_launch": csq, "derived": derive(csq, 3)}
if sq or csq:
    json.dump(d, open(f"{out}/{tag}_sq.json", "w"), indent=1)
print(json.dumps(summary))
