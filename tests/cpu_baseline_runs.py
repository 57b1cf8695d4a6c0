#!/usr/bin/env python3
"""BASELINE.md §2 runs B1/B2: the restated reference CPU path (oracle) on this box's host cores — single thread with and
without PolicyWithCache, then thread-per-worker on all cores. Lives under tests/ because it runs the oracle
(test infrastructure). Usage: python tests/cpu_baseline_runs.py [games_single] [threads]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib
from bench import make_weights

oracle = oracle_lib.load()
blob = make_weights()
cfg = oracle_lib.parity_rollout_config(800)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
out = {"cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
       "logical_cpus": os.cpu_count()}
for name, threads, games, cache in (("B1_single_thread_no_cache", 1, n1, False), ("B1_single_thread_cache", 1, n1, True),
                                    ("B2_all_threads_cache", T, 4 * T, True)):
    r = oracle.c4_selfplay(cfg, blob, base_seed=0, n_games=games, threads=threads, use_cache=cache,
                           nn_mode=oracle.ACC_SLIMNN, outputs=False)
    c, s = r["counters"], r["seconds"]
    out[name] = {"threads": threads, "games": games, "seconds": round(s, 3), "games_per_s": games / s,
                 "leaf_evals_per_s": c["policy_evals"] / s, "explores_per_s": c["explores"] / s,
                 "plies_per_game": c["moves"] / games if "moves" in c else None}
print(json.dumps(out, indent=1))
