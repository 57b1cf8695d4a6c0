#!/usr/bin/env python3
"""Developer tool: BASELINE configs[1] as worded (4,096 concurrent games, host-side MCTS, batched HIP inference:
syn_selfplay_run_lockstep) by host thread count; and the search form on 4,096 roots. One JSON line per run.
  usage: python3 tools/lockstep_bench.py [--threads 0,16] [--explores 800] [--games 4096]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="0")
    ap.add_argument("--explores", type=int, default=800)
    ap.add_argument("--games", type=int, default=4096)
    args = ap.parse_args()
    import synthesis_amd as sa

    eng = sa.Engine(concurrent_games=4096, max_explores=args.explores)
    eng.load_weights(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "c4net_blob_f32.npy")))
    cfg = sa.parity_rollout_config(args.explores)
    eng.selfplay_lockstep(sa.parity_rollout_config(8), 1, 64)
    for th in [int(x) for x in args.threads.split(",")]:
        t0 = time.perf_counter()
        r = eng.selfplay_lockstep(cfg, 3, args.games, host_threads=th)
        dt = time.perf_counter() - t0
        st = r["stats"]
        print(json.dumps({"lib": os.environ.get("SYNTHESIS_AMD_LIB", "in-tree"), "host_threads": th, "games": args.games,
                          "games_per_s": round(args.games / dt, 1), "seconds": round(dt, 3), "seconds_policy": round(st["seconds_policy"], 3),
                          "launches": st["rounds"], "evals": st["positions_evaluated"], "plies": int(r["plies"].sum())}), flush=True)
    # the search form on mid-game roots taken from the games just played (ply 10 of every game that has one)
    keep = r["plies"] > 10
    my, op = r["states_bb"][keep, 10, 0][:4096].copy(), r["states_bb"][keep, 10, 1][:4096].copy()
    for th in [int(x) for x in args.threads.split(",")]:
        t0 = time.perf_counter()
        sr = eng.mcts_search_lockstep(cfg.mcts_cfg, my, op, args.explores, host_threads=th)
        dt = time.perf_counter() - t0
        print(json.dumps({"search_roots": int(my.size), "host_threads": th, "searches_per_s": round(my.size / dt, 1), "seconds": round(dt, 3),
                          "seconds_policy": round(sr["stats"]["seconds_policy"], 3), "launches": sr["stats"]["rounds"]}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
