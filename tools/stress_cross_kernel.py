#!/usr/bin/env python3
"""Developer tool (GPU): every launch shape must give the same games for the same (config, seed) — results depend only
on the game index. Plays 3,000 games per configuration family on the row-per-tree kernels and on the lane-per-tree kernel
(8 and 12 waves, with and without the policy cache) and compares every output array."""
import os
os.environ["SYN_DEBUG"] = "1"  # developer knobs (SYN_LANES, SYN_PROFILE, ...) are honoured only with SYN_DEBUG=1
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthesis_amd as sa
from bench import make_weights

blob = make_weights()
variants = {
    "parity": dict(),
    "uct_no_extend": dict(mcts_cfg=sa.MCTSConfig(exploration=sa.Exploration.Uct, c=2.0, auto_extend=False, fpu_value=float("inf"))),
    "parentq_noise": dict(mcts_cfg=sa.MCTSConfig(fpu=sa.Fpu.ParentQ, root_policy_noise=sa.PolicyNoise.Equal, noise_weight=0.25)),
    "nosolve_stop": dict(mcts_cfg=sa.MCTSConfig(solve=False), stop_games_when_solved=True, action=sa.ActionSelection.Q),
    "z_targets_deep": dict(mcts_cfg=sa.MCTSConfig(c=0.5, fpu_value=-1.0), value_target=sa.ValueTarget.QtoZ, value_target_from=0.2, value_target_to=0.8),
}
N, E = 3000, 300
ok = True
for name, kw in variants.items():
    cfg = sa.parity_rollout_config(E, **kw)
    ref = None
    for label, env, cache in (("row", {"SYN_LANES": "0"}, 0), ("lanes8", {"SYN_LANES": "8"}, 0), ("lanes12+cache", {"SYN_LANES": "12"}, 20)):
        for k in ("SYN_LANES",):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = sa.Engine(concurrent_games=2048, max_explores=E, policy_cache_log2=cache)
        eng.load_weights(blob)
        t = time.perf_counter()
        r = eng.selfplay(cfg, base_seed=99, n_games=N, counters=True)
        dt = time.perf_counter() - t
        shape = eng.last_launch_shape()
        eng.close()
        if ref is None:
            ref = r
            print(f"{name:16s} {label:14s} shape {shape} {dt:.2f}s plies {r['plies'].mean():.2f} draws {(r['final_kind'] == 1).mean():.3f}")
            continue
        same = np.array_equal(r["plies"], ref["plies"])
        mask = np.arange(63)[None, :] < ref["plies"][:, None]
        for k in ("states_bb", "pis", "vs", "actions", "root_nodes"):
            same = same and np.array_equal(r[k][mask], ref[k][mask])
        same = same and np.array_equal(r["final_kind"], ref["final_kind"]) and r["counters"] == ref["counters"]
        print(f"{name:16s} {label:14s} shape {shape} {dt:.2f}s identical={same}")
        ok = ok and same
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
