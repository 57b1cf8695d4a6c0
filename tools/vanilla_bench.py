"""Throughput of the evaluator's baseline on the device: VanillaMCTS<a> vs VanillaMCTS<b> matches (evaluator.rs:200-228),
all games of a batch in lockstep. (The same matches on the CPU oracle, for scale and a move-for-move check:
tests/vanilla_baseline_cpu.py.)  usage: python tools/vanilla_bench.py [games] [explores_a] [explores_b] [moves_out.npy]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import synthesis_amd as sa  # noqa: E402
from synthesis_amd import match  # noqa: E402

games = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ea = int(sys.argv[2]) if len(sys.argv) > 2 else 800
eb = int(sys.argv[3]) if len(sys.argv) > 3 else 200
moves_out = sys.argv[4] if len(sys.argv) > 4 else None

eng = sa.Engine(concurrent_games=games, max_explores=max(ea, eb))
a, b = match.vanilla_player(ea), match.vanilla_player(eb)
match.play_match(eng, a, b, min(games, 512), seed=1)  # warm-up
t = time.time()
kernel_ms = 0.0
rec = {}
reward, plies = match.play_match(eng, a, b, games, seed=0, record=rec)
dt = time.time() - t
w, d, l, s, elo = match.score(reward)
searches = int(plies.sum())
print(f"{games} matches VanillaMCTS{ea} vs VanillaMCTS{eb}: {dt:.2f} s wall = {games / dt:.0f} matches/s, "
      f"{searches / dt:.0f} searches/s, {rec['rng_words'].sum() / dt / 1e6:.0f} M playout moves/s; first player scores {s:.3f}")
if moves_out:
    np.save(moves_out, np.concatenate([rec["moves"][:64], reward[:64, None].astype(np.int8).view(np.uint8)], axis=1))
