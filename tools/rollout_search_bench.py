"""Developer tool: throughput of MCTS over RolloutPolicy (syn_mcts_search_rollout, lane-per-tree kernel) from the empty board."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synthesis_amd as sa
from synthesis_amd import match
n = 65536
eng = sa.Engine(concurrent_games=n, max_explores=400)
p = match.rollout_player(400)
z = np.zeros(n, np.uint64)
eng.mcts_search(p.mcts_cfg, z[:1024], z[:1024], 400, rollout_seed=1)
t = time.time()
r = eng.mcts_search(p.mcts_cfg, z, z, 400, rollout_seed=5)
dt = time.time() - t
print(f"MCTS over RolloutPolicy: {n} roots x 400 explores in {dt:.3f} s = {n / dt:.0f} searches/s; checksum {int(r['child_N'].sum())} {int(r['best_action'].sum())}")
