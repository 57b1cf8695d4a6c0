"""The evaluator's baseline matches on the CPU oracle (test infrastructure; one thread): VanillaMCTS<a> vs VanillaMCTS<b>
(evaluator.rs:200-228), seeds 0..n-1 — the CPU side of tools/vanilla_bench.py. With a moves file written by that tool
(first 64 matches of the device run) every match is also compared move for move.
usage: python tests/vanilla_baseline_cpu.py [matches] [explores_a] [explores_b] [moves.npy]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tests import oracle_lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ea = int(sys.argv[2]) if len(sys.argv) > 2 else 800
eb = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = np.load(sys.argv[4]) if len(sys.argv) > 4 else None

o = oracle_lib.load()
cfg = oracle_lib.parity_mcts_config(exploration=0, c=2.0, auto_extend=0, fpu_value=float("inf"))
t = time.time()
same = 0
for g in range(n):
    r, moves, _ = o.c4_mcts_vs_mcts(cfg, 0, ea, eb, g, rollout_action=0)  # rollout_action: Q (main.rs:72), as vanilla_player
    if dev is not None and g < dev.shape[0]:
        same += int(np.array_equal(dev[g, :moves.size], moves) and np.int8(r) == dev[g, 63].view(np.int8))
dt = time.time() - t
msg = f"oracle, 1 thread: {n} matches VanillaMCTS{ea} vs VanillaMCTS{eb} in {dt:.2f} s = {n / dt:.1f} matches/s"
if dev is not None:
    msg += f"; {same}/{min(n, dev.shape[0])} matches identical to the device run move for move"
print(msg)
