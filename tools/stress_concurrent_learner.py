#!/usr/bin/env python3
"""Developer check (GPU): the persistent epoch kernel of the learner launched WHILE a self-play kernel of another engine occupies
every CU. Its 16 workers cannot all be resident until the self-play launch drains; the ones that start first wait at the step
barrier (10 s budget) instead of failing — the epoch must finish with the bits of an undisturbed run."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthesis_amd as sa  # noqa: E402
from bench import make_weights  # noqa: E402

blob = make_weights()
play = sa.Engine(concurrent_games=262144, max_explores=800)
play.load_weights(blob)
learn = sa.Engine(concurrent_games=4096, max_explores=64)
learn.load_weights(blob)
r = learn.selfplay(sa.parity_rollout_config(64), base_seed=1, n_games=8192)
sel = np.arange(63)[None, :] < r["plies"][:, None]
d = learn.replay_deduplicate(r["states_bb"][..., 0][sel], r["states_bb"][..., 1][sel], r["pis"][sel], r["vs"][sel])
perm = np.random.default_rng(1).permutation(d["num"].size).astype(np.int32)[: 2000 * 32]


def epoch():
    learn.trainer_init(blob)
    learn.train_set_data(d["my_bb"], d["op_bb"], d["pis"], d["vs"])
    t = time.perf_counter()
    losses = learn.train_epoch(perm, 32, 1e-3)
    return time.perf_counter() - t, learn.trainer_state()["weights"], losses


dt0, w0, l0 = epoch()
print(f"undisturbed: 2000 steps in {dt0 * 1e3:.1f} ms")
box = {}
th = threading.Thread(target=lambda: box.update(sp=play.selfplay(sa.parity_rollout_config(800), base_seed=0, n_games=524288, outputs=False)))
t0 = time.perf_counter()
th.start()
time.sleep(1.0)          # the self-play kernel now holds all 256 CUs
dt1, w1, l1 = epoch()
t_epoch_done = time.perf_counter() - t0
th.join()
t_all = time.perf_counter() - t0
print(f"beside a {t_all:.1f} s self-play launch: epoch call returned after {t_epoch_done:.1f} s (own time {dt1:.2f} s); "
      f"weights identical: {np.array_equal(w0.view(np.uint32), w1.view(np.uint32))}, losses identical: {np.array_equal(l0, l1)}")
