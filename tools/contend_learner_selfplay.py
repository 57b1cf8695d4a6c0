#!/usr/bin/env python3
"""Developer tool (GPU): the conv learner's four-workgroup epoch kernel launched while another engine's self-play kernel holds every
CU. The epoch's workgroups become resident as that launch drains (or the call falls back to the one-workgroup kernel after ~10 s);
either way the epoch must end with the weights and losses of the same epoch run alone. Measured: 7.0 s beside an 8.2 s self-play
launch, identical."""
import sys, threading, time, numpy as np
sys.path.insert(0, ".")
import synthesis_amd as sa
from bench import make_weights, make_conv_weights
blob, cblob = make_weights(), make_conv_weights()
a = sa.Engine(concurrent_games=262144, max_explores=800); a.load_weights(blob)
b = sa.Engine(concurrent_games=4096, max_explores=64); b.load_weights(blob)
r = b.selfplay(sa.parity_rollout_config(64), base_seed=1, n_games=4096)
sel = np.arange(63)[None, :] < r["plies"][:, None]
d = b.replay_deduplicate(r["states_bb"][..., 0][sel], r["states_bb"][..., 1][sel], r["pis"][sel], r["vs"][sel])
perm = np.random.default_rng(1).permutation(int(d["num"].size)).astype(np.int32)[: 1500 * 32]
b.trainer_init_conv(cblob); b.train_set_data(d["my_bb"], d["op_bb"], d["pis"], d["vs"])
l0 = b.train_epoch(perm, 32, 1e-3); w0 = b.trainer_state()["weights"].copy()     # alone
b.trainer_init_conv(cblob); b.train_set_data(d["my_bb"], d["op_bb"], d["pis"], d["vs"])
cfg = sa.parity_rollout_config(800)
a.selfplay(cfg, 0, 4096, outputs=False)
out = {}
def play():
    t = time.perf_counter(); a.selfplay(cfg, 0, 524288, first_game=4096, outputs=False); out["selfplay_s"] = time.perf_counter() - t
th = threading.Thread(target=play); th.start(); time.sleep(1.0)
t = time.perf_counter(); l1 = b.train_epoch(perm, 32, 1e-3); out["train_s"] = time.perf_counter() - t
th.join()
w1 = b.trainer_state()["weights"]
print(out, "identical to the run alone:", bool(np.array_equal(w0, w1) and np.array_equal(l0, l1)))
