"""Microbenchmark of the learner-side kernels: train step latency and replay de-duplication rate.
Usage: python tools/train_bench.py   (needs a GPU)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import synthesis_amd as sa  # noqa: E402

blob = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c4net_blob_f32.npy"))
eng = sa.Engine(concurrent_games=4096, max_explores=800)
eng.load_weights(blob)
r = eng.selfplay(sa.parity_rollout_config(64), base_seed=1, n_games=8192)
my = np.concatenate([r["states_bb"][g, : r["plies"][g], 0] for g in range(8192)])
op = np.concatenate([r["states_bb"][g, : r["plies"][g], 1] for g in range(8192)])
pi = np.concatenate([r["pis"][g, : r["plies"][g]] for g in range(8192)])
v = np.concatenate([r["vs"][g, : r["plies"][g]] for g in range(8192)])
print("states", my.size)
for rep in range(2):
    t = time.perf_counter(); d = eng.replay_deduplicate(my, op, pi, v); dt = time.perf_counter() - t
print(f"dedup: {my.size} -> {d['num'].size} in {dt*1e3:.2f} ms (host buffers, incl. copies) = {my.size/dt/1e6:.1f} M states/s")
eng.trainer_init(blob)
for B in (32, 256, 1024):
    n = 200
    idx = np.random.default_rng(0).integers(0, d["num"].size, size=(n, B))
    eng.train_step(d["my_bb"][idx[0]], d["op_bb"][idx[0]], d["pis"][idx[0]], d["vs"][idx[0]], 1e-3)
    t = time.perf_counter()
    for i in range(n):
        eng.train_step(d["my_bb"][idx[i]], d["op_bb"][idx[i]], d["pis"][idx[i]], d["vs"][idx[i]], 1e-3)
    dt = (time.perf_counter() - t) / n
    print(f"train_step B={B}: {dt*1e6:.1f} us/step (host batch in, losses out) = {B/dt:.0f} samples/s")
# device-resident epochs (syn_train_set_data + syn_train_epoch): the path the learning loop uses
eng.train_set_data(d["my_bb"], d["op_bb"], d["pis"], d["vs"])
nu = d["num"].size
perm = np.random.default_rng(1).permutation(nu)
steps = nu // 32
eng.train_epoch(perm[: 256 * 32], 32, 1e-3)
t = time.perf_counter(); eng.train_epoch(perm[: steps * 32], 32, 1e-3); dt = time.perf_counter() - t
print(f"train_epoch B=32: {steps} steps in {dt*1e3:.1f} ms = {dt/steps*1e6:.2f} us/step = {steps/dt:.0f} steps/s "
      f"({'VALU kernel' if os.environ.get('SYN_TRAIN_VALU') else 'matrix-core kernel'})")

# the learner of the conv policy/value network (Connect4ConvNet, train_conv.cuh): queued gradient + Adam launches per step
from bench import make_conv_weights  # noqa: E402
eng.trainer_init_conv(make_conv_weights())
eng.train_set_data(d["my_bb"], d["op_bb"], d["pis"], d["vs"])
eng.train_epoch(perm[: 64 * 32], 32, 1e-3)
csteps = min(steps, 2000)
t = time.perf_counter(); eng.train_epoch(perm[: csteps * 32], 32, 1e-3); dt = time.perf_counter() - t
print(f"train_epoch Connect4ConvNet B=32: {csteps} steps in {dt*1e3:.1f} ms = {dt/csteps*1e6:.2f} us/step = {csteps/dt:.0f} steps/s (VALU kernel)")
