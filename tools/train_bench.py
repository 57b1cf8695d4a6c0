#!/usr/bin/env python3
"""Developer tool: optimiser steps per second of the learners at the reference's batch of 32 (alpha_zero.rs:72-94 step; main.rs:22).
Connect4Net: persistent epoch kernel (train_epoch.cuh); Connect4ConvNet: persistent four-workgroup epoch kernel (train_conv_mfma.cuh;
SYN_DEBUG=1 SYN_TRAIN_CONV_MW=0: the one-workgroup kernel) in f32 and bf16. SYN_DEBUG=1 SYN_TRAIN_PROFILE=1 prints the phase stamps. Run under rocprofv3 --kernel-trace --stats for per-kernel times."""
import os, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthesis_amd as sa
from bench import make_conv_weights, make_weights

blob, cblob = make_weights(), make_conv_weights()
eng = sa.Engine(concurrent_games=4096, max_explores=64)
eng.load_weights(blob)
r = eng.selfplay(sa.parity_rollout_config(64), base_seed=1, n_games=8192)
sel = np.arange(63)[None, :] < r["plies"][:, None]
d = eng.replay_deduplicate(r["states_bb"][..., 0][sel], r["states_bb"][..., 1][sel], r["pis"][sel], r["vs"][sel])
nu = int(d["num"].size)
perm = np.random.default_rng(1).permutation(nu).astype(np.int32)
steps = min(nu // 32, 3000)
print(f"{nu} unique positions, {steps} steps of 32")
for name, init, w0, prec in (("Connect4Net", eng.trainer_init, blob, None), ("Connect4ConvNet f32", eng.trainer_init_conv, cblob, "f32"),
                             ("Connect4ConvNet bf16", eng.trainer_init_conv, cblob, "bf16")):
    for queued in (False,):   # (SYN_TRAIN_QUEUED is read once per process: run the tool with it set to time the queued launches)
        if queued: os.environ["SYN_DEBUG"] = "1"; os.environ["SYN_TRAIN_QUEUED"] = "1"
        else: os.environ.pop("SYN_TRAIN_QUEUED", None)
        init(w0)
        if prec: eng.trainer_set_precision(prec)
        eng.train_set_data(d["my_bb"], d["op_bb"], d["pis"], d["vs"])
        n = steps if not queued else min(steps, 500)
        eng.train_epoch(perm[: 64 * 32], 32, 1e-3)
        t0 = time.perf_counter()
        l = eng.train_epoch(perm[: n * 32], 32, 1e-3)
        dt = time.perf_counter() - t0
        print(f"{name:22s} {'queued launches' if queued else 'epoch kernel   '}: {n} steps in {dt * 1e3:.1f} ms = {dt / n * 1e6:.2f} us/step = {n / dt:.0f} steps/s   "
              f"(first / last pi-loss {l[0, 0]:.4f} / {l[-1, 0]:.4f})", flush=True)
eng.close()
