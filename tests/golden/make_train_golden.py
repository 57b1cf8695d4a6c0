#!/usr/bin/env python3
"""Generates tests/golden/train_torch_goldens.npz: an independent second opinion (this container's torch, float64) for
the learner step that follows the self-play path (alpha_zero.rs:72-94: forward, log_softmax, kl_div(Sum)/batch, Adam with
weight decay) — 12 consecutive minibatches of 32 self-play positions (positions/targets produced by the oracle's
self-play so they are realistic), final weights, per-step losses and the first step's gradient. No reference test covers
this step; run once in the authoring container."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib  # noqa: E402

o = oracle_lib.load()
here = os.path.dirname(os.path.abspath(__file__))
blob = np.load(os.path.join(here, "c4net_blob_f32.npy"))
dims = [63, 128, 96, 64, 48, 12]
rng = np.random.RandomState(7)
B, S = 32, 12
r = o.c4_selfplay(oracle_lib.parity_rollout_config(40), blob, 3, 40, threads=4, nn_mode=1)
my, op, pi, v = [], [], [], []
for g in range(40):
    n = r["plies"][g]
    my += list(r["states_bb"][g, :n, 0]); op += list(r["states_bb"][g, :n, 1])
    pi += list(r["pis"][g, :n]); v += list(r["vs"][g, :n])
idx = rng.permutation(len(my))[: B * S]
my = np.array(my, np.uint64)[idx]; op = np.array(op, np.uint64)[idx]
pi = np.array(pi, np.float32)[idx]; v = np.array(v, np.float32)[idx]
X = o.c4_features(my, op).reshape(S, B, 63); TPI = pi.reshape(S, B, 9); TV = v.reshape(S, B, 3)
lrs = np.array([1e-3] * 6 + [5e-4] * 6, np.float32)


def run(dtype):
    params, off = [], 0
    for i in range(5):
        W = torch.tensor(blob[off:off + dims[i] * dims[i + 1]].reshape(dims[i + 1], dims[i]), dtype=dtype, requires_grad=True)
        off += dims[i] * dims[i + 1]
        b = torch.tensor(blob[off:off + dims[i + 1]], dtype=dtype, requires_grad=True)
        off += dims[i + 1]
        params += [W, b]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-6)
    losses, g0 = [], None
    for s in range(S):
        for g in opt.param_groups:
            g["lr"] = float(lrs[s])
        x = torch.tensor(X[s], dtype=dtype)
        for i in range(5):
            x = torch.nn.functional.linear(x, params[2 * i], params[2 * i + 1])
            if i < 4:
                x = torch.relu(x)
        lp = torch.log_softmax(x[:, :9], -1); lv = torch.log_softmax(x[:, 9:], -1)
        pl = (1.0 / B) * torch.nn.functional.kl_div(lp, torch.tensor(TPI[s], dtype=dtype), reduction="sum")
        vl = (1.0 / B) * torch.nn.functional.kl_div(lv, torch.tensor(TV[s], dtype=dtype), reduction="sum")
        loss = 1.0 * pl + 1.0 * vl
        opt.zero_grad(); loss.backward()
        if s == 0:
            g0 = np.concatenate([p.grad.detach().numpy().ravel() for p in params])
        opt.step(); losses.append([pl.item(), vl.item()])
    return np.concatenate([p.detach().numpy().ravel() for p in params]), np.array(losses), g0


w64, l64, g64 = run(torch.float64)
w32, l32, g32 = run(torch.float32)
hp = oracle_lib.default_train_hyper()
wo, mo, vo, st, lo = o.train_steps(blob, hp, X, TPI, TV, lrs)
go, _ = o.train_gradients(blob, hp, X[0], TPI[0], TV[0])
upd = np.abs(w64 - blob).max()
print("oracle vs torch64: max|dw|", np.abs(wo - w64).max(), "relative to the largest update", np.abs(wo - w64).max() / upd,
      "loss err", np.abs(lo - l64).max(), "grad err", np.abs(go - g64).max(), "max|grad|", np.abs(g64).max())
print("torch32 vs torch64: max|dw|", np.abs(w32 - w64).max(), "loss", np.abs(l32 - l64).max())
np.savez_compressed(os.path.join(here, "train_torch_goldens.npz"), my_bb=my.reshape(S, B), op_bb=op.reshape(S, B),
                    target_pi=TPI, target_v=TV, lrs=lrs, weights_f64=w64, losses_f64=l64, grad0_f64=g64)
print("written", os.path.join(here, "train_torch_goldens.npz"))
