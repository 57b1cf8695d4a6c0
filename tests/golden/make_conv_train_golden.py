#!/usr/bin/env python3
"""Generates tests/golden/conv_train_torch_goldens.npz: an independent second opinion (this container's torch, float64) for the
learner step of Connect4ConvNet (Conv2d<2,16,3,pad 1> + ReLU + Linear<1008,12>; the loss and Adam of alpha_zero.rs:72-94) — 8
consecutive minibatches of 32 positions taken from train_torch_goldens.npz (realistic self-play positions and targets), final
weights, per-step losses and the first step's gradient. The reference has no such network; run once in the authoring container."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_conv_weights  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
g = np.load(os.path.join(here, "train_torch_goldens.npz"))
S, B = 8, 32
my = g["my_bb"].reshape(-1)[: S * B].reshape(S, B); op = g["op_bb"].reshape(-1)[: S * B].reshape(S, B)
TPI = g["target_pi"].reshape(-1, 9)[: S * B].reshape(S, B, 9); TV = g["target_v"].reshape(-1, 3)[: S * B].reshape(S, B, 3)
blob = make_conv_weights()
lrs = np.array([1e-3] * 4 + [5e-4] * 4, np.float32)


def planes(my_row, op_row):
    x = np.zeros((len(my_row), 2, 7, 9), np.float64)
    for b, (m, o) in enumerate(zip(my_row, op_row)):
        for pl, bb in enumerate((int(m), int(o))):
            for r in range(7):
                for c in range(9):
                    x[b, pl, r, c] = (bb >> (r + 7 * c)) & 1
    return x


def run(dtype):
    cw = torch.tensor(blob[:288].reshape(16, 2, 3, 3), dtype=dtype, requires_grad=True)
    cb = torch.tensor(blob[288:304], dtype=dtype, requires_grad=True)
    hw = torch.tensor(blob[304:304 + 12 * 1008].reshape(12, 1008), dtype=dtype, requires_grad=True)
    hb = torch.tensor(blob[-12:], dtype=dtype, requires_grad=True)
    params = [cw, cb, hw, hb]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-6)
    losses, g0 = [], None
    for s in range(S):
        for pg in opt.param_groups:
            pg["lr"] = float(lrs[s])
        x = torch.tensor(planes(my[s], op[s]), dtype=dtype)
        y = torch.relu(torch.nn.functional.conv2d(x, cw, cb, padding=1)).reshape(B, -1)
        out = torch.nn.functional.linear(y, hw, hb)
        lp = torch.log_softmax(out[:, :9], -1); lv = torch.log_softmax(out[:, 9:], -1)
        pl = (1.0 / B) * torch.nn.functional.kl_div(lp, torch.tensor(TPI[s], dtype=dtype), reduction="sum")
        vl = (1.0 / B) * torch.nn.functional.kl_div(lv, torch.tensor(TV[s], dtype=dtype), reduction="sum")
        loss = 1.0 * pl + 1.0 * vl
        opt.zero_grad(); loss.backward()
        if s == 0:
            g0 = np.concatenate([p.grad.detach().numpy().ravel() for p in params])
        opt.step()
        losses.append([float(pl), float(vl)])
    w = np.concatenate([p.detach().numpy().ravel() for p in params])
    return w, np.array(losses), g0


w64, l64, g64 = run(torch.float64)
np.savez_compressed(os.path.join(here, "conv_train_torch_goldens.npz"), my_bb=my, op_bb=op, target_pi=TPI, target_v=TV, lrs=lrs,
                    final_weights_f64=w64, losses_f64=l64, first_grad_f64=g64)
print("wrote conv_train_torch_goldens.npz", w64.shape, l64[0], l64[-1])
