#!/usr/bin/env python3
"""Developer tool (GPU): achieved HBM rate of the stand-alone streaming kernels — features_kernel, linear_kernel, conv2d_kernel,
activation_kernel, policy_eval_kernel — and the search rate of frozen_search_kernel, each against its roof. The layer entry points
take host pointers, so the kernel time is taken from HIP events inside the library (syn_last_timing) and the bytes are the
algorithmic ones (inputs read once + outputs written once). Writes one JSON object (profiles/rNN_small_kernels.json)."""
import json
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthesis_amd as sa  # noqa: E402
from bench import make_weights  # noqa: E402

PEAK_HBM = 8000.0
eng = sa.Engine(concurrent_games=65536, max_explores=800)
eng.load_weights(make_weights())
rng = np.random.default_rng(0)
out = {}


def line(name, nbytes, ms, extra=None):
    d = {"bound": "hbm", "algorithmic_bytes": int(nbytes), "kernel_ms": ms, "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM,
         "unit": "GB/s"}
    d["frac"] = d["achieved"] / PEAK_HBM
    if extra:
        d.update(extra)
    out[name] = d
    print(name, json.dumps(d), flush=True)


n = 4_000_000
my = rng.integers(0, 1 << 62, n, dtype=np.uint64); op = (~my) & np.uint64((1 << 62) - 1)
eng.features(my[:1000], op[:1000])
eng.features(my, op)
line("features4_kernel", n * (16 + 63 * 4), eng.last_kernel_ms(), {"positions": n})
eng.policy_eval(my, op)
ms = eng.last_kernel_ms()
line("policy_eval_kernel (compute-bound: f32 MFMA)", n * 64, ms, {"positions": n, "tflops": n * 60288 / (ms * 1e-3) / 1e12,
                                                                    "mfma_frac": n * 60288 / (ms * 1e-3) / 1e12 / 157.3})
# the same network in the f16x2 arithmetic (f16x2_tile.cuh): priced against the f16 matrix peak by the MFMAs it executes (184,320 FLOP
# per evaluation: three products, inputs padded to 32s) and, for comparison with the line above, against the f32 peak by the 60,288 FLOP
# the network needs
eng.set_network_arithmetic("f16x2")
eng.policy_eval(my[:65536], op[:65536]); eng.policy_eval(my, op)
ms = eng.last_kernel_ms()
line("policy_eval_f16x2_kernel (compute-bound: f16 MFMA)", n * 64, ms, {"positions": n, "g_evals_per_s": n / (ms * 1e-3) / 1e9,
                                                                          "tflops_executed": n * 184320 / (ms * 1e-3) / 1e12,
                                                                          "f16_mfma_frac": n * 184320 / (ms * 1e-3) / 1e12 / 2500.0,
                                                                          "f32_equivalent_tflops": n * 60288 / (ms * 1e-3) / 1e12,
                                                                          "f32_equivalent_frac_of_157.3": n * 60288 / (ms * 1e-3) / 1e12 / 157.3})
eng.set_network_arithmetic("f32")
# the conv policy/value network's stand-alone evaluation (convnet.cuh), on its own engine (it replaces the engine's network)
from bench import CONV_FLOP_PER_EVAL, make_conv_weights  # noqa: E402
ce = sa.Engine(concurrent_games=256, max_explores=16)
ce.load_weights_conv(make_conv_weights())
ce.policy_eval(my[:65536], op[:65536]); ce.policy_eval(my, op)
ms = ce.last_kernel_ms()
line("policy_eval_conv_kernel (compute-bound: f32 MFMA)", n * 64, ms, {"positions": n, "tflops": n * CONV_FLOP_PER_EVAL / (ms * 1e-3) / 1e12,
                                                                         "mfma_frac": n * CONV_FLOP_PER_EVAL / (ms * 1e-3) / 1e12 / 157.3})
ce.close()
B = 1_000_000
W = rng.standard_normal((128, 63), dtype=np.float32); b = rng.standard_normal(128, dtype=np.float32)
x = rng.standard_normal((B, 63), dtype=np.float32)
eng.linear(W, b, x[:100]); eng.linear(W, b, x)
ms = eng.last_kernel_ms()
# slimnn's two-rounding multiply-add cannot use fma or the matrix cores: 2 FLOP per MAC at the packed non-fused VALU rate
# (v_pk_mul_f32 + v_pk_add_f32: 256 CUs x 64 lanes x 2 x 2.4 GHz = 78.6 TFLOP/s) is the roof that binds, long before HBM
line("linear_tiled_kernel 63->128", B * (63 + 128) * 4, ms, {"batch": B, "valu_tflops": B * 63 * 128 * 2 / (ms * 1e-3) / 1e12,
                                                              "valu_frac_of_78.6_tflops": B * 63 * 128 * 2 / (ms * 1e-3) / 1e12 / 78.6})
Wc = rng.standard_normal((4, 2, 3, 3), dtype=np.float32); bc = rng.standard_normal(4, dtype=np.float32)
xc = rng.standard_normal((B, 2, 7, 9), dtype=np.float32)
yc = eng.conv2d(Wc, bc, xc, row_pad=1, col_pad=1)
ms = eng.last_kernel_ms()
line("conv2d_tiled_kernel 2->4 3x3 pad 1 on 7x9", B * (2 * 63 + 4 * 63) * 4, ms,
     {"batch": B, "valu_tflops": B * 4 * 2 * 475 * 2 / (ms * 1e-3) / 1e12, "note": "475 in-board taps per (channel, plane); index and bounds arithmetic dominates"})
xa = rng.standard_normal((B, 12), dtype=np.float32)
eng.activation(1, xa); eng.activation(1, xa)
line("activation_kernel tanh", B * 12 * 8, eng.last_kernel_ms())
eng.activation(2, xa)
line("activation_kernel softmax(12)", B * 12 * 8, eng.last_kernel_ms())
# evaluator baseline: FrozenMCTS over RolloutPolicy, 800 explores from the empty board on 131,072 trees (latency-bound pointer
# chasing + integer playouts: reported as searches/s and explores/s, no roofline claim)
roll = sa.MCTSConfig(exploration=sa.Exploration.Uct, c=2.0, auto_extend=False, fpu=sa.Fpu.Const, fpu_value=float("inf"))
m = 131072
z = np.zeros(m, np.uint64)
eng.frozen_search(roll, np.arange(m, dtype=np.uint64), np.zeros(m, np.uint64), z, z, 800, action_selection=0)
ms = eng.last_kernel_ms()
out["frozen_search_kernel"] = {"searches": m, "explores_each": 800, "kernel_ms": ms, "searches_per_s": m / (ms * 1e-3),
                               "explores_per_s": m * 800 / (ms * 1e-3), "bound": "latency (dependent pointer chase + playouts)"}
print("frozen_search_kernel", json.dumps(out["frozen_search_kernel"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "small_kernels.json"), "w"), indent=1)
