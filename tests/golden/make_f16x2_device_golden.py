"""Writes gpurun_out/c4net_f16x2_device.npz: what the HIP engine's SYN_NET_ARITH_F16X2 arithmetic computes ON AN MI355X for 512 reachable
positions under the random-init blob and the trained checkpoint (raw logits and outcome probabilities), plus the f16x2 plan the engine
chose. Needs a GPU (run through gpurun); the result is committed as tests/golden/c4net_f16x2_device.npz and replayed on the CPU by
tests/test_oracle_f16x2.py against oracle/nn_f16x2.hpp — so the restatement stays pinned to device-produced bits without a GPU.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import synthesis_amd as sa  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.test_gpu_parity import random_positions  # noqa: E402

orc = oracle_lib.load()
my, op = random_positions(orc, 512, seed=1234)
my[0] = 0; op[0] = 0
out = dict(my_bb=my, op_bb=op)
eng = sa.Engine(concurrent_games=256, max_explores=16, device=0)
for name in ("c4net_blob_f32", "c4net_trained_f32"):
    blob = np.load(os.path.join(ROOT, "tests", "golden", name + ".npy"))
    eng.load_weights(blob)
    eng.set_network_arithmetic("f16x2")
    l, v = eng.policy_eval(my, op)
    _, plan = eng.network_arithmetic()
    out[name + "_logits"] = l; out[name + "_value"] = v
    out[name + "_plan"] = np.array(plan["activation_exp"] + plan["weight_exp"] + [plan["out_exp"]], np.int32)
    eng.set_network_arithmetic("f32")
eng.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "c4net_f16x2_device.npz"), **out)
print("written")
