#!/usr/bin/env python3
"""Developer tool: host-visible latency of syn_policy_eval_batch (the batched Policy::eval, policies/traits.rs:4-6) from pageable host
buffers — what a Rust `impl Policy for HipPolicy` pays per call — by batch size. Prints one JSON line.
  usage: python3 tools/eval_latency.py [--reps 300]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--sizes", default="1,16,256,1024,4096,16384,65536,1048576")
    args = ap.parse_args()
    import synthesis_amd as sa
    from synthesis_amd.engine import _p

    eng = sa.Engine(concurrent_games=4096, max_explores=64)
    eng.load_weights(sa.init_weights(0) if hasattr(sa, "init_weights") else np.load(
        os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c4net_blob_f32.npy")))
    rng = np.random.default_rng(1)
    out = {"lib": os.environ.get("SYNTHESIS_AMD_LIB", "in-tree"), "us_per_call": {}, "evals_per_s": {}, "kernel_us": {}}
    for n in [int(x) for x in args.sizes.split(",")]:
        # random legal-looking positions: disjoint bitboards inside the 63 board bits (the network does not care about reachability)
        a = rng.integers(0, 2 ** 62, n, dtype=np.uint64)
        b = rng.integers(0, 2 ** 62, n, dtype=np.uint64)
        my = a & ~b
        op = b & ~a
        logits = np.zeros((n, 9), np.float32)
        value = np.zeros((n, 3), np.float32)
        fn = eng._lib.syn_policy_eval_batch
        h = eng._h
        pm, po, pl, pv = _p(my), _p(op), _p(logits), _p(value)
        reps = max(5, min(args.reps, int(3e7 // max(n, 1)) + 5))
        for _ in range(5):
            assert fn(h, pm, po, n, pl, pv) == 0
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(h, pm, po, n, pl, pv)
        dt = (time.perf_counter() - t0) / reps
        out["us_per_call"][str(n)] = round(dt * 1e6, 2)
        out["evals_per_s"][str(n)] = round(n / dt)
        out["kernel_us"][str(n)] = round(eng.last_kernel_ms() * 1e3, 2)   # between the two events around the kernel, last call
        # the answers must not depend on the transfer path: against one big call of the same positions
        if n <= 65536:
            l2, v2 = eng.policy_eval(np.concatenate([my, my]), np.concatenate([op, op]))
            assert np.array_equal(l2[:n].view(np.uint32), logits.view(np.uint32)) and np.array_equal(v2[n:].view(np.uint32), value.view(np.uint32))
    # the context path, split: submit (staging + launch, returns at once) and wait (until the answers are copied out)
    ctx = eng.eval_context()
    lib, c = eng._lib, ctx._c
    out["ctx_submit_us"], out["ctx_wait_us"] = {}, {}
    for n in (1, 256, 512, 1024, 2048, 4096):
        a = rng.integers(0, 2 ** 62, n, dtype=np.uint64)
        b = rng.integers(0, 2 ** 62, n, dtype=np.uint64)
        my, op = a & ~b, b & ~a
        logits = np.zeros((n, 9), np.float32)
        value = np.zeros((n, 3), np.float32)
        pm, po, pl, pv = _p(my), _p(op), _p(logits), _p(value)
        ts = tw = 0.0
        for i in range(205):
            t0 = time.perf_counter()
            lib.syn_eval_ctx_submit(c, pm, po, n)
            t1 = time.perf_counter()
            lib.syn_eval_ctx_wait(c, pl, pv)
            t2 = time.perf_counter()
            if i >= 5:
                ts += t1 - t0
                tw += t2 - t1
        out["ctx_submit_us"][str(n)] = round(ts / 200 * 1e6, 2)
        out["ctx_wait_us"][str(n)] = round(tw / 200 * 1e6, 2)
    ctx.close()
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
