#!/usr/bin/env python3
"""Developer micro-benchmarks (not part of the judged bench): phase breakdown of the fused kernel (SYN_PROFILE=1),
stand-alone Connect4Net MFMA throughput, concurrency sweep."""
import argparse
import os
os.environ["SYN_DEBUG"] = "1"  # developer knobs (SYN_LANES, SYN_PROFILE, ...) are honoured only with SYN_DEBUG=1
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (import BEFORE the engine library: one HIP runtime per process, torch's bundled one)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthesis_amd as sa  # noqa: E402
from bench import FLOP_PER_EVAL, make_weights  # noqa: E402


def policy_eval_throughput(eng, n):
    import torch

    rng = np.random.RandomState(0)
    my = torch.from_numpy(rng.randint(0, 2**62, size=n, dtype=np.int64)).cuda()
    op = torch.from_numpy(rng.randint(0, 2**62, size=n, dtype=np.int64)).cuda()
    op = op & ~my
    logits = torch.empty((n, 9), dtype=torch.float32, device="cuda")
    value = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        eng.policy_eval_device(my.data_ptr(), op.data_ptr(), n, logits.data_ptr(), value.data_ptr(), sync=True)
    ms = []
    for _ in range(10):
        eng.policy_eval_device(my.data_ptr(), op.data_ptr(), n, logits.data_ptr(), value.data_ptr(), sync=True)
        ms.append(eng.last_kernel_ms())
    ms = float(np.median(ms))
    print(f"policy_eval n={n}: {ms:.3f} ms  {n / ms / 1e3:.1f} M evals/s  {n * FLOP_PER_EVAL / ms / 1e9:.2f} TFLOP/s "
          f"({n * FLOP_PER_EVAL / ms / 1e9 / 157.3 * 100:.1f}% of f32 MFMA peak)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="all")
    ap.add_argument("--concurrent", type=int, default=4096)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--explores", type=int, default=800)
    args = ap.parse_args()
    blob = make_weights()
    if args.what in ("all", "eval"):
        eng = sa.Engine(concurrent_games=16, max_explores=8)
        eng.load_weights(blob)
        for n in (4096, 65536, 1 << 20, 1 << 22):
            policy_eval_throughput(eng, n)
        eng.close()
    if args.what in ("all", "phases"):
        os.environ["SYN_PROFILE"] = "1"
        eng = sa.Engine(concurrent_games=args.concurrent, max_explores=args.explores)
        eng.load_weights(blob)
        r = eng.selfplay(sa.parity_rollout_config(args.explores), 0, args.games, outputs=False)
        print(f"profiled run: {args.games} games, kernel {r['kernel_ms']:.1f} ms")
        del os.environ["SYN_PROFILE"]
        eng.close()
    if args.what in ("all", "sweep"):
        for conc in (1024, 2048, 4096, 8192, 16384):
            eng = sa.Engine(concurrent_games=conc, max_explores=args.explores)
            eng.load_weights(blob)
            cfg = sa.parity_rollout_config(args.explores)
            eng.selfplay(cfg, 0, conc, outputs=False)
            n = 4 * conc
            t0 = time.perf_counter()
            r = eng.selfplay(cfg, 0, n, first_game=conc, outputs=False)
            dt = time.perf_counter() - t0
            print(f"concurrent={conc}: {n} games in {dt:.3f} s = {n / dt:.0f} games/s (kernel {r['kernel_ms']:.1f} ms)")
            eng.close()


if __name__ == "__main__":
    main()
