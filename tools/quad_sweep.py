#!/usr/bin/env python3
"""Developer tool: throughput of the fused kernels for (concurrency, SYN_QUADS) combinations."""
import os
os.environ["SYN_DEBUG"] = "1"  # developer knobs (SYN_LANES, SYN_PROFILE, ...) are honoured only with SYN_DEBUG=1
import os, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthesis_amd as sa
from bench import make_weights
blob = make_weights()
combos = [tuple(map(int, a.split(":"))) for a in sys.argv[1:]] or [(4096, 0), (8192, 0), (8192, 2), (12288, 3), (16384, 4)]
cfg = sa.parity_rollout_config(800)
for conc, nq in combos:
    os.environ["SYN_QUADS"] = str(nq)
    eng = sa.Engine(concurrent_games=conc, max_explores=800)
    eng.load_weights(blob)
    eng.selfplay(cfg, 0, conc, outputs=False)
    n = 4 * conc
    t0 = time.perf_counter()
    r = eng.selfplay(cfg, 0, n, first_game=conc, outputs=False)
    dt = time.perf_counter() - t0
    print(f"concurrent={conc} quads={nq}: {n} games in {dt:.3f} s = {n / dt:.0f} games/s", flush=True)
    eng.close()
