#!/usr/bin/env python3
"""Developer tool: throughput of the lane-per-tree / producer-consumer kernels. Args: conc:nw[:games[:policy_cache_log2]] ...
(nw = waves per workgroup of the lane kernel, 0 = the engine's own choice, negative = producer/consumer kernel with -nw
virtual waves per tree wave)"""
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
combos = [tuple(map(int, a.split(":"))) for a in sys.argv[1:]] or [(65536, 16), (131072, 16), (262144, 16)]
cfg = sa.parity_rollout_config(800)
for c in combos:
    conc, nw = c[0], c[1]
    n = c[2] if len(c) > 2 else 2 * conc
    clog = c[3] if len(c) > 3 else 0
    os.environ.pop("SYN_LANES", None); os.environ.pop("SYN_PC", None)
    if nw > 0: os.environ["SYN_LANES"] = str(nw); os.environ["SYN_PC"] = "0"
    elif nw < 0: os.environ["SYN_PC"] = str(-nw)
    eng = sa.Engine(concurrent_games=conc, max_explores=800, policy_cache_log2=clog)
    eng.load_weights(blob)
    eng.selfplay(cfg, 0, 256, outputs=False)
    t0 = time.perf_counter()
    r = eng.selfplay(cfg, 0, n, first_game=conc, outputs=False)
    dt = time.perf_counter() - t0
    hits, misses = eng.last_cache_stats()
    extra = f"  policy cache 2^{clog}: {hits / max(1, hits + misses):.3f} hit rate" if clog else ""
    print(f"concurrent={conc} lanes_nw={nw} shape={eng.last_launch_shape()}: {n} games in {dt:.3f} s = {n / dt:.0f} games/s  (mean plies {r['plies'].mean():.2f}){extra}", flush=True)
    eng.close()
