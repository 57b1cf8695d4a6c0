"""Developer tool: BASELINE configs[1]'s size — 4,096 concurrent games, 16,384 games — in the f32 and the f16x2 network arithmetic
(args: arithmetic names; SYN_DEBUG=1 SYN_PROFILE=1 prints the phase stamps of the launch shape chosen)."""
import os, time, numpy as np, sys
sys.path.insert(0, ".")
import synthesis_amd as sa
from bench import make_weights
for arith in (sys.argv[1:] or ["f32", "f16x2"]):
    eng = sa.Engine(concurrent_games=4096, max_explores=800)
    eng.load_weights(make_weights())
    eng.set_network_arithmetic(arith)
    cfg = sa.parity_rollout_config(800)
    eng.selfplay(cfg, 1, 4096, outputs=False)
    t = time.perf_counter(); r = eng.selfplay(cfg, 1, 16384, first_game=4096, outputs=False); dt = time.perf_counter() - t
    print(f"4096 concurrent, {arith}: {16384 / dt:.0f} games/s", eng.last_launch_shape(), flush=True)
    eng.close()
