import os, time, numpy as np, sys
sys.path.insert(0, ".")
import synthesis_amd as sa
from bench import make_weights
eng = sa.Engine(concurrent_games=4096, max_explores=800)
eng.load_weights(make_weights())
cfg = sa.parity_rollout_config(800)
eng.selfplay(cfg, 1, 4096, outputs=False)
t=time.perf_counter(); r=eng.selfplay(cfg, 1, 16384, first_game=4096, outputs=False); dt=time.perf_counter()-t
print("4096 concurrent:", 16384/dt, "games/s", eng.last_launch_shape(), flush=True)
