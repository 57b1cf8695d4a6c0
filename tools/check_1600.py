"""Developer tool (GPU): self-play at the reference's default 1,600 explores per move on the lane-per-tree kernel, and the
same 4,096 games on a small engine for comparison."""
import os
os.environ["SYN_DEBUG"] = "1"  # developer knobs (SYN_LANES, SYN_PROFILE, ...) are honoured only with SYN_DEBUG=1
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import synthesis_amd as sa
from bench import make_weights
blob = make_weights()
eng = sa.Engine(concurrent_games=131072, max_explores=1600)
eng.load_weights(blob)
cfg = sa.parity_rollout_config(1600)
t = time.perf_counter(); r = eng.selfplay(cfg, 3, 262144, outputs=False); dt = time.perf_counter() - t
print("1600 explores: 262144 games in %.2f s = %.0f games/s, plies %.2f, shape %s" % (dt, 262144 / dt, r["plies"].mean(), eng.last_launch_shape()))
r2 = eng.selfplay(cfg, 3, 4096, outputs=True, counters=True)
os.environ['SYN_LANES'] = '0'  # row-per-tree kernels
small = sa.Engine(concurrent_games=4096, max_explores=1600); small.load_weights(blob)
r3 = small.selfplay(cfg, 3, 4096, outputs=True, counters=True)
m = np.arange(63)[None, :] < r3["plies"][:, None]
print("row vs lane at 1600 explores identical:", np.array_equal(r2["plies"], r3["plies"]) and np.array_equal(r2["pis"][m], r3["pis"][m]) and r2["counters"] == r3["counters"], eng.last_launch_shape(), small.last_launch_shape())
