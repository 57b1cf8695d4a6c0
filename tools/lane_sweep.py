#!/usr/bin/env python3
"""Developer tool (a first argument "f16x2" selects the f16x2 network arithmetic): throughput of the lane-per-tree / producer-consumer kernels. Args: conc:nw[:games[:policy_cache_log2]] ...
(nw = waves per workgroup of the lane kernel, 0 = the engine's own choice, negative = producer/consumer kernel with -nw
virtual waves per tree wave, 208 / 212 = the two-trees-per-lane kernel with 8 / 12 waves, 365 .. 428 = the pool kernel with
nw - 300 trees per wave, 300 = the pool kernel switched off). A first argument "reference" runs the
reference's own self-play configuration (trained checkpoint + Fpu::Func(Normal(1.0, 0.1)); give a policy_cache_log2 too). A first argument "conv" runs Connect4ConvNet (convnet.cuh) instead of Connect4Net; "eval" appended
measures the stand-alone batched Policy::eval of the chosen network on 4M positions as well."""
import os
os.environ["SYN_DEBUG"] = "1"  # developer knobs (SYN_LANES, SYN_PROFILE, ...) are honoured only with SYN_DEBUG=1
import os, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthesis_amd as sa
from bench import make_conv_weights, make_weights
argv = sys.argv[1:]
f16x2 = bool(argv) and argv[0] == "f16x2"      # Connect4Net in the f16x2 arithmetic (syn_set_network_arithmetic)
if f16x2: argv = argv[1:]
conv = bool(argv) and argv[0] == "conv"
if conv: argv = argv[1:]
reference = bool(argv) and argv[0] == "reference"
if reference: argv = ["trained"] + argv[1:]
parentq = bool(argv) and argv[0] == "parentq"   # a non-parity configuration without draws (general kernel instantiation): Fpu::ParentQ
if parentq: argv = ["trained"] + argv[1:]
trained = bool(argv) and argv[0] == "trained"   # the trained Connect4Net checkpoint (deep narrow trees) instead of the fixed-seed init
if trained: argv = argv[1:]
do_eval = bool(argv) and argv[-1] == "eval"
if do_eval: argv = argv[:-1]
blob = make_conv_weights() if conv else make_weights()
if trained: blob = np.load(os.path.join(ROOT, 'tests', 'golden', 'c4net_trained_f32.npy'))
combos = [tuple(map(int, a.split(":"))) for a in argv] or [(65536, 16), (131072, 16), (262144, 16)]
cfg = sa.parity_rollout_config(800, mcts_cfg=sa.reference_selfplay_mcts_config()) if reference else sa.parity_rollout_config(800)
if parentq: cfg = sa.parity_rollout_config(800, mcts_cfg=sa.MCTSConfig(fpu=sa.Fpu.ParentQ))
for c in combos:
    conc, nw = c[0], c[1]
    n = c[2] if len(c) > 2 else 2 * conc
    clog = c[3] if len(c) > 3 else 0
    if len(c) > 4: os.environ["SYN_POOL_FIRE"] = str(c[4])   # pool kernel: leaves that fire a round
    else: os.environ.pop("SYN_POOL_FIRE", None)
    if len(c) > 5: os.environ["SYN_POOL_NW"] = str(c[5])     # pool kernel: waves per workgroup (8 or 12)
    else: os.environ.pop("SYN_POOL_NW", None)
    if len(c) > 6: os.environ["SYN_SCAN_MIN"] = str(c[6])    # lane kernel, Fpu::Func: waiting lanes that trigger a scan iteration
    else: os.environ.pop("SYN_SCAN_MIN", None)
    os.environ.pop("SYN_LANES", None); os.environ.pop("SYN_PC", None); os.environ.pop("SYN_LANES2", None); os.environ.pop("SYN_POOL", None)
    if 365 <= nw <= 428: os.environ["SYN_POOL"] = str(nw - 300)   # the pool kernel with nw - 300 trees per wave (pool_kernel.cuh)
    elif nw == 300: os.environ["SYN_POOL"] = "0"                   # ... switched off (the lane kernel the engine would pick without it)
    elif nw in (208, 212): os.environ["SYN_LANES2"] = str(nw - 200)
    elif nw > 0: os.environ["SYN_LANES"] = str(nw); os.environ["SYN_PC"] = "0"
    elif nw < 0: os.environ["SYN_PC"] = str(-nw)
    eng = sa.Engine(concurrent_games=conc, max_explores=800, policy_cache_log2=clog)
    (eng.load_weights_conv if conv else eng.load_weights)(blob)
    if f16x2: eng.set_network_arithmetic("f16x2")
    eng.selfplay(cfg, 0, 256, outputs=False)
    t0 = time.perf_counter()
    r = eng.selfplay(cfg, 0, n, first_game=conc, outputs=False)
    dt = time.perf_counter() - t0
    hits, misses = eng.last_cache_stats()
    extra = f"  policy cache 2^{clog}: {hits / max(1, hits + misses):.3f} hit rate" if clog else ""
    print(f"concurrent={conc} lanes_nw={nw} shape={eng.last_launch_shape()}: {n} games in {dt:.3f} s = {n / dt:.0f} games/s  (mean plies {r['plies'].mean():.2f}){extra}", flush=True)
    eng.close()
if do_eval:
    eng = sa.Engine(concurrent_games=256, max_explores=16)
    (eng.load_weights_conv if conv else eng.load_weights)(blob)
    if f16x2: eng.set_network_arithmetic("f16x2")
    rng = np.random.default_rng(0)
    n = 1 << 22
    my = rng.integers(0, 1 << 62, n, dtype=np.uint64); op = rng.integers(0, 1 << 62, n, dtype=np.uint64) & ~my
    eng.policy_eval(my[:65536], op[:65536])
    eng.policy_eval(my, op)
    ms = eng.last_kernel_ms()
    ms = ms[0] if isinstance(ms, (tuple, list)) else ms
    print(f"policy_eval {'Connect4ConvNet' if conv else 'Connect4Net'}: {n} positions, kernel {ms:.2f} ms = {n / ms / 1e6:.3f} G evals/s", flush=True)
    eng.close()
