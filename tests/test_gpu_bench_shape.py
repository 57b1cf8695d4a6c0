"""GPU tests of the headline launch shape AT ITS OWN SIZE and of the multi-rank / multi-engine plumbing.

* bench shape: 262,144 concurrent games (1,024 per CU: the 16-wave lane-per-tree launch of bench.py) and 196,608 (12 waves),
  the engine's own launch selection (no developer knob), at least one full wave of games plus refill, replay outputs on —
  sampled games equal the oracle game for game and every game's length, last position and result equal the 4,096-slot
  engine's (row-per-tree kernel), i.e. the 60 GB / 45 GB pools (14-bit block ids, > 4 GB offsets) compute the same
  games as the small engines the other parity tests use (alpha_zero.rs:120-169: the fan-out over workers).
* two engines on one device driven from two host threads (one handle per host thread: SURVEY §8b threading row).
* `python bench.py --gpus 2 --dist-backend gloo` started bare: two ranks (both on GPU 0), one JSON line with n_gpus 2.
"""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def blob(golden_dir):
    return np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))


# (round 5: Connect4Net's compile-time-folded families run 12 waves on at most 768 trees per CU whatever the pool holds — the headline
#  engine of 262,144 slots plays on 196,608 of them; the conv network keeps 16 waves)
@pytest.mark.parametrize("net,conc,expect", [("mlp", 262144, (4, 256, 768)), ("mlp", 196608, (4, 256, 768)),
                                             ("conv", 262144, (4, 256, 1024)),
                                             # every other leg bench.py times on the headline engine, at its own shape (round 6):
                                             ("mlp-f16x2", 262144, (4, 256, 768)), ("mlp-trained", 262144, (4, 256, 768)),
                                             ("mlp-f16x2-trained", 262144, (4, 256, 768))])
def test_bench_shape_matches_oracle_and_small_engine(blob, oracle, golden_dir, monkeypatch, net, conc, expect):
    """The configurations bench.py times, at their own size and with the launch shape the engine picks by itself: Connect4Net (the
    headline; `with_trained_weights`: the trained checkpoint), the same two in the f16x2 arithmetic (`with_f16x2_network`: POLICY 3 of
    the lane-per-tree kernel against the oracle's ACC_F16X2) and the conv policy/value network (the `with_conv_policy` leg; layers
    slimnn/src/conv.rs:45-85, linear.rs:17-25)."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_rollout_config
    from tests.test_gpu_parity import assert_selfplay_equal

    for k in ("SYN_DEBUG", "SYN_LANES", "SYN_LANES2", "SYN_QUADS", "SYN_LANE_THRESH", "SYN_PROFILE", "SYN_PC", "SYN_FREE", "SYN_POOL"):
        monkeypatch.delenv(k, raising=False)
    f16x2, trained = "f16x2" in net, "trained" in net
    onet = "conv" if net == "conv" else "mlp"
    if net == "conv":
        from tests.test_gpu_convnet import conv_blob
        weights = conv_blob()
    elif trained:
        weights = np.load(os.path.join(golden_dir, "c4net_trained_f32.npy"))
    else:
        weights = blob
    nn_mode = oracle.ACC_F16X2 if f16x2 else oracle.ACC_FMA

    def load(e):
        if net == "conv":
            e.load_weights_conv(weights)
        else:
            if f16x2:
                e.set_network_arithmetic("f16x2")
            e.load_weights(weights)

    n_games, seed = conc + 12288, 20260
    cfg = sa.parity_rollout_config(800)
    big = sa.Engine(concurrent_games=conc, max_explores=800, device=0)
    load(big)
    got = big.selfplay(cfg, base_seed=seed, n_games=n_games)   # counters off: the very kernel instantiation bench.py times
    shape, grid, threads = big.last_launch_shape()
    assert (shape, grid, threads) == expect, "the launch shape bench.py's configuration takes by default"
    big.close()
    assert got["plies"].min() >= 7 and got["plies"].max() <= 63

    # (1) game for game against the oracle: blocks of games from the first wave, the last slots of the pool (highest
    # slab offsets: > 40 GB into the pool) and the refill tail
    played = grid * threads    # tree slots the launch plays on
    for first in (0, 65536 + 5, played - 4, conc - 8, conc, n_games - 8):
        ref = oracle.c4_selfplay(parity_rollout_config(800), weights, seed, 8, first_game=first, threads=8, nn_mode=nn_mode, net=onet)
        sub = {k: got[k][first:first + 8] for k in ("plies", "states_bb", "pis", "vs", "actions", "root_nodes", "final_kind")}
        assert_selfplay_equal(sub, ref, f"bench shape ({net}), games {first}..{first + 7}")

    # (2) every game against a 4,096-slot engine (Connect4Net f32: row-per-tree kernel, another node layout; f16x2: the free-running
    # four-trees-per-wave kernel; conv: the 4-wave lane-per-tree shape): same lengths, same last positions and same final results
    small = sa.Engine(concurrent_games=4096, max_explores=800, device=0)
    load(small)
    ref = small.selfplay(cfg, base_seed=seed, n_games=n_games)
    assert small.last_launch_shape()[0] in ((4,) if net == "conv" else (7,) if f16x2 else (1, 2))
    small.close()
    assert np.array_equal(got["plies"], ref["plies"])
    assert np.array_equal(got["final_kind"], ref["final_kind"])
    last = got["plies"] - 1
    idx = np.arange(n_games)
    assert np.array_equal(got["states_bb"][idx, last], ref["states_bb"][idx, last])
    assert np.array_equal(got["actions"][idx, last], ref["actions"][idx, last])
    assert np.array_equal(got["root_nodes"][idx, last], ref["root_nodes"][idx, last])


def test_reference_configuration_leg_matches_oracle_at_its_bench_shape(oracle, golden_dir, monkeypatch):
    """bench.py's `reference_selfplay_config` leg as it is timed: the trained checkpoint, PolicyWithCache with 2^28 entries, the
    reference's own mcts_cfg (study-connect4/src/main.rs:37-49: Fpu::Func(|| Normal(1.0, 0.1)), alpha_zero.rs:197-198 for the cache),
    a 262,144-slot engine, 800 explores, the engine's own launch selection (the compile-time family-2 instantiation at 12 waves) —
    sampled games of the first wave, of the pool's last slots and of the refill equal the oracle's game for game, draws included."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config
    from tests.test_gpu_parity import assert_selfplay_equal

    for k in ("SYN_DEBUG", "SYN_LANES", "SYN_LANES2", "SYN_QUADS", "SYN_LANE_THRESH", "SYN_PROFILE", "SYN_PC"):
        monkeypatch.delenv(k, raising=False)
    weights = np.load(os.path.join(golden_dir, "c4net_trained_f32.npy"))
    conc, seed = 262144, 4242
    n_games = conc + 8192
    cfg = sa.parity_rollout_config(800, mcts_cfg=sa.reference_selfplay_mcts_config())
    eng = sa.Engine(concurrent_games=conc, max_explores=800, device=0, policy_cache_log2=28)
    eng.load_weights(weights)
    got = eng.selfplay(cfg, base_seed=seed, n_games=n_games)
    assert eng.last_launch_shape() == (4, 256, 768), "the launch shape this configuration takes by default"
    hits, misses = eng.last_cache_stats()
    eng.close()
    assert hits > 0.5 * (hits + misses)
    ocfg = parity_rollout_config(800, mcts=parity_mcts_config(fpu=2, fpu_value=1.0, fpu_std=0.1))
    for first in (0, 100000 + 3, 196608 - 4, conc - 8, conc, n_games - 8):
        ref = oracle.c4_selfplay(ocfg, weights, seed, 8, first_game=first, threads=8, nn_mode=oracle.ACC_FMA)
        sub = {k: got[k][first:first + 8] for k in ("plies", "states_bb", "pis", "vs", "actions", "root_nodes", "final_kind")}
        assert_selfplay_equal(sub, ref, f"reference configuration at its bench shape, games {first}..{first + 7}")


def test_policy_cache_at_bench_scale_changes_no_game(blob):
    """PolicyWithCache at the bench's own size: 262,144 concurrent games hammering one lock-free table (2^24 entries: heavily
    contended, torn and overwritten entries must read as misses) play exactly the games of the uncached run."""
    import synthesis_amd as sa

    cfg = sa.parity_rollout_config(800)
    n = 262144 + 4096
    plain = sa.Engine(concurrent_games=262144, max_explores=800, device=0)
    plain.load_weights(blob)
    a = plain.selfplay(cfg, base_seed=99, n_games=n, outputs=False)
    plain.close()
    cached = sa.Engine(concurrent_games=262144, max_explores=800, device=0, policy_cache_log2=24)
    cached.load_weights(blob)
    b = cached.selfplay(cfg, base_seed=99, n_games=n, outputs=False)
    hits, misses = cached.last_cache_stats()
    assert cached.last_launch_shape() == (4, 256, 768)   # with the cache on: 12 waves, 768 of the 1,024 slots per CU
    cached.close()
    assert np.array_equal(a["plies"], b["plies"])
    assert hits > 0.2 * (hits + misses)


def test_two_engines_on_one_device_from_two_threads(blob, oracle):
    """One handle per host thread: two engines on cuda:0, each driven by its own thread at the same time
    (ctypes releases the GIL during the call), play disjoint game ranges; both equal the oracle's games."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_rollout_config
    from tests.test_gpu_parity import assert_selfplay_equal

    cfg = sa.parity_rollout_config(200)
    out, errs = {}, []

    def worker(r):
        try:
            eng = sa.Engine(concurrent_games=512, max_explores=200, device=0)
            eng.load_weights(blob)
            first, count = sa.shard_games(1500, r, 2)
            out[r] = (first, count, eng.selfplay(cfg, base_seed=3, n_games=count, first_game=first))
            eng.close()
        except Exception as e:  # surfaced below
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert out[0][0] == 0 and out[1][0] == out[0][1] and out[0][1] + out[1][1] == 1500
    for r in range(2):
        first, count, got = out[r]
        ref = oracle.c4_selfplay(parity_rollout_config(200), blob, 3, count, first_game=first, threads=8, nn_mode=oracle.ACC_FMA)
        assert_selfplay_equal(got, ref, f"thread {r}")


def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: the bench starts both ranks itself; with --dist-backend gloo both share
    GPU 0 (dry run of the N > 1 path on a 1-GPU box). One JSON line, n_gpus 2, both ranks' games counted."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.check_output(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0",
         "--concurrent", "16384", "--games-per-step", "16384", "--explores", "100", "--no-4096", "--no-policy-cache",
         "--no-cpu-baseline"], env=env, stderr=subprocess.DEVNULL, timeout=900).decode()
    lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 1
    assert line["config"]["games_per_step_per_gpu"] == 16384
    # BASELINE configs[4]'s loop ran on both ranks: self-play everywhere, the learner on rank 0, one weight broadcast per iteration
    ll = line["learner_loop"]
    assert ll["ranks"] == 2 and ll["games_per_iteration"] == 16384 and ll["optimiser_steps"] > 0 and ll["seconds"]["broadcast"] >= 0
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 2 * 16384) < 1.0   # value = games of BOTH ranks / time
    assert 7 <= line["plies_per_game"] <= 63


def test_rccl_process_group_at_world_size_1():
    """What of the N > 1 plumbing a one-GPU box can run over RCCL itself (backend "nccl"): bench.py's process group with a device id,
    barrier, the MAX / SUM reductions of dist_util.reduce_scalars on the device, an all-reduce of the learner's 30,492-float gradient
    message, teardown (tools/check_rccl_world1.py, in a child process: a process group belongs to a process)."""
    from tests.conftest import free_port

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rccl_world1.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "rccl world-1 ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_engine_on_a_device_that_is_not_there_fails_loudly(blob):
    """One rank per GPU: a rank whose GPU is not visible must fail at syn_engine_create with SYN_ERR_NO_DEVICE and a message that
    names the device — never fall back to GPU 0 (two ranks would silently share a GPU and the scaling curve would lie)."""
    import torch

    import synthesis_amd as sa

    n = torch.cuda.device_count()
    for dev in (n, n + 6, -1):
        with pytest.raises(sa.SynthesisAmdError) as e:
            sa.Engine(concurrent_games=64, max_explores=16, device=dev)
        assert e.value.code == -2 and f"device {dev}" in str(e.value) and f"{n} visible" in str(e.value)
    # the last visible device is a valid home for an engine, and an engine says which device it lives on
    eng = sa.Engine(concurrent_games=64, max_explores=16, device=n - 1)
    assert eng.device == n - 1
    eng.load_weights(blob)
    l, v = eng.policy_eval(np.zeros(1, np.uint64), np.zeros(1, np.uint64))
    assert np.isfinite(l).all() and abs(float(v.sum()) - 1.0) < 1e-5
    eng.close()


def test_progress_and_cancel_from_another_thread(blob, oracle):
    """One self-play launch is a single kernel of seconds to tens of seconds: syn_progress reports jobs started / games finished
    while it runs, syn_cancel stops the hand-out of new games — the call returns SYN_ERR_CANCELLED after the started games have
    finished, their outputs identical to the oracle's, plies == 0 for games that never started."""
    import time

    import synthesis_amd as sa
    from tests.oracle_lib import parity_rollout_config
    from tests.test_gpu_parity import assert_selfplay_equal

    eng = sa.Engine(concurrent_games=4096, max_explores=200, device=0)
    eng.load_weights(blob)
    assert eng.progress() == (0, 0)   # nothing launched yet: zeros, not whatever the allocation held
    n = 400000   # ~10 s of work at this size if left alone
    box = {}

    def run():
        box["r"] = eng.selfplay(sa.parity_rollout_config(200), base_seed=5, n_games=n)

    t = threading.Thread(target=run)
    t0 = time.perf_counter()
    t.start()
    seen = []
    while time.perf_counter() - t0 < 60:
        started, finished = eng.progress()
        seen.append((started, finished))
        if finished >= 6000:
            break
        time.sleep(0.02)
    eng.cancel()
    t.join(timeout=120)
    dt = time.perf_counter() - t0
    assert not t.is_alive() and "r" in box
    r = box["r"]
    assert r.get("cancelled") is True and dt < 30
    assert all(b[0] >= a[0] and b[1] >= a[1] for a, b in zip(seen, seen[1:])) and seen[-1][0] >= seen[-1][1] >= 6000
    played = np.nonzero(r["plies"])[0]
    assert 6000 <= played.size < n and played.max() < seen[-1][0] + 4096 + 16   # nothing started after the cancel (+ slots in flight)
    started_after, finished_after = eng.progress()
    assert finished_after == played.size and started_after <= played.size <= started_after + 4096 + 16
    first = int(played[100])
    block = [g for g in range(first, first + 6) if r["plies"][g] > 0]
    ref = oracle.c4_selfplay(parity_rollout_config(200), blob, 5, 6, first_game=first, threads=6, nn_mode=oracle.ACC_FMA)
    sub = {k: r[k][first:first + 6] for k in ("plies", "states_bb", "pis", "vs", "actions", "root_nodes", "final_kind")}
    if len(block) == 6:
        assert_selfplay_equal(sub, ref, "cancelled run, finished games")
    # the engine is usable afterwards
    r2 = eng.selfplay(sa.parity_rollout_config(200), base_seed=5, n_games=64)
    assert "cancelled" not in r2 and (r2["plies"] > 0).all()
    # nothing in flight: there is nothing to cancel, and the next call is not affected by the refused one
    with pytest.raises(sa.SynthesisAmdError):
        eng.cancel()
    r3 = eng.selfplay(sa.parity_rollout_config(200), base_seed=5, n_games=64)
    assert "cancelled" not in r3 and np.array_equal(r3["plies"], r2["plies"])

    # the same bracket around a search: roots that were never handed out come back all zero and the call says so
    from tests.oracle_lib import parity_mcts_config
    from tests.test_gpu_parity import assert_search_equal, random_positions
    my, op = random_positions(oracle, 512, seed=3, max_moves=30)
    reps = 600
    my_all, op_all = np.tile(my, reps), np.tile(op, reps)
    box.clear()

    def search():
        box["s"] = eng.mcts_search(sa.parity_mcts_config(), my_all, op_all, 200)

    t = threading.Thread(target=search)
    t0 = time.perf_counter()
    t.start()
    while time.perf_counter() - t0 < 60:
        if eng.progress()[0] >= 8192:
            break
        time.sleep(0.01)
    eng.cancel()
    t.join(timeout=120)
    assert not t.is_alive() and "s" in box
    sres = box["s"]
    done = sres["num_nodes"] > 0
    assert sres.get("cancelled") is True and 8192 <= int(done.sum()) < my_all.size
    assert not sres["child_N"][~done].any() and not sres["target_pi"][~done].any()
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my[:64], op[:64], 200, nn_mode=oracle.ACC_FMA)
    assert done[:64].all()
    assert_search_equal({k: v[:64] for k, v in sres.items() if k != "cancelled"}, ref, "cancelled search, searched roots")
    after = eng.mcts_search(sa.parity_mcts_config(), my[:64], op[:64], 200)
    assert "cancelled" not in after
    assert_search_equal(after, ref, "search after a cancelled one")
    eng.close()
