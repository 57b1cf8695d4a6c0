"""GPU parity tests of SYN_NET_ARITH_F16X2: Connect4Net evaluated as pairs of f16 numbers on v_mfma_f32_16x16x32_f16
(synthesis_amd/csrc/f16x2_tile.cuh; include/synthesis_amd.h syn_set_network_arithmetic).

Bars:
  * the network: EXACTLY the oracle's ACC_F16X2 (oracle/nn_f16x2.hpp restates the instruction's accumulation bit for bit) and
    within north_star's 1e-5 — with a 3x margin — of the canonical slimnn-order evaluation (study-connect4/src/policies.rs:28-59,
    slimnn/src/linear.rs:17-25) and of the torch-f64 goldens;
  * everything above the network (searches, whole games): bit-exact against the oracle's MCTS / run_game driven by that same
    arithmetic (synthesis/src/mcts.rs:310-488, alpha_zero.rs:229-338), on every launch shape the arithmetic runs in;
  * the fused kernels and the stand-alone evaluation kernel produce the same bits (a search's priors ARE the stand-alone softmax).
"""
import json
import os

import numpy as np
import pytest

from tests.test_gpu_parity import SELFPLAY_KEYS, assert_search_equal, assert_selfplay_equal, random_positions

pytestmark = pytest.mark.gpu

NN_TOL = 1e-5        # north_star: "policy/value outputs within 1e-5 fp32"
NN_MARGIN = 3.0      # ... held with this margin by the f16x2 arithmetic on the random-init network


@pytest.fixture(scope="module")
def blob(golden_dir):
    return np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))


@pytest.fixture(scope="module")
def trained(golden_dir):
    return np.load(os.path.join(golden_dir, "c4net_trained_f32.npy"))


@pytest.fixture(scope="module")
def engine(blob):
    import synthesis_amd as sa

    eng = sa.Engine(concurrent_games=256, max_explores=800, device=0)
    eng.load_weights(blob)
    eng.set_network_arithmetic("f16x2")
    yield eng
    eng.close()


def test_f16x2_network_is_the_oracles_bits_and_within_tolerance(engine, oracle, blob, golden_dir):
    name, plan = engine.network_arithmetic()
    assert name == "f16x2" and plan is not None and plan["activation_exp"][0] == 8
    my, op = random_positions(oracle, 3000, seed=3)
    my[0] = 0; op[0] = 0
    logits, value = engine.policy_eval(my, op)
    ref_l, ref_v = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_F16X2)
    assert np.array_equal(logits.view(np.uint32), ref_l.view(np.uint32)) and np.array_equal(value.view(np.uint32), ref_v.view(np.uint32))
    sl, sv = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_SLIMNN)
    assert np.abs(logits - sl).max() < NN_TOL / NN_MARGIN and np.abs(value - sv).max() < NN_TOL / NN_MARGIN
    g = json.load(open(os.path.join(golden_dir, "c4net_torch_goldens.json")))
    gl, gv = engine.policy_eval(g["my_bb"], g["op_bb"])
    assert np.abs(gl - np.array(g["logits_f64"])).max() < NN_TOL / NN_MARGIN
    assert np.abs(gv - np.array(g["value_f64"])).max() < NN_TOL / NN_MARGIN
    # ragged sizes and tile boundaries; evaluation contexts (one worker's policy object) run the same arithmetic
    for n in (0, 1, 15, 16, 17, 33, 70, 2999):
        l, v = engine.policy_eval(my[:n], op[:n])
        assert l.shape == (n, 9) and np.array_equal(l, ref_l[:n]) and np.array_equal(v, ref_v[:n])
    ctx = engine.eval_context()
    for n in (1, 700, 3000):
        l, v = ctx.eval(my[:n], op[:n])
        assert np.array_equal(l, ref_l[:n]) and np.array_equal(v, ref_v[:n])
    # a large batch: every tile of the grid-stride loop gives the first tile's answer (the 1,024-thread launch)
    big_l, big_v = engine.policy_eval(np.tile(my[:2048], 160), np.tile(op[:2048], 160))
    assert np.array_equal(big_l.reshape(160, 2048, 9), np.broadcast_to(ref_l[:2048], (160, 2048, 9)))
    assert np.array_equal(big_v.reshape(160, 2048, 3), np.broadcast_to(ref_v[:2048], (160, 2048, 3)))


def test_f16x2_trained_checkpoint_is_as_close_to_f64_as_f32(oracle, trained):
    """Logits of a trained network reach magnitude ~10^2, where 1e-5 absolute is below f32's own rounding: there the bar is
    'no further from the slimnn-order evaluation than the f32 arithmetic, relative to the logits' size', the outcome probabilities
    still inside 1e-5."""
    import synthesis_amd as sa

    my, op = random_positions(oracle, 4000, seed=9)
    eng = sa.Engine(concurrent_games=256, max_explores=64, device=0)
    try:
        eng.load_weights(trained)
        f32_l, f32_v = eng.policy_eval(my, op)
        eng.set_network_arithmetic("f16x2")
        l, v = eng.policy_eval(my, op)
        ref_l, ref_v = oracle.c4net_eval(trained, my, op, mode=oracle.ACC_F16X2)
        assert np.array_equal(l.view(np.uint32), ref_l.view(np.uint32)) and np.array_equal(v.view(np.uint32), ref_v.view(np.uint32))
        sl, sv = oracle.c4net_eval(trained, my, op, mode=oracle.ACC_SLIMNN)
        scale = max(1.0, float(np.abs(sl).max()))
        assert scale > 50.0   # (the fixture really has large logits)
        assert np.abs(l - sl).max() / scale < 2e-6 and np.abs(v - sv).max() < NN_TOL
        assert np.abs(l - sl).max() < 4.0 * max(np.abs(f32_l - sl).max(), 1e-7)
        # switching back restores the f32 bits (and the cache never mixes the two: covered below)
        eng.set_network_arithmetic("f32")
        b_l, b_v = eng.policy_eval(my[:500], op[:500])
        assert np.array_equal(b_l, f32_l[:500]) and np.array_equal(b_v, f32_v[:500])
    finally:
        eng.close()


def test_f16x2_search_and_selfplay_match_the_oracle(engine, oracle, blob):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    my, op = random_positions(oracle, 96, seed=7)
    my[0] = 0; op[0] = 0
    for explores in (0, 1, 37, 200):
        got = engine.mcts_search(sa.parity_mcts_config(), my, op, explores)
        ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, explores, nn_mode=oracle.ACC_F16X2)
        assert_search_equal(got, ref, f"explores={explores}")
    got = engine.mcts_search(sa.parity_mcts_config(), my[:24], op[:24], 800)
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my[:24], op[:24], 800, nn_mode=oracle.ACC_F16X2)
    assert_search_equal(got, ref, "explores=800")
    assert engine.last_launch_shape()[0] == 7   # <= 16 trees per CU: free-running waves (free_kernel.cuh)
    # the root's priors are the stand-alone kernel's softmax inputs: fused == stand-alone, bit for bit
    l, _ = engine.policy_eval(my[:24], op[:24])
    for i in range(24):
        legal = [c for c in range(9) if not ((int(my[i]) | int(op[i])) >> (6 + 7 * c)) & 1]
        assert np.array_equal(got["child_P"][i][legal] > 0, np.ones(len(legal), bool))
    # whole games, refill included (700 games over 256 slots), with the event counters
    got = engine.selfplay(sa.parity_rollout_config(800), base_seed=11, n_games=12, counters=True)
    ref = oracle.c4_selfplay(parity_rollout_config(800), blob, 11, 12, threads=8, nn_mode=oracle.ACC_F16X2)
    assert_selfplay_equal(got, ref, "800 explores")
    for k in ("explores", "select_levels", "children_scanned", "expansions", "new_nodes", "policy_evals", "backprop_levels"):
        assert got["counters"][k] == ref["counters"][k], k
    got = engine.selfplay(sa.parity_rollout_config(40), base_seed=3, n_games=700)
    ref = oracle.c4_selfplay(parity_rollout_config(40), blob, 3, 700, threads=8, nn_mode=oracle.ACC_F16X2)
    assert_selfplay_equal(got, ref, "700 games")
    assert engine.last_launch_shape()[0] == 7


def test_f16x2_free_running_kernel_at_4096_games(oracle, trained):
    """BASELINE configs[1]'s size: 4,096 concurrent games = 16 trees per CU — four free-running waves of four trees on every CU
    (free_kernel.cuh), trained checkpoint (deep trees, uneven descents), a runtime-switched configuration as well: the oracle's games."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    eng = sa.Engine(concurrent_games=4096, max_explores=200, device=0)
    try:
        eng.load_weights(trained)
        eng.set_network_arithmetic("f16x2")
        got = eng.selfplay(sa.parity_rollout_config(60), base_seed=17, n_games=5000)
        assert eng.last_launch_shape() == (7, 256, 256)
        ref = oracle.c4_selfplay(parity_rollout_config(60), trained, 17, 700, threads=8, nn_mode=oracle.ACC_F16X2)
        assert_selfplay_equal({k: got[k][:700] for k in SELFPLAY_KEYS}, ref, "4096 concurrent, games 0..699")
        ref = oracle.c4_selfplay(parity_rollout_config(60), trained, 17, 200, first_game=4700, threads=8, nn_mode=oracle.ACC_F16X2)
        assert_selfplay_equal({k: got[k][4700:4900] for k in SELFPLAY_KEYS}, ref, "4096 concurrent, refilled slots")
        my, op = random_positions(oracle, 120, seed=5)
        got = eng.mcts_search(sa.MCTSConfig(exploration=sa.Exploration.Uct, c=1.5, fpu=sa.Fpu.ParentQ, select_solved_nodes=False), my, op, 150)
        ref = oracle.c4_mcts_search(parity_mcts_config(exploration=0, c=1.5, fpu=1, select_solved_nodes=0), trained, my, op, 150, nn_mode=oracle.ACC_F16X2)
        assert_search_equal(got, ref, "general configuration")
        assert eng.last_launch_shape()[0] == 7
    finally:
        eng.close()


@pytest.mark.parametrize("nw", [4, 8, 12, 16])
def test_f16x2_every_wave_count_plays_the_same_games(oracle, trained, monkeypatch, nw):
    """The lane-per-tree kernel at 4 / 8 / 12 / 16 waves per workgroup, a trained network (deep trees, near-ties), the policy cache
    on and off, the parity family, the reference's own Fpu::Func family and a runtime-switched configuration: identical to the oracle."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    monkeypatch.setenv("SYN_DEBUG", "1")
    monkeypatch.setenv("SYN_LANES", str(nw))
    my, op = random_positions(oracle, 40, seed=21)
    ref_s = oracle.c4_mcts_search(parity_mcts_config(), trained, my, op, 300, nn_mode=oracle.ACC_F16X2)
    ref_g = oracle.c4_selfplay(parity_rollout_config(120), trained, 5, 150, threads=8, nn_mode=oracle.ACC_F16X2)
    for cache_log2 in (0, 12):
        eng = sa.Engine(concurrent_games=64 * nw, max_explores=300, device=0, policy_cache_log2=cache_log2)
        try:
            eng.load_weights(trained)
            eng.set_network_arithmetic("f16x2")
            assert_search_equal(eng.mcts_search(sa.parity_mcts_config(), my, op, 300), ref_s, f"nw={nw} cache={cache_log2}")
            assert_selfplay_equal(eng.selfplay(sa.parity_rollout_config(120), base_seed=5, n_games=150), ref_g, f"nw={nw} cache={cache_log2}")
            shape = eng.last_launch_shape()
            assert shape[0] == 4 and shape[2] == 64 * nw, shape   # the lane-per-tree kernel at that wave count
        finally:
            eng.close()
    if nw in (4, 8):
        eng = sa.Engine(concurrent_games=64 * nw, max_explores=300, device=0)
        try:
            eng.load_weights(trained)
            eng.set_network_arithmetic("f16x2")
            # the reference's own configuration (Fpu::Func(Normal), family 2) and a runtime-switched one (Uct + ParentQ)
            fcfg = dict(fpu=2, fpu_value=1.0, fpu_std=0.1)
            got = eng.mcts_search(sa.parity_mcts_config(fpu=sa.Fpu.Func, fpu_value=1.0, fpu_std=0.1), my, op, 200)
            ref = oracle.c4_mcts_search(parity_mcts_config(**fcfg), trained, my, op, 200, nn_mode=oracle.ACC_F16X2)
            assert_search_equal(got, ref, f"Fpu::Func nw={nw}")
            got = eng.mcts_search(sa.parity_mcts_config(exploration=sa.Exploration.Uct, c=1.5, fpu=sa.Fpu.ParentQ), my, op, 200)
            ref = oracle.c4_mcts_search(parity_mcts_config(exploration=0, c=1.5, fpu=1), trained, my, op, 200, nn_mode=oracle.ACC_F16X2)
            assert_search_equal(got, ref, f"Uct/ParentQ nw={nw}")
        finally:
            eng.close()


def test_f16x2_arithmetic_switches_cleanly(oracle, blob):
    """The choice is explicit and never leaks: the cache is emptied on a switch, the conv network refuses it, engines created
    for more explores than the lane kernels address refuse it, a learner's published weights arrive in the chosen arithmetic."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config

    my, op = random_positions(oracle, 32, seed=33)
    eng = sa.Engine(concurrent_games=256, max_explores=200, device=0, policy_cache_log2=10)
    try:
        eng.load_weights(blob)
        a = eng.mcts_search(sa.parity_mcts_config(), my, op, 200)
        eng.set_network_arithmetic("f16x2")
        b = eng.mcts_search(sa.parity_mcts_config(), my, op, 200)
        eng.set_network_arithmetic("f32")
        c = eng.mcts_search(sa.parity_mcts_config(), my, op, 200)
        assert_search_equal(a, oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 200, nn_mode=oracle.ACC_FMA), "f32 first")
        assert_search_equal(b, oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 200, nn_mode=oracle.ACC_F16X2), "f16x2")
        assert_search_equal(c, a, "f32 again")
        with pytest.raises(sa.SynthesisAmdError):
            eng.set_network_arithmetic(7)
        # weights loaded AFTER the choice are evaluated in it
        eng.set_network_arithmetic("f16x2")
        blob2 = (blob * np.float32(1.25)).astype(np.float32)
        eng.load_weights(blob2)
        l, v = eng.policy_eval(my, op)
        rl, rv = oracle.c4net_eval(blob2, my, op, mode=oracle.ACC_F16X2)
        assert np.array_equal(l, rl) and np.array_equal(v, rv)
        # non-finite parameters have no plan
        bad = blob.copy(); bad[5] = np.inf
        with pytest.raises(sa.SynthesisAmdError):
            eng.load_weights(bad)
        # ... and a refused load changes nothing: the engine still holds blob2 in the f16x2 arithmetic, for policy_eval, an
        # evaluation context and the searches alike (nothing may run on a stale or missing image)
        assert eng.network_arithmetic()[0] == "f16x2"
        l, v = eng.policy_eval(my, op)
        assert np.array_equal(l, rl) and np.array_equal(v, rv)
        ctx = eng.eval_context()
        cl, cv = ctx.eval(my, op)
        ctx.close()
        assert np.array_equal(cl, rl) and np.array_equal(cv, rv)
        got = eng.mcts_search(sa.parity_mcts_config(), my[:8], op[:8], 60)
        assert_search_equal(got, oracle.c4_mcts_search(parity_mcts_config(), blob2, my[:8], op[:8], 60, nn_mode=oracle.ACC_F16X2), "after a refused load")
        # the same from the f32 side: the switch itself is refused for parameters without a plan and the engine stays in f32
        eng.set_network_arithmetic("f32")
        wide = blob.copy(); wide[7] = np.float32(3e38)
        eng.load_weights(wide)
        with pytest.raises(sa.SynthesisAmdError):
            eng.set_network_arithmetic("f16x2")
        assert eng.network_arithmetic()[0] == "f32"
        eng.load_weights(blob2)
        eng.set_network_arithmetic("f16x2")
        # the learner's hand-off (syn_trainer_publish_weights: model_{i+1}.ot of alpha_zero.rs:97,194) arrives in the chosen arithmetic:
        # a few optimiser steps, publish, and the engine evaluates the TRAINED parameters as the oracle's ACC_F16X2 does
        eng.load_weights(blob)
        eng.trainer_init(blob)
        rs = np.random.RandomState(4)
        tpi = rs.dirichlet(np.ones(9), 32).astype(np.float32); tv = rs.dirichlet(np.ones(3), 32).astype(np.float32)
        for _ in range(5):
            eng.train_step(my, op, tpi, tv, 1e-2)
        eng.trainer_publish_weights()
        trained_now = eng.trainer_state()["weights"]
        assert np.abs(trained_now - blob).max() > 1e-3
        l, v = eng.policy_eval(my, op)
        rl, rv = oracle.c4net_eval(trained_now, my, op, mode=oracle.ACC_F16X2)
        assert np.array_equal(l, rl) and np.array_equal(v, rv)
        got = eng.mcts_search(sa.parity_mcts_config(), my[:8], op[:8], 100)
        assert_search_equal(got, oracle.c4_mcts_search(parity_mcts_config(), trained_now, my[:8], op[:8], 100, nn_mode=oracle.ACC_F16X2), "published")
    finally:
        eng.close()
    big = sa.Engine(concurrent_games=16, max_explores=8000, device=0)
    try:
        big.load_weights(blob)
        with pytest.raises(sa.SynthesisAmdError) as e:
            big.set_network_arithmetic("f16x2")
        assert e.value.code == -5
    finally:
        big.close()
