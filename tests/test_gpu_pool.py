"""GPU parity of the pool kernel (synthesis_amd/csrc/pool_kernel.cuh): the lane-per-tree kernel's algorithm — synthesis/src/mcts.rs:310-488
(explore / select_best_child / visit / backprop) and synthesis/src/alpha_zero.rs:229-338 (run_game ...) — with the trees UNBOUND from the
lanes: a wave works on a pool of up to 128 trees, a lane whose descent arrives binds the next READY tree in the same iteration, a round
fires on 64 leaves. A schedule must not change a result, so the bar is the lane kernel's: searches, whole self-play games with slot
refill, event counters, every value target, the policy cache, the reference's own Fpu::Func configuration, the trained checkpoint and
the f16x2 arithmetic — all bit-identical to the CPU oracle. The kernel is forced on small engines here (partial pools, trees without a
job, pools of 65 .. 128 trees). Measured slower than the lane kernel on every leg (profiles/r06_pool_unbinding_ab.txt): a debug
launch shape of `make DEBUG_SHAPES=1` builds; these tests skip on the default library."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import assert_search_equal, assert_selfplay_equal, random_positions

pytestmark = pytest.mark.gpu

SHAPE_POOL = 8


@pytest.fixture(scope="module")
def blob(golden_dir):
    return np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))


@pytest.fixture(params=[128, 80, 65])
def pool(monkeypatch, request, debug_shapes):
    monkeypatch.setenv("SYN_DEBUG", "1")  # developer knobs are honoured only with SYN_DEBUG=1
    monkeypatch.setenv("SYN_POOL", str(request.param))
    return request.param


@pytest.mark.parametrize("conc", [1100, 4000])
def test_pool_kernel_matches_oracle(blob, oracle, pool, conc):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    eng = sa.Engine(concurrent_games=conc, max_explores=800)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 300, seed=31, max_moves=60)
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 120)
    assert eng.last_launch_shape()[0] == SHAPE_POOL and eng.last_launch_shape()[2] == 64 * int(os.environ.get("SYN_POOL_NW", "12"))
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 120, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, "pool search")
    for explores in (0, 1, 2):
        got = eng.mcts_search(sa.parity_mcts_config(), my[:70], op[:70], explores)
        ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my[:70], op[:70], explores, nn_mode=oracle.ACC_FMA)
        assert_search_equal(got, ref, f"pool search, {explores} explores")
    n_games = 2500 if conc == 1100 else 9000
    got = eng.selfplay(sa.parity_rollout_config(50), base_seed=77, n_games=n_games, counters=True)
    assert eng.last_launch_shape()[0] == SHAPE_POOL
    ref = oracle.c4_selfplay(parity_rollout_config(50), blob, 77, n_games, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "pool self-play")
    for k in ("explores", "select_levels", "children_scanned", "expansions", "new_nodes", "policy_evals",
              "backprop_levels", "solver_children", "solved_hits", "max_depth"):
        assert got["counters"][k] == ref["counters"][k], k
    assert got["counters"]["games"] == n_games and got["counters"]["moves"] == int(got["plies"].sum())
    for sv, ov in ((dict(value_target=sa.ValueTarget.Z), dict(value_target=0)),
                   (dict(value_target=sa.ValueTarget.QZaverage, value_target_p=0.3), dict(value_target=2, vt_p=0.3)),
                   (dict(value_target=sa.ValueTarget.QtoZ, value_target_from=0.1, value_target_to=0.9),
                    dict(value_target=3, vt_from=0.1, vt_to=0.9)),
                   (dict(stop_games_when_solved=True, action=sa.ActionSelection.Q, random_actions_until=3),
                    dict(stop_games_when_solved=1, action=0, random_actions_until=3))):
        got = eng.selfplay(sa.parity_rollout_config(40, **sv), base_seed=9, n_games=64)
        ref = oracle.c4_selfplay(parity_rollout_config(40, **ov), blob, 9, 64, threads=8, nn_mode=oracle.ACC_FMA)
        assert_selfplay_equal(got, ref, f"pool {sv}")
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=5, n_games=6)
    ref = oracle.c4_selfplay(parity_rollout_config(800), blob, 5, 6, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "pool 800 explores")
    # a single game and a single root: every other tree of the launch has no job
    got = eng.selfplay(sa.parity_rollout_config(100), base_seed=123, n_games=1)
    ref = oracle.c4_selfplay(parity_rollout_config(100), blob, 123, 1, threads=1, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "pool one game")
    # a configuration outside the two folded families is not this kernel's: the engine falls back to the lane kernel by itself
    got = eng.mcts_search(sa.MCTSConfig(fpu=sa.Fpu.ParentQ), my[:64], op[:64], 60)
    assert eng.last_launch_shape()[0] != SHAPE_POOL
    assert_search_equal(got, oracle.c4_mcts_search(parity_mcts_config(fpu=1), blob, my[:64], op[:64], 60, nn_mode=oracle.ACC_FMA), "fall-back")
    eng.close()


@pytest.mark.parametrize("log2", [10, 22])
def test_pool_policy_cache_is_semantics_neutral(blob, oracle, pool, log2):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    if pool == 65:
        pytest.skip("one pool size less for the cache: same code path")
    eng = sa.Engine(concurrent_games=1100, max_explores=800, policy_cache_log2=log2)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 300, seed=31, max_moves=60)
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 120, nn_mode=oracle.ACC_FMA)
    for rep in range(2):  # the second pass finds the first pass's entries
        got = eng.mcts_search(sa.parity_mcts_config(), my, op, 120)
        assert_search_equal(got, ref, f"pool cache 2^{log2} search pass {rep}")
    assert eng.last_launch_shape()[0] == SHAPE_POOL
    hits, misses = eng.last_cache_stats()
    assert hits + misses > 0 and (log2 == 10 or hits > misses)
    ref = oracle.c4_selfplay(parity_rollout_config(50), blob, 77, 2500, threads=8, nn_mode=oracle.ACC_FMA)
    got = eng.selfplay(sa.parity_rollout_config(50), base_seed=77, n_games=2500, counters=True)
    assert_selfplay_equal(got, ref, f"pool cache 2^{log2} self-play")
    hits, misses = eng.last_cache_stats()
    assert hits + misses == got["counters"]["policy_evals"] == ref["counters"]["policy_evals"]
    if log2 == 22:
        assert hits > 0.3 * (hits + misses)
    eng.close()


@pytest.mark.parametrize("scan", [1, 24, 64])
def test_pool_reference_configuration_matches_oracle(blob, oracle, pool, scan, monkeypatch):
    """Fpu::Func(|| Normal(1.0, 0.1)) — the reference's own self-play configuration, study-connect4/src/main.rs:37-49 — on the pool
    kernel: the draws are a function of (game, turn, scan number, slot), never of the schedule — whatever number of waiting lanes
    triggers a scan iteration."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    if pool == 65 and scan != 24:
        pytest.skip("the scan thresholds on two pool sizes")
    monkeypatch.setenv("SYN_POOL_SCAN", str(scan))
    eng = sa.Engine(concurrent_games=1100, max_explores=400)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 120, seed=77, max_moves=50)
    my[0] = 0; op[0] = 0
    scfg, okw = sa.reference_selfplay_mcts_config(), dict(fpu=2, fpu_value=1.0, fpu_std=0.1)
    for explores in (0, 150):
        got = eng.mcts_search(scfg, my, op, explores)
        assert eng.last_launch_shape()[0] == SHAPE_POOL
        ref = oracle.c4_mcts_search(parity_mcts_config(**okw), blob, my, op, explores, nn_mode=oracle.ACC_FMA)
        assert_search_equal(got, ref, f"pool Fpu::Func explores {explores}")
    got = eng.selfplay(sa.parity_rollout_config(60, mcts_cfg=scfg), base_seed=31, n_games=1500, first_game=7, counters=True)
    ref = oracle.c4_selfplay(parity_rollout_config(60, mcts=parity_mcts_config(**okw)), blob, 31, 1500, first_game=7, threads=8,
                             nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "pool self-play with Fpu::Func")
    assert got["counters"]["policy_evals"] == ref["counters"]["policy_evals"]
    eng.close()


@pytest.mark.parametrize("arith", ["f32", "f16x2"])
def test_pool_trained_checkpoint_deep_trees(golden_dir, oracle, pool, arith):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    if pool == 65:
        pytest.skip("two pool sizes are enough for the deep trees")
    trained = np.load(os.path.join(golden_dir, "c4net_trained_f32.npy"))
    mode = oracle.ACC_F16X2 if arith == "f16x2" else oracle.ACC_FMA
    eng = sa.Engine(concurrent_games=1100, max_explores=800)
    eng.set_network_arithmetic(arith)
    eng.load_weights(trained)
    my, op = random_positions(oracle, 64, seed=19, max_moves=40)
    my[0] = 0; op[0] = 0
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 800)
    assert eng.last_launch_shape()[0] == SHAPE_POOL
    ref = oracle.c4_mcts_search(parity_mcts_config(), trained, my, op, 800, nn_mode=mode)
    assert_search_equal(got, ref, f"pool trained weights ({arith}), 800 explores")
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=3, n_games=8, counters=True)
    ref = oracle.c4_selfplay(parity_rollout_config(800), trained, 3, 8, threads=8, nn_mode=mode)
    assert_selfplay_equal(got, ref, f"pool trained weights ({arith}), self-play")
    for k in ("explores", "select_levels", "backprop_levels", "policy_evals", "max_depth"):
        assert got["counters"][k] == ref["counters"][k], k
    # the reference's own configuration on the trained checkpoint with the cache: the bench's `reference_selfplay_config` leg in small
    eng.close()
    eng = sa.Engine(concurrent_games=1100, max_explores=800, policy_cache_log2=20)
    eng.set_network_arithmetic(arith)
    eng.load_weights(trained)
    cfg = sa.parity_rollout_config(300, mcts_cfg=sa.reference_selfplay_mcts_config())
    got = eng.selfplay(cfg, base_seed=11, n_games=40)
    assert eng.last_launch_shape()[0] == SHAPE_POOL
    ref = oracle.c4_selfplay(parity_rollout_config(300, mcts=parity_mcts_config(fpu=2, fpu_value=1.0, fpu_std=0.1)), trained, 11, 40, threads=8,
                             nn_mode=mode)
    assert_selfplay_equal(got, ref, f"pool reference configuration on the trained checkpoint ({arith})")
    eng.close()
