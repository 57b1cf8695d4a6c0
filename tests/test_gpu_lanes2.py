"""GPU parity of the two-trees-per-lane kernel (synthesis_amd/csrc/lane2_kernel.cuh), a debug launch shape
(`make DEBUG_SHAPES=1`; the tests skip on the default library).

It is the lane-per-tree kernel's algorithm — synthesis/src/mcts.rs:310-488 (explore / select_best_child / visit / backprop) and
synthesis/src/alpha_zero.rs:229-338 (run_game ...) — with a different schedule: every lane owns two trees and descends one while
the other waits for the network. A schedule must not change a result, so the bar is the lane kernel's: searches, every MCTS
configuration family, whole self-play games with slot refill, event counters, value targets, the policy cache, the device noise
samplers, the trained checkpoint and the conv network — all bit-identical to the CPU oracle. The kernel is forced on small engines
here (partial waves, contexts without a job); tests/test_gpu_bench_shape.py holds it to the oracle at the size bench.py times."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import assert_search_equal, assert_selfplay_equal, random_positions

pytestmark = pytest.mark.gpu

SHAPE_LANES2 = 6


@pytest.fixture(scope="module")
def blob(golden_dir):
    return np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))


@pytest.fixture(params=[8, 12])
def lanes2(monkeypatch, request, debug_shapes):
    monkeypatch.setenv("SYN_DEBUG", "1")  # developer knobs are honoured only with SYN_DEBUG=1
    monkeypatch.setenv("SYN_LANES2", str(request.param))
    return request.param


@pytest.mark.parametrize("conc", [1100, 3000])
def test_lanes2_kernel_matches_oracle(blob, oracle, lanes2, conc):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    eng = sa.Engine(concurrent_games=conc, max_explores=800)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 300, seed=31, max_moves=60)
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 120)
    assert eng.last_launch_shape()[0] == SHAPE_LANES2 and eng.last_launch_shape()[2] == 64 * lanes2
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 120, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, "lanes2 search")
    for explores in (0, 1, 2):
        got = eng.mcts_search(sa.parity_mcts_config(), my[:70], op[:70], explores)
        ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my[:70], op[:70], explores, nn_mode=oracle.ACC_FMA)
        assert_search_equal(got, ref, f"lanes2 search, {explores} explores")
    for kw in (dict(exploration=0, c=1.4), dict(fpu=1), dict(solve=0), dict(correct_values_on_solve=0),
               dict(select_solved_nodes=0), dict(auto_extend=0), dict(noise=1, noise_weight=0.25)):
        ocfg = parity_mcts_config(**kw)
        scfg = sa.MCTSConfig(exploration=sa.Exploration(ocfg.exploration), c=ocfg.c, solve=bool(ocfg.solve),
                             correct_values_on_solve=bool(ocfg.correct_values_on_solve),
                             select_solved_nodes=bool(ocfg.select_solved_nodes), auto_extend=bool(ocfg.auto_extend),
                             fpu=sa.Fpu(ocfg.fpu), fpu_value=ocfg.fpu_value,
                             root_policy_noise=sa.PolicyNoise(ocfg.noise), noise_weight=ocfg.noise_weight)
        got = eng.mcts_search(scfg, my[:120], op[:120], 90, action_selection=0)
        ref = oracle.c4_mcts_search(ocfg, blob, my[:120], op[:120], 90, action_selection=0, nn_mode=oracle.ACC_FMA)
        assert_search_equal(got, ref, f"lanes2 variant {kw}")
    n_games = 2500 if conc == 1100 else 7000
    got = eng.selfplay(sa.parity_rollout_config(50), base_seed=77, n_games=n_games, counters=True)
    assert eng.last_launch_shape()[0] == SHAPE_LANES2
    ref = oracle.c4_selfplay(parity_rollout_config(50), blob, 77, n_games, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "lanes2 self-play")
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
        assert_selfplay_equal(got, ref, f"lanes2 {sv}")
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=5, n_games=6)
    ref = oracle.c4_selfplay(parity_rollout_config(800), blob, 5, 6, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "lanes2 800 explores")
    # a single game and a single root: every other context of the launch has no job
    got = eng.selfplay(sa.parity_rollout_config(100), base_seed=123, n_games=1)
    ref = oracle.c4_selfplay(parity_rollout_config(100), blob, 123, 1, threads=1, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "lanes2 one game")
    eng.close()


@pytest.mark.parametrize("log2", [10, 22])
def test_lanes2_policy_cache_is_semantics_neutral(blob, oracle, lanes2, log2):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    eng = sa.Engine(concurrent_games=1100, max_explores=800, policy_cache_log2=log2)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 300, seed=31, max_moves=60)
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 120, nn_mode=oracle.ACC_FMA)
    for rep in range(2):  # the second pass finds the first pass's entries
        got = eng.mcts_search(sa.parity_mcts_config(), my, op, 120)
        assert_search_equal(got, ref, f"lanes2 cache 2^{log2} search pass {rep}")
    assert eng.last_launch_shape()[0] == SHAPE_LANES2
    hits, misses = eng.last_cache_stats()
    assert hits + misses > 0 and (log2 == 10 or hits > misses)
    ref = oracle.c4_selfplay(parity_rollout_config(50), blob, 77, 2500, threads=8, nn_mode=oracle.ACC_FMA)
    got = eng.selfplay(sa.parity_rollout_config(50), base_seed=77, n_games=2500, counters=True)
    assert_selfplay_equal(got, ref, f"lanes2 cache 2^{log2} self-play")
    hits, misses = eng.last_cache_stats()
    assert hits + misses == got["counters"]["policy_evals"] == ref["counters"]["policy_evals"]
    if log2 == 22:
        assert hits > 0.3 * (hits + misses)
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=5, n_games=6)
    ref = oracle.c4_selfplay(parity_rollout_config(800), blob, 5, 6, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, f"lanes2 cache 2^{log2} 800 explores")
    eng.close()


def test_lanes2_noise_samplers_match_oracle(blob, oracle, lanes2):
    """Fpu::Func(Normal) — the reference's own self-play configuration, study-connect4/src/main.rs:37-49 — and
    PolicyNoise::Dirichlet (mcts.rs:241-256): the draws are a function of (game, turn, draw number), never of the schedule."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    eng = sa.Engine(concurrent_games=1100, max_explores=400)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 120, seed=77, max_moves=50)
    my[0] = 0; op[0] = 0
    variants = [
        (sa.reference_selfplay_mcts_config(), dict(fpu=2, fpu_value=1.0, fpu_std=0.1)),
        (sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=1.0, noise_weight=0.5, fpu=sa.Fpu.Func, fpu_value=1.0,
                       fpu_std=0.1), dict(noise=2, noise_alpha=1.0, noise_weight=0.5, fpu=2, fpu_value=1.0, fpu_std=0.1)),
        (sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=2.5, noise_weight=0.25, auto_extend=False),
         dict(noise=2, noise_alpha=2.5, noise_weight=0.25, auto_extend=0)),
    ]
    for scfg, okw in variants:
        for explores in (0, 150):
            got = eng.mcts_search(scfg, my, op, explores)
            assert eng.last_launch_shape()[0] == SHAPE_LANES2
            ref = oracle.c4_mcts_search(parity_mcts_config(**okw), blob, my, op, explores, nn_mode=oracle.ACC_FMA)
            assert_search_equal(got, ref, f"lanes2 noise variant {okw} explores {explores}")
    for scfg, okw in variants[:2]:
        got = eng.selfplay(sa.parity_rollout_config(60, mcts_cfg=scfg), base_seed=31, n_games=300, first_game=7, counters=True)
        ref = oracle.c4_selfplay(parity_rollout_config(60, mcts=parity_mcts_config(**okw)), blob, 31, 300, first_game=7, threads=8,
                                 nn_mode=oracle.ACC_FMA)
        assert_selfplay_equal(got, ref, f"lanes2 self-play with {okw}")
        assert got["counters"]["policy_evals"] == ref["counters"]["policy_evals"]
    eng.close()


def test_lanes2_trained_checkpoint_deep_trees(golden_dir, oracle, lanes2):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    trained = np.load(os.path.join(golden_dir, "c4net_trained_f32.npy"))
    eng = sa.Engine(concurrent_games=1100, max_explores=800)
    eng.load_weights(trained)
    my, op = random_positions(oracle, 64, seed=19, max_moves=40)
    my[0] = 0; op[0] = 0
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 800)
    assert eng.last_launch_shape()[0] == SHAPE_LANES2
    ref = oracle.c4_mcts_search(parity_mcts_config(), trained, my, op, 800, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, "lanes2 trained weights, 800 explores")
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=3, n_games=8, counters=True)
    ref = oracle.c4_selfplay(parity_rollout_config(800), trained, 3, 8, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "lanes2 trained weights, self-play")
    for k in ("explores", "select_levels", "backprop_levels", "policy_evals", "max_depth"):
        assert got["counters"][k] == ref["counters"][k], k
    eng.close()


def test_lanes2_conv_policy(oracle, lanes2):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config
    from tests.test_gpu_convnet import conv_blob

    cblob = conv_blob()
    eng = sa.Engine(concurrent_games=1100, max_explores=800)
    eng.load_weights_conv(cblob)
    my, op = random_positions(oracle, 200, seed=17, max_moves=60)
    my[0] = 0; op[0] = 0
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 150)
    assert eng.last_launch_shape()[0] == SHAPE_LANES2
    ref = oracle.c4_mcts_search(parity_mcts_config(), cblob, my, op, 150, nn_mode=oracle.ACC_FMA, net="conv")
    assert_search_equal(got, ref, "lanes2 conv search")
    got = eng.selfplay(sa.parity_rollout_config(50), base_seed=11, n_games=2300, counters=True)
    ref = oracle.c4_selfplay(parity_rollout_config(50), cblob, 11, 2300, threads=8, nn_mode=oracle.ACC_FMA, net="conv")
    assert_selfplay_equal(got, ref, "lanes2 conv self-play")
    assert got["counters"]["policy_evals"] == ref["counters"]["policy_evals"]
    eng.close()
