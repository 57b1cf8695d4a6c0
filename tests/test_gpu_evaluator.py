"""GPU parity of the evaluator's baseline opponent: FrozenMCTS over RolloutPolicy (synthesis/src/evaluator.rs:163-534) on the
device vs the CPU oracle (oracle/frozen_mcts.hpp). Bar: every float of every root record bit-identical, identical stream
positions, identical moves and rewards of whole matches."""
import numpy as np
import pytest

from tests.test_gpu_parity import random_positions

pytestmark = pytest.mark.gpu

FROZEN_KEYS = ("child_N", "child_cum", "child_P", "child_sol", "root_stat", "root_sol", "num_nodes", "best_action", "rng_words")


@pytest.fixture(scope="module")
def blob(golden_dir):
    import os

    return np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))


@pytest.fixture(scope="module")
def engine(blob):
    import synthesis_amd as sa

    eng = sa.Engine(concurrent_games=256, max_explores=400)
    eng.load_weights(blob)
    yield eng
    eng.close()


def rollout_cfgs(**kw):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config

    dev = dict(exploration=sa.Exploration.Uct, c=2.0, auto_extend=False, fpu=sa.Fpu.Const, fpu_value=float("inf"))
    orc = dict(exploration=0, c=2.0, auto_extend=0, fpu_value=float("inf"))
    for k, v in kw.items():
        dev[k] = v
        orc[k] = int(v) if isinstance(v, bool) else v
    return sa.MCTSConfig(**dev), parity_mcts_config(**orc)


def assert_frozen_equal(got, ref, what):
    for k in FROZEN_KEYS:
        g, r = np.asarray(got[k]), np.asarray(ref[k])
        if g.dtype == np.float32:
            g, r = g.view(np.uint32), r.view(np.uint32)
        bad = np.nonzero(np.any((g != r).reshape(g.shape[0], -1), axis=1))[0]
        assert bad.size == 0, f"{what}: {k} differs at roots {bad[:8]} ({bad.size} of {g.shape[0]})"


@pytest.mark.parametrize("variant", [dict(), dict(fpu_value=1.0, c=1.3), dict(solve=False), dict(fpu_value=0.25, c=4.0)])
def test_frozen_search_matches_oracle(engine, oracle, variant):
    """Random reachable positions from the empty board to the last few cells (solver-heavy), per-root explores, per-root seeds,
    searches that start in the middle of a generator's stream."""
    dcfg, ocfg = rollout_cfgs(**variant)
    my, op = random_positions(oracle, 600, seed=41, max_moves=62)
    rs = np.random.RandomState(5)
    explores = rs.choice([0, 1, 2, 9, 50, 150, 400], size=my.size).astype(np.int32)
    seeds = rs.randint(0, 2**62, size=my.size).astype(np.uint64)
    words = rs.choice([0, 1, 15, 16, 17, 63, 64, 65, 1000, 123457], size=my.size).astype(np.uint64)
    for sel in (0, 1):
        got = engine.frozen_search(dcfg, seeds, words, my, op, explores, action_selection=sel)
        ref = oracle.c4_frozen_search(ocfg, seeds, words, my, op, explores, action_selection=sel)
        assert_frozen_equal(got, ref, f"{variant} selection {sel}")
    assert engine.last_launch_shape()[0] == 5
    assert (got["root_stat"][:, 0] == explores + 1).all()      # explore_n never stops early
    assert (got["rng_words"] > words).all()                    # the root's own visit always plays out


def test_frozen_search_more_roots_than_lanes(engine, oracle):
    """5,000 roots on an engine with 1,792 tree slabs (28 waves): lanes take further roots in a grid-stride loop"""
    dcfg, ocfg = rollout_cfgs()
    my, op = random_positions(oracle, 250, seed=43, max_moves=50)
    my, op = np.tile(my, 20), np.tile(op, 20)
    seeds = np.arange(my.size, dtype=np.uint64) * np.uint64(7919)
    got = engine.frozen_search(dcfg, seeds, 0, my, op, 60)
    ref = oracle.c4_frozen_search(ocfg, seeds, 0, my, op, 60)
    assert_frozen_equal(got, ref, "grid-stride")


def test_frozen_search_deeper_than_the_engines_self_play_trees(engine, oracle):
    """The baseline ladder goes far beyond self-play's explores (evaluator.rs:24-41: VanillaMCTS800 ... 204800): the call
    re-partitions the node pool, so an engine built for 400-explore self-play runs 20,000-explore baseline searches (fewer at
    a time). Mixed with tiny searches in the same batch."""
    dcfg, ocfg = rollout_cfgs()
    my, op = random_positions(oracle, 40, seed=47, max_moves=30)
    explores = np.array([20000, 3, 7000, 0] * 10, np.int32)
    seeds = np.arange(40, dtype=np.uint64) + np.uint64(900)
    got = engine.frozen_search(dcfg, seeds, 11, my, op, explores)
    ref = oracle.c4_frozen_search(ocfg, seeds, 11, my, op, explores)
    assert_frozen_equal(got, ref, "deep baseline")
    assert got["num_nodes"].max() > 2 * (1 + 9 * 401)   # really beyond the self-play slab


def test_frozen_search_at_the_top_of_the_reference_ladder(engine, oracle):
    """VanillaMCTS204800 (main.rs:71), the deepest baseline the reference evaluates against: 1.8 M node records per tree (21-bit
    ids), millions of stream words per search, on the same small engine."""
    dcfg, ocfg = rollout_cfgs()
    my, op = random_positions(oracle, 6, seed=53, max_moves=12)
    explores = np.array([204800, 51200, 9, 204800, 0, 102400], np.int32)
    seeds = np.arange(6, dtype=np.uint64) + np.uint64(31)
    got = engine.frozen_search(dcfg, seeds, 5, my, op, explores, action_selection=0)
    ref = oracle.c4_frozen_search(ocfg, seeds, 5, my, op, explores, action_selection=0)
    assert_frozen_equal(got, ref, "ladder top")
    assert got["root_stat"][0, 0] == 204801.0 and got["rng_words"].max() > 2_000_000


def test_mcts_vs_mcts_replays_the_oracle(engine, oracle):
    """evaluator.rs:200-228: VanillaMCTS a vs VanillaMCTS b, one StdRng per game shared by both sides for the whole game"""
    from synthesis_amd import match

    dcfg, ocfg = rollout_cfgs()
    n = 40
    for first_explores, second_explores, sel in ((200, 100, 0), (50, 400, 1)):
        rec = {}
        reward, plies = match.play_match(engine, match.vanilla_player(first_explores, action=sel), match.vanilla_player(second_explores, action=sel), n,
                                         seed=100, record=rec)
        for g in range(n):
            r, moves, words = oracle.c4_mcts_vs_mcts(ocfg, 0, first_explores, second_explores, 100 + g, rollout_action=sel)
            assert plies[g] == moves.size and np.array_equal(rec["moves"][g, :moves.size], moves), f"game {g}"
            assert reward[g] == r and rec["rng_words"][g] == words[-1]
        # the same match with the roles swapped (player = second mover, evaluator.rs:38-39)
        r, moves, _ = oracle.c4_mcts_vs_mcts(ocfg, 1, second_explores, first_explores, 100, rollout_action=sel)
        assert np.array_equal(rec["moves"][0, :moves.size], moves) and reward[0] == r


def test_eval_against_rollout_replays_the_oracle(engine, oracle, blob):
    """evaluator.rs:163-198: the network's MCTS::exploit on one side, the vanilla baseline on the other, both colours"""
    import synthesis_amd as sa
    from synthesis_amd import match
    from tests.oracle_lib import parity_mcts_config

    dcfg, ocfg = rollout_cfgs()
    net = match.Player("net", 100, sa.parity_mcts_config())
    n = 24
    for net_moves_first in (True, False):
        rec = {}
        sel = 0 if net_moves_first else 1   # rollout_action: Q (the reference's choice) and NumVisits
        van = match.vanilla_player(150, action=sel)
        a, b = (net, van) if net_moves_first else (van, net)
        reward, plies = match.play_match(engine, a, b, n, seed=7, record=rec)
        for g in range(n):
            r, moves, words = oracle.c4_eval_against_rollout(parity_mcts_config(), 100, blob, ocfg, 0 if net_moves_first else 1,
                                                             150, 7 + g, rollout_action=sel, nn_mode=oracle.ACC_FMA)
            assert plies[g] == moves.size and np.array_equal(rec["moves"][g, :moves.size], moves), f"game {g}"
            assert reward[g] == r and rec["rng_words"][g] == words[-1]


def test_frozen_search_rejects_what_the_reference_panics_on(engine):
    import synthesis_amd as sa

    z = np.zeros(1, np.uint64)
    dcfg, _ = rollout_cfgs()
    with pytest.raises(sa.SynthesisAmdError, match="Uct only"):
        engine.frozen_search(sa.MCTSConfig(exploration=sa.Exploration.PolynomialUct, fpu=sa.Fpu.Const), 0, 0, z, z, 10)
    with pytest.raises(sa.SynthesisAmdError, match="Fpu::Const only"):
        engine.frozen_search(sa.MCTSConfig(exploration=sa.Exploration.Uct, fpu=sa.Fpu.ParentQ), 0, 0, z, z, 10)
    with pytest.raises(sa.SynthesisAmdError, match="needs up to"):
        engine.frozen_search(dcfg, 0, 0, z, z, 300000)   # beyond 21 bits of node index
    with pytest.raises(sa.SynthesisAmdError, match="searchable"):
        engine.frozen_search(dcfg, 0, 0, np.array([3], np.uint64), np.array([1], np.uint64), 10)
    got = engine.frozen_search(dcfg, 0, 0, np.zeros(0, np.uint64), np.zeros(0, np.uint64), 10)
    assert got["best_action"].shape == (0,)


def test_evaluation_round_follows_the_reference_schedule(engine, oracle, blob):
    """evaluator.rs:22-99 for one model: the PGN text (pairings, colours, order, results) equals the one assembled from the
    oracle's three match loops; the older model is a second, different network on the same engine."""
    from synthesis_amd import match
    from tests.oracle_lib import parity_mcts_config
    import synthesis_amd as sa

    old_w = np.random.RandomState(3).normal(0, 0.2, blob.size).astype(np.float32)
    cfg = match.EvaluationConfig(policy_num_explores=80, policy_mcts_cfg=sa.parity_mcts_config(), num_games_against_rollout=2,
                                 rollout_num_explores=(60, 120, 240))
    got = match.evaluation_round(engine, cfg, 4, "model_4.ot", blob, [("model_1.ot", old_w)])
    _, ocfg = rollout_cfgs()
    pcfg = parity_mcts_config()
    want = []
    i = 4 % 3
    for j in range(3):
        if j != i:
            r, _, _ = oracle.c4_mcts_vs_mcts(ocfg, 0, cfg.rollout_num_explores[i], cfg.rollout_num_explores[j], 4, rollout_action=0)
            want.append(match.pgn_records(f"VanillaMCTS{cfg.rollout_num_explores[i]}", f"VanillaMCTS{cfg.rollout_num_explores[j]}", [r]))
    for ex in cfg.rollout_num_explores:
        for sd in range(2):
            r, _, _ = oracle.c4_eval_against_rollout(pcfg, 80, blob, ocfg, 0, ex, sd, rollout_action=0, nn_mode=oracle.ACC_FMA)
            want.append(match.pgn_records("model_4.ot", f"VanillaMCTS{ex}", [r]))
            r, _, _ = oracle.c4_eval_against_rollout(pcfg, 80, blob, ocfg, 1, ex, sd, rollout_action=0, nn_mode=oracle.ACC_FMA)
            want.append(match.pgn_records(f"VanillaMCTS{ex}", "model_4.ot", [r]))
    r, _ = oracle.c4_eval_against_old(pcfg, 80, blob, old_w, nn_mode=oracle.ACC_FMA)
    want.append(match.pgn_records("model_4.ot", "model_1.ot", [r]))
    r, _ = oracle.c4_eval_against_old(pcfg, 80, old_w, blob, nn_mode=oracle.ACC_FMA)
    want.append(match.pgn_records("model_1.ot", "model_4.ot", [r]))
    assert got == "".join(want)
    engine.load_weights(blob)   # leave the shared engine as the other tests expect it
