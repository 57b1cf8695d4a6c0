"""CPU checks of the oracle's restatement of the evaluator's baseline tree (oracle/frozen_mcts.hpp, evaluator.rs:230-534).
The reference holds no test for FrozenMCTS, so these pin its documented behaviour by hand-derivable cases and invariants:
parity of this one class is "unpinned" against reference outputs."""
import numpy as np

from tests.oracle_lib import parity_mcts_config


def cfg(**kw):
    base = dict(exploration=0, c=2.0, auto_extend=0, fpu_value=float("inf"))
    base.update(kw)
    return parity_mcts_config(**base)


def position(oracle, moves):
    r = oracle.c4_play(moves)
    assert not np.any(r["over"])
    return np.array([r["my_bb"]], np.uint64), np.array([r["op_bb"]], np.uint64)


def test_root_visit_priors_and_counts(oracle):
    """0 explores = with_capacity's own visit: nine unvisited children with uniform priors (RolloutPolicy's logits are zero),
    root visited once with the playout's value, one playout drawn from the generator."""
    z = np.zeros(1, np.uint64)
    r = oracle.c4_frozen_search(cfg(), 5, 0, z, z, 0)
    assert r["num_nodes"][0] == 10 and r["root_stat"][0, 0] == 1.0 and r["root_stat"][0, 1] in (-1.0, 0.0, 1.0)
    assert np.array_equal(r["child_P"][0], np.full(9, np.float32(1.0) / np.float32(9.0)))
    assert not r["child_N"].any() and r["rng_words"][0] >= 7
    # fpu = inf: the first nine explores visit each child once, in action order (ties keep the first maximum)
    r = oracle.c4_frozen_search(cfg(), 5, 0, z, z, 9)
    assert np.array_equal(r["child_N"][0], np.ones(9)) and r["root_stat"][0, 0] == 10.0 and r["num_nodes"][0] == 10 + 81


def test_mate_in_one_is_proven_and_played(oracle):
    """Three in a column for the side to move: the root's own visit creates the terminal child Lose(0), backprop marks the
    root Win(0) with value -cum + (visits + 1) (evaluator.rs:500-503); best_action takes the +inf child."""
    my, op = position(oracle, [0, 1, 0, 1, 0, 1])
    r = oracle.c4_frozen_search(cfg(), 9, 0, my, op, 25)
    assert r["best_action"][0] == 0
    assert list(r["root_sol"][0]) == [1, 2, 0]            # Some(Win(0))
    assert list(r["child_sol"][0, 0]) == [1, 0, 0]        # Some(Lose(0)) from the child's side to move
    assert r["root_stat"][0, 0] == 26.0                   # explore_n never stops early: solved roots keep counting
    assert r["child_N"][0].sum() == 0.0                   # ... without ever descending (explore returns at the solved root)
    # every later explore backs up outcome.value() = +1 to the solved root only
    first = oracle.c4_frozen_search(cfg(), 9, 0, my, op, 0)
    assert first["root_stat"][0, 0] == 1.0 and first["root_stat"][0, 1] == 1.0   # -0 + (0 + 1)
    assert r["root_stat"][0, 1] == 26.0
    off = oracle.c4_frozen_search(cfg(solve=0), 9, 0, my, op, 25)
    assert off["root_sol"][0, 0] == 0 and off["best_action"][0] == 0 and off["child_N"][0, 0] > 0


def test_forced_loss_is_proven_two_plies_deep(oracle):
    """The opponent threatens two different fours: every reply loses. Children are marked Win(0) — the baseline does not count
    the turns of a win it proves (evaluator.rs:500-502) — so the root ends up worst.reversed() = Lose(1)."""
    # second player holds columns 2,3,4 on the bottom row with both ends (1 and 5) open; first player's stones are elsewhere
    my, op = position(oracle, [8, 2, 8, 3, 0, 4])
    r = oracle.c4_frozen_search(cfg(), 3, 0, my, op, 400)
    assert list(r["root_sol"][0]) == [1, 0, 1]                                       # Some(Lose(1))
    assert (r["child_sol"][0] == np.array([1, 2, 0])).all()                          # every child Some(Win(0))
    assert r["root_stat"][0, 0] == 401.0


def test_stream_position_continues_across_searches(oracle):
    """A search that starts where the previous one stopped equals the match-level restatement's second move"""
    reward, moves, words = oracle.c4_mcts_vs_mcts(cfg(), 0, 120, 60, 77)
    assert reward in (-1.0, 0.0, 1.0) and 7 <= moves.size <= 63 and np.all(np.diff(words.astype(np.int64)) > 0)
    z = np.zeros(1, np.uint64)
    a = oracle.c4_frozen_search(cfg(), 77, 0, z, z, 120)
    assert a["best_action"][0] == moves[0] and a["rng_words"][0] == words[0]
    my, op = position(oracle, [int(moves[0])])
    b = oracle.c4_frozen_search(cfg(), 77, words[0], my, op, 60)
    assert b["best_action"][0] == moves[1] and b["rng_words"][0] == words[1]
    # `player` only decides who gets which explore count
    r2, m2, _ = oracle.c4_mcts_vs_mcts(cfg(), 1, 60, 120, 77)
    assert r2 == reward and np.array_equal(m2, moves)


def test_more_explores_wins_more(oracle):
    """Sanity of the baseline ladder the evaluator relies on: VanillaMCTS400 beats VanillaMCTS25 from both colours"""
    score = 0.0
    for g in range(12):
        score += oracle.c4_mcts_vs_mcts(cfg(), 0, 400, 25, 1000 + g)[0]
        score -= oracle.c4_mcts_vs_mcts(cfg(), 1, 400, 25, 2000 + g)[0]
    assert score >= 12   # >= 75 % of 24 games


def test_network_policy_variant_uses_legal_softmax(oracle, golden_dir):
    import os

    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    my, op = position(oracle, [4] * 7 + [3])   # column 4 is full
    r = oracle.c4_frozen_search(cfg(fpu_value=1.0), 0, 0, my, op, 30, blob=blob)
    assert r["child_P"][0, 4] == 0.0 and abs(r["child_P"][0].sum() - 1.0) < 1e-6 and r["child_N"][0, 4] == 0.0
    assert r["root_stat"][0, 0] == 31.0 and r["rng_words"][0] == 0


def test_cpp_oracle_equals_the_independent_python_restatement(oracle):
    """tests/frozen_py.py restates evaluator.rs:230-534 + rollout.rs a second time (objects and Python ints instead of flat
    arrays and bitboard tricks). Both restatements must build bit-identical trees: opening, middle-game and solver-heavy
    end-game positions, both action selections, solver on and off, a finite fpu, a stream that starts mid-way."""
    from tests import frozen_py
    from tests.test_gpu_parity import random_positions

    my, op = random_positions(oracle, 14, seed=61, max_moves=61)
    my = np.concatenate([np.zeros(1, np.uint64), my]); op = np.concatenate([np.zeros(1, np.uint64), op])
    variants = [dict(), dict(fpu_value=0.5, c=1.2), dict(solve=0)]
    for i in range(my.size):
        kw = variants[i % 3]
        explores = [30, 120, 7][i % 3]
        by_q = i % 2 == 0
        ref = oracle.c4_frozen_search(cfg(**kw), 900 + i, 3 * i, my[i:i + 1], op[i:i + 1], explores, action_selection=0 if by_q else 1)
        got = frozen_py.frozen_search(oracle, my[i], op[i], 900 + i, 3 * i, explores, c=kw.get("c", 2.0),
                                      fpu=kw.get("fpu_value", np.inf), solve=bool(kw.get("solve", 1)), by_q=by_q)
        for k in ("child_N", "child_cum", "child_P"):
            assert np.array_equal(got[k].view(np.uint32), ref[k][0].view(np.uint32)), (i, k)
        assert np.array_equal(got["child_sol"], ref["child_sol"][0]) and np.array_equal(got["root_sol"], ref["root_sol"][0]), i
        assert np.array_equal(got["root_stat"].view(np.uint32), ref["root_stat"][0].view(np.uint32)), i
        assert (got["num_nodes"], got["best_action"], got["rng_words"]) == (ref["num_nodes"][0], ref["best_action"][0], ref["rng_words"][0]), i
