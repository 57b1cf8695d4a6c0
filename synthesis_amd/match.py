"""Matches between two MCTS players with every search on the GPU — the compute of the reference's evaluation
(synthesis/src/evaluator.rs:129-227: `MCTS::exploit` per move = `with_capacity` + `explore_n` + `best_action`).

All games of a match advance in lockstep on the host: per ply, the positions where player A is to move go through one
batched `Engine.mcts_search` call and player B's through another, each root on its own device-resident tree. A player is
(MCTSConfig, explores, action selection, kind):
  * the engine's network under `MCTS` (evaluator.rs:129-160 eval_against_old, and the policy side of :163-198),
  * `vanilla_player`: the evaluator's "VanillaMCTS<n>" baseline = `FrozenMCTS` over `RolloutPolicy` (evaluator.rs:163-228,
    230-534), every game on its own StdRng::seed_from_u64(seed) that lasts the whole game, as in the reference — so
    `play_match(first=net, second=vanilla_player(n), seeds=...)` replays eval_against_rollout_mcts move for move and two
    vanilla players replay mcts_vs_mcts,
  * `rollout_player`: the self-play `MCTS` tree over RolloutPolicy (the pairing of the reference's MCTS tests).

Host code only applies the chosen moves (numpy bitboards, connect4.rs:221-233) — no search, no network."""
from dataclasses import dataclass, field

import numpy as np

from .config import ActionSelection, Exploration, Fpu, MCTSConfig

_U = np.uint64
_COLS0TO5 = _U(sum(0x7F << (7 * c) for c in range(6)))
_ROW0 = _U(sum(1 << (7 * c) for c in range(9)))
_ROWS = lambda lo, hi: _U(sum(int(_ROW0) << i for i in range(lo, hi + 1)))  # noqa: E731
_D1, _D2, _H, _V = _COLS0TO5 & _ROWS(3, 6), _COLS0TO5 & _ROWS(0, 3), _COLS0TO5, _ROWS(0, 3)
FULL = _U((1 << 63) - 1)


def won(bb):
    """four in a row on a bitboard (bit = row + 7 * col), vectorised; connect4.rs:77-83"""
    bb = np.asarray(bb, dtype=np.uint64)
    s = lambda k: bb >> _U(k)  # noqa: E731
    d1 = bb & s(6) & s(12) & s(18) & _D1
    d2 = bb & s(8) & s(16) & s(24) & _D2
    h = bb & s(7) & s(14) & s(21) & _H
    v = bb & s(1) & s(2) & s(3) & _V
    return (d1 | d2 | h | v) != 0


def step(my, op, col):
    """Game::step for arrays of positions: returns (my', op', over, mover_won)"""
    my = np.asarray(my, np.uint64); op = np.asarray(op, np.uint64); col = np.asarray(col, np.int64)
    occ = my | op
    colbits = (occ >> (_U(7) * col.astype(np.uint64))) & _U(0x7F)
    height = np.zeros(col.shape, np.uint64)
    for r in range(7):
        height += (colbits >> _U(r)) & _U(1)
    bit = _U(1) << (height + _U(7) * col.astype(np.uint64))
    mover = my | bit
    w = won(mover)
    over = w | ((occ | bit) == FULL)
    return op.copy(), mover, over, w


@dataclass
class Player:
    name: str
    explores: int
    mcts_cfg: MCTSConfig = field(default_factory=MCTSConfig)
    action: ActionSelection = ActionSelection.NumVisits
    rollout: bool = False          # False: the engine's network is the leaf policy; True: RolloutPolicy playouts
    frozen: bool = False           # True: the evaluator's FrozenMCTS baseline over RolloutPolicy (evaluator.rs:230-534)
    weights: object = None         # network players: this player's own weight blob (None: whatever the engine holds) — lets
                                   # two different checkpoints meet on one engine (eval_against_old, evaluator.rs:129-160)


def rollout_player(explores, name=None):
    """The self-play MCTS tree over RolloutPolicy under rollout_mcts_cfg of study-connect4/src/main.rs:74-82 (Uct{c: 2}, no
    auto-extend, fpu = inf), moves by visit count"""
    return Player(name or f"RolloutMCTS{explores}", explores,
                  MCTSConfig(exploration=Exploration.Uct, c=2.0, auto_extend=False, fpu=Fpu.Const, fpu_value=float("inf")),
                  ActionSelection.NumVisits, rollout=True)


def vanilla_player(explores, name=None, action=ActionSelection.Q):
    """The evaluator's baseline (evaluator.rs:184-190, 212-222) under rollout_mcts_cfg / rollout_action of
    study-connect4/src/main.rs:72-82: FrozenMCTS, Uct{c: 2}, fpu = inf, ActionSelection::Q"""
    return Player(name or f"VanillaMCTS{explores}", explores,
                  MCTSConfig(exploration=Exploration.Uct, c=2.0, auto_extend=False, fpu=Fpu.Const, fpu_value=float("inf")),
                  action, rollout=True, frozen=True)


def play_match(engine, first: Player, second: Player, n_games, seed=0, seeds=None, record=None):
    """n_games from the empty board, `first` moving first. Returns rewards for `first` per game: +1 win, 0 draw, -1 loss
    (game.reward(first_player), evaluator.rs:160) and the number of plies. Game g's rollout generator is
    StdRng::seed_from_u64(seeds[g]) (default seed + g), shared by both sides and alive for the whole game when the players
    are vanilla players (evaluator.rs:171-172, 207-208). `record`: optional dict that receives "moves" [n_games, 63]."""
    net = [p for p in (first, second) if not p.frozen and not p.rollout]
    if len(net) == 2 and (net[0].weights is None) != (net[1].weights is None):
        # the engine cannot hand back the weights it currently holds, so after the first swap a `weights=None` player would
        # silently search with its opponent's network
        raise ValueError("play_match: give both network players explicit weights, or neither (both then use the engine's network)")
    my = np.zeros(n_games, np.uint64); op = np.zeros(n_games, np.uint64)
    alive = np.ones(n_games, bool)
    reward = np.zeros(n_games, np.float32)
    plies = np.zeros(n_games, np.int32)
    seeds = (np.uint64(seed) + np.arange(n_games, dtype=np.uint64)) if seeds is None else np.asarray(seeds, np.uint64)
    words = np.zeros(n_games, np.uint64)
    moves = np.full((n_games, 63), 255, np.uint8)
    loaded = None
    for ply in range(63):
        idx = np.nonzero(alive)[0]
        if idx.size == 0:
            break
        p = first if ply % 2 == 0 else second
        if p.frozen:
            res = engine.frozen_search(p.mcts_cfg, seeds[idx], words[idx], my[idx], op[idx], p.explores,
                                       action_selection=int(p.action))
            words[idx] = res["rng_words"]
        else:
            if p.weights is not None and not p.rollout and loaded is not p.weights:
                engine.load_weights(p.weights)   # 122 KB upload; also empties the policy cache
                loaded = p.weights
            kw = dict(rollout_seed=int(seed) + ply * n_games) if p.rollout else {}
            res = engine.mcts_search(p.mcts_cfg, my[idx], op[idx], p.explores, action_selection=int(p.action), **kw)
        moves[idx, ply] = res["best_action"]
        nmy, nop, over, w = step(my[idx], op[idx], res["best_action"])
        my[idx], op[idx] = nmy, nop
        plies[idx] += 1
        mover_is_first = ply % 2 == 0
        reward[idx[over & w]] = 1.0 if mover_is_first else -1.0
        alive[idx[over]] = False
    if record is not None:
        record["moves"] = moves
        record["rng_words"] = words
    return reward, plies


def pgn_records(white_name, black_name, white_rewards):
    """add_pgn_result (synthesis/src/utils.rs:31-52) for a batch of games: the text the evaluator appends to results.pgn,
    which its rating tool (the external bayeselo binary, utils.rs:54-72) reads."""
    out = []
    for r in np.asarray(white_rewards, dtype=np.float32).ravel():
        if r == 1.0:
            result = "1-0"
        elif r == -1.0:
            result = "0-1"
        else:
            assert r == 0.0
            result = "1/2-1/2"
        out.append(f'[White "{white_name}"]\n[Black "{black_name}"]\n[Result "{result}"]\n{result}\n')
    return "".join(out)


@dataclass
class EvaluationConfig:
    """config.rs:59-73 (without `logs`); defaults = eval_cfg of study-connect4/src/main.rs:52-83 except the explore counts,
    which the caller sets to its budget (the reference: policy 1600; baselines 800, 1600, ..., 204800)."""
    policy_num_explores: int = 800
    policy_action: ActionSelection = ActionSelection.NumVisits
    policy_mcts_cfg: MCTSConfig = field(default_factory=MCTSConfig)
    num_best_policies: int = 10
    num_games_against_rollout: int = 5
    rollout_action: ActionSelection = ActionSelection.Q
    rollout_num_explores: tuple = (800, 1600, 3200)


def evaluation_round(engine, cfg: EvaluationConfig, i_iter, model_name, model_weights, best_k):
    """One pass of the evaluator's loop body (evaluator.rs:22-99) for model `i_iter`, every search on the device and all
    pairings of a kind batched into one lockstep match each. Returns the PGN text the reference appends to results.pgn, in its
    order: (1) baseline i_iter % n against every other baseline, seed i_iter (mcts_vs_mcts, :24-41); (2) the model against
    every baseline, seeds 0..num_games_against_rollout-1, as first and as second player (:63-86); (3) the model against each
    kept older model, both colours (eval_against_old, :89-95). `best_k`: list of (name, weights). Ratings and the choice of
    which models to keep are the external bayeselo's job in the reference (utils.rs:54-72) and stay with the caller."""
    pgn = []
    ladder = list(cfg.rollout_num_explores)
    i = i_iter % len(ladder)
    for j in range(len(ladder)):
        if i == j:
            continue
        a, b = vanilla_player(ladder[i], action=cfg.rollout_action), vanilla_player(ladder[j], action=cfg.rollout_action)
        r, _ = play_match(engine, a, b, 1, seeds=[i_iter])
        pgn.append(pgn_records(a.name, b.name, r))
    me = Player(model_name, cfg.policy_num_explores, cfg.policy_mcts_cfg, cfg.policy_action, weights=model_weights)
    seeds = np.arange(cfg.num_games_against_rollout, dtype=np.uint64)
    for ex in ladder:
        opp = vanilla_player(ex, action=cfg.rollout_action)
        r1, _ = play_match(engine, me, opp, seeds.size, seeds=seeds)
        r2, _ = play_match(engine, opp, me, seeds.size, seeds=seeds)
        for g in range(seeds.size):  # the reference interleaves the two colours per seed
            pgn.append(pgn_records(me.name, opp.name, r1[g:g + 1]))
            pgn.append(pgn_records(opp.name, me.name, r2[g:g + 1]))
    for prev_name, prev_w in best_k:
        old = Player(prev_name, cfg.policy_num_explores, cfg.policy_mcts_cfg, cfg.policy_action, weights=prev_w)
        r, _ = play_match(engine, me, old, 1)
        pgn.append(pgn_records(me.name, old.name, r))
        r, _ = play_match(engine, old, me, 1)
        pgn.append(pgn_records(old.name, me.name, r))
    return "".join(pgn)


def score(rewards):
    """(wins, draws, losses, score in [0, 1], Elo difference implied by the score)"""
    r = np.asarray(rewards)
    w, d, l = int((r > 0).sum()), int((r == 0).sum()), int((r < 0).sum())
    s = (w + 0.5 * d) / max(1, r.size)
    s_c = min(max(s, 1e-3), 1 - 1e-3)
    return w, d, l, s, float(-400.0 * np.log10(1.0 / s_c - 1.0))
