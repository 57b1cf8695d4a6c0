"""TEST INFRASTRUCTURE: an independent plain-Python restatement of the reference's self-play game loop, written from the
text of synthesis/src/alpha_zero.rs:229-338 (run_game, sample_action, fill_state_info, store_rewards) on top of the Python
tree of tests/mcts_py.py. One game = one StdRng::seed_from_u64(seed), the oracle's (and the engine's) per-game seeding.
`tests/test_oracle_kats.py` demands identical trajectories and bit-identical targets from the C++ oracle.

rand 0.8.3 pieces used by sample_action, restated from the crate's published algorithm: `gen_range(0..n as u8)` (in
tests/frozen_py.py), `WeightedIndex::<f32>::new(w).sample(rng)` = running f32 sums of the weights, one `Uniform::new(0, total)`
draw (23 random mantissa bits in [1, 2) minus one, times total), the number of running sums <= the draw."""
import numpy as np

from tests.frozen_py import F, Game, Stream, sol_reversed
from tests.mcts_py import ONEHOT, MctsPy


def weighted_index(rng, weights):
    cum, total = [], F(weights[0])
    for w in weights[1:]:
        cum.append(total)
        total = F(total + F(w))
    bits = (int(rng.words[rng.pos]) >> 9) | 0x3F800000
    rng.pos += 1
    v01 = F(np.array([bits], np.uint32).view(F)[0] - F(1.0))
    chosen = F(F(v01 * total) + F(0.0))
    return sum(1 for c in cum if c <= chosen)


def run_game(oracle, blob, rollout_cfg, seed, nn_mode=1):
    rc = {k: getattr(rollout_cfg, k) for k, _ in rollout_cfg._fields_ if k != "mcts"}
    mc = {k: getattr(rollout_cfg.mcts, k) for k, _ in rollout_cfg.mcts._fields_}
    rng = Stream(oracle, seed, 0, budget=4096)
    game, solution, num_turns = Game(0, 0), None, 0
    states, pis, infos, actions, tree_sizes = [], [], [], [], []
    while solution is None:
        mcts = MctsPy(oracle, blob, mc, game, nn_mode)
        mcts.explore_n(rc["num_explores"])
        pi = mcts.target_policy()
        states.append((game.my, game.op))
        pis.append(pi)
        infos.append(dict(turn=num_turns + 1, q=mcts.target_q(), z=np.zeros(3, F), t=F(0.0)))
        tree_sizes.append(mcts.count)
        # sample_action (alpha_zero.rs:270-293)
        best = mcts.best_action(rc["action"] == 0)
        sol_of = lambda a: next((ch.solution for ch in mcts.root.children if ch.action == a), None)  # noqa: E731
        if num_turns < rc["random_actions_until"]:
            acts = game.actions()
            action = acts[rng.gen_range_u8(len(acts))]
        elif num_turns < rc["sample_actions_until"] and (sol_of(best) is None or not rc["stop_games_when_solved"]):
            action = weighted_index(rng, pi)
        else:
            action = best
        actions.append(action)
        solution = sol_of(action)
        game, is_over = game.step(action)
        if is_over:
            r = game.reward_for_mover_to_be()
            solution = ("W", 0) if r > 0 else (("L", 0) if r < 0 else ("D", 0))
        elif not rc["stop_games_when_solved"]:
            solution = None
        num_turns += 1
    # fill_state_info (alpha_zero.rs:295-307): the last position's mover sees solution.reversed()
    outcome = sol_reversed(solution)
    n = len(infos)
    for info in reversed(infos):
        info["z"][ONEHOT[outcome[0]]] = F(1.0)
        info["t"] = F(F(info["turn"]) / F(n))
        outcome = sol_reversed(outcome)
    # store_rewards (alpha_zero.rs:309-338)
    vs = []
    for info in infos:
        q, z, t = info["q"], info["z"], info["t"]
        if rc["value_target"] == 1:
            v = q
        elif rc["value_target"] == 0:
            v = z
        elif rc["value_target"] == 2:
            p = F(rc["vt_p"])
            v = np.array([F(F(q[i] * p) + F(z[i] * F(F(1.0) - p))) for i in range(3)], F)
        else:
            p = F(F(F(F(1.0) - t) * F(rc["vt_from"])) + F(t * F(rc["vt_to"])))
            v = np.array([F(F(q[i] * F(F(1.0) - p)) + F(z[i] * p)) for i in range(3)], F)
        vs.append(np.asarray(v, F))
    return dict(plies=n, states=states, pis=np.array(pis, F), vs=np.array(vs, F), actions=actions, tree_sizes=tree_sizes,
                final_kind=ONEHOT[solution[0]])
