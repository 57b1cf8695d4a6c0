"""TEST INFRASTRUCTURE: a second, independent restatement of the reference's self-play tree in plain Python, written from the
text of synthesis/src/mcts.rs:28-488 (Node, with_capacity / explore_n, target_policy / target_q, root noise, best_action,
explore / select_best_child / exploit_value / explore_value, visit with auto-extend and the legal softmax, backprop with the
solver and value correction) — objects, lists and Python-int boards here; flat arrays and bitboard tricks in oracle/mcts.hpp.
Small cases only. `tests/test_oracle_kats.py` demands bit-identical trees from both restatements, which guards the oracle (and
through it the HIP kernels) against transcription mistakes on the Connect4 configurations the reference's own TicTacToe tests
do not reach (PolynomialUct, auto-extend, Fpu::ParentQ, equalising noise, value correction on / off).

Shared with the oracle on purpose: `Policy::eval` (the network's 9 logits and 3 outcome probabilities per position, from
`oracle.c4net_eval`), and the deterministic `expf` / `logf`. All tree arithmetic is numpy float32 in the reference's order."""
import numpy as np

from tests.frozen_py import F, INF, Game, sol_key, sol_reversed, sol_value

ONEHOT = {"L": 0, "D": 1, "W": 2}


class Node:
    __slots__ = ("parent", "children", "game", "solution", "action", "prob", "w", "n")

    def __init__(self, parent, game, solution, action, prob):
        self.parent, self.children, self.game, self.solution, self.action = parent, [], game, solution, action
        self.prob, self.w, self.n = F(prob), [F(0.0), F(0.0), F(0.0)], F(0.0)

    def q(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            return F(F(self.w[2] - self.w[0]) / self.n)


def onehot(sol):
    d = [F(0.0), F(0.0), F(0.0)]
    d[ONEHOT[sol[0]]] = F(1.0)
    return d


def opt_gt(a, b):
    """Rust's `Some(a) > b` for Option<f32> / Option<(f32, f32)> with partial order: anything beats None, NaN beats nothing"""
    if b is None:
        return True
    if isinstance(a, tuple):
        if a[0] != b[0]:
            return a[0] > b[0]
        return a[1] > b[1]
    return a > b


class MctsPy:
    def __init__(self, oracle, blob, cfg, game, nn_mode, policy_fn=None):
        self.o, self.blob, self.cfg, self.nn_mode = oracle, blob, cfg, nn_mode
        if policy_fn is not None:          # e.g. RolloutPolicy (tests/frozen_py.rollout_eval on a Stream)
            self.policy = policy_fn
        self.root = Node(None, game, None, 0, 0.0)
        self.count = 1
        leaf, dist, any_solved = self.visit(self.root)
        self.backprop(leaf, dist, any_solved)
        if cfg["noise"] == 1 and len(self.root.children) >= 2:          # PolicyNoise::Equal (mcts.rs:258-269)
            w = F(cfg["noise_weight"])
            noise = F(F(1.0) / F(len(self.root.children)))
            for ch in self.root.children:
                ch.prob = F(F(ch.prob * F(F(1.0) - w)) + F(w * noise))

    def policy(self, game):
        lg, v = self.o.c4net_eval(self.blob, np.array([game.my], np.uint64), np.array([game.op], np.uint64), mode=self.nn_mode)
        return lg[0], [F(v[0][0]), F(v[0][1]), F(v[0][2])]

    def explore_n(self, n):
        for _ in range(n):
            if self.root.solution is not None:
                break
            node = self.root
            while True:
                if node.solution is not None:
                    self.backprop(node, onehot(node.solution), True)
                    break
                if not node.children:          # is_unvisited (no solution here)
                    leaf, dist, any_solved = self.visit(node)
                    self.backprop(leaf, dist, any_solved)
                    break
                node = self.select(node)

    def select(self, parent):
        cfg = self.cfg
        best, best_v = None, None
        for ch in parent.children:
            if ch.solution is not None:
                q = sol_value(sol_reversed(ch.solution)) if cfg["select_solved_nodes"] else -INF
            elif not ch.children:
                q = F(cfg["fpu_value"]) if cfg["fpu"] == 0 else parent.q()
            else:
                q = F(-ch.q())
            with np.errstate(divide="ignore", invalid="ignore"):
                if cfg["exploration"] == 0:
                    visits = np.sqrt(F(F(cfg["c"]) * self.o.det_logf([parent.n])[0]), dtype=F)
                    u = F(visits / np.sqrt(ch.n, dtype=F))
                else:
                    visits = np.sqrt(parent.n, dtype=F)
                    u = F(F(F(F(cfg["c"]) * ch.prob) * visits) / F(F(1.0) + ch.n))
                v = F(q + u)
            if opt_gt(v, best_v):
                best, best_v = ch, v
        return best

    def visit(self, node):
        if node.solution is not None:
            return node, onehot(node.solution), True
        any_solved = False
        for a in node.game.actions():
            child_game, is_over = node.game.step(a)
            sol = None
            if is_over:
                any_solved = True
                r = child_game.reward_for_mover_to_be()
                sol = ("W", 0) if r > 0 else (("L", 0) if r < 0 else ("D", 0))
            node.children.append(Node(node, child_game, sol, a, 1.0))
            self.count += 1
        if self.cfg["auto_extend"] and len(node.children) == 1:
            return self.visit(node.children[0])
        logits, dist = self.policy(node.game)
        max_logit = -INF
        for ch in node.children:
            ch.prob = F(logits[ch.action])
            max_logit = max(max_logit, ch.prob)
        total = F(0.0)
        for ch in node.children:
            ch.prob = self.o.det_expf([F(ch.prob - max_logit)])[0]
            total = F(total + ch.prob)
        for ch in node.children:
            ch.prob = F(ch.prob / total)
        return node, dist, any_solved

    def backprop(self, node, dist, solved):
        cfg = self.cfg
        dist = list(dist)
        while True:
            if cfg["solve"] and solved:
                all_solved, best = True, node.solution
                for ch in node.children:
                    soln = sol_reversed(ch.solution) if ch.solution is not None else None
                    all_solved = all_solved and soln is not None
                    if soln is not None and (best is None or sol_key(soln) >= sol_key(best)):   # Ord::max keeps the later of equals
                        best = soln
                if best is not None and best[0] == "W":
                    node.solution = best
                    if cfg["correct_values_on_solve"]:
                        dist = [F(-x) for x in node.w]
                        dist[2] = F(dist[2] + F(node.n + F(1.0)))
                elif best is not None and all_solved:
                    node.solution = best
                    if cfg["correct_values_on_solve"]:
                        dist = [F(-x) for x in node.w]
                        k = 1 if best[0] == "D" else 0
                        dist[k] = F(dist[k] + F(node.n + F(1.0)))
                else:
                    solved = False
            node.w = [F(node.w[i] + dist[i]) for i in range(3)]
            node.n = F(node.n + F(1.0))
            if node.parent is None:
                return
            dist[0], dist[2] = dist[2], dist[0]
            node = node.parent

    def best_action(self, by_q):
        best, best_v = None, None
        for ch in self.root.children:
            if ch.solution is None:
                v = (1.0, F(-ch.q())) if by_q else (1.0, ch.n)
            elif ch.solution[0] == "W":
                v = (0.0, F(ch.solution[1]))
            elif ch.solution[0] == "D":
                v = (2.0, F(-F(ch.solution[1])))
            else:
                v = (3.0, F(-F(ch.solution[1])))
            if opt_gt(v, best_v):
                best, best_v = ch.action, v
        return best

    def target_policy(self):
        pi = np.zeros(9, F)
        total = F(0.0)
        root = self.root
        if root.n == 1.0:
            for ch in root.children:
                if root.solution is not None and root.solution[0] == "W":
                    v = F(1.0) if (ch.solution is not None and ch.solution[0] == "L") else F(0.0)
                else:
                    v = F(1.0)
                pi[ch.action] = v
                total = F(total + v)
        else:
            for ch in root.children:
                pi[ch.action] = ch.n
                total = F(total + ch.n)
        with np.errstate(divide="ignore", invalid="ignore"):
            return (pi / total).astype(F)

    def target_q(self):
        if self.root.solution is not None:
            return np.array(onehot(self.root.solution), F)
        return np.array([F(self.root.w[i] / self.root.n) for i in range(3)], F)


def mcts_search(oracle, blob, cfg_struct, my_bb, op_bb, explores, by_q=False, nn_mode=1, policy_fn=None):
    cfg = {k: getattr(cfg_struct, k) for k, _ in cfg_struct._fields_}
    t = MctsPy(oracle, blob, cfg, Game(int(my_bb), int(op_bb)), nn_mode, policy_fn)
    t.explore_n(explores)
    out = dict(child_N=np.zeros(9, F), child_W=np.zeros((9, 3), F), child_P=np.zeros(9, F), child_sol=np.zeros((9, 3), np.int32),
               root_stat=np.array([t.root.n] + t.root.w, F), num_nodes=t.count, best_action=t.best_action(by_q),
               target_pi=t.target_policy(), target_q=t.target_q(),
               root_sol=np.array([1, ONEHOT[t.root.solution[0]], t.root.solution[1]] if t.root.solution else [0, 0, 0], np.int32))
    for ch in t.root.children:
        out["child_N"][ch.action], out["child_P"][ch.action] = ch.n, ch.prob
        out["child_W"][ch.action] = ch.w
        if ch.solution:
            out["child_sol"][ch.action] = [1, ONEHOT[ch.solution[0]], ch.solution[1]]
    return out
