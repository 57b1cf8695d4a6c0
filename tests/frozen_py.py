"""TEST INFRASTRUCTURE: a second, independent restatement of the evaluator's baseline tree in plain Python, written from the
text of synthesis/src/evaluator.rs:230-534 and policies/rollout.rs:8-31 without looking at oracle/frozen_mcts.hpp's structure
(objects and lists here, flat arrays there). Small cases only. It exists to catch transcription mistakes in the C++ oracle:
`tests/test_oracle_frozen.py` demands bit-identical trees from both.

Shared with the oracle on purpose: the raw StdRng output words (`oracle.stdrng_u32`, pinned by the rand crates' own
constants) and the deterministic `logf` (`oracle.det_logf`; exp is only ever taken of 0 here). All arithmetic is numpy float32
in the reference's operation order."""
import numpy as np

F = np.float32
INF = F(np.inf)
W, H = 9, 7
FULL = (1 << 63) - 1


def _lines():
    """every four cells in a row, column or diagonal of the 7-row x 9-column board, as bit masks (bit = row + 7 * col)"""
    out = []
    for r in range(H):
        for c in range(W):
            for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
                cells = [(r + k * dr, c + k * dc) for k in range(4)]
                if all(0 <= rr < H and 0 <= cc < W for rr, cc in cells):
                    out.append(sum(1 << (rr + 7 * cc) for rr, cc in cells))
    return out


LINES = _lines()


def won(bb):
    """four in a row (connect4.rs:77-83), by geometry rather than by the reference's shift-and-mask formulation"""
    return any(bb & m == m for m in LINES)


class Game:
    """Connect4 as the baseline sees it: stones of the side to move / of the other side (connect4.rs:108-233)"""
    __slots__ = ("my", "op")

    def __init__(self, my, op):
        self.my, self.op = my, op

    def actions(self):
        occ = self.my | self.op
        return [c for c in range(W) if not (occ >> (7 * c + H - 1)) & 1]      # iter_actions: free columns, ascending

    def over(self):
        return won(self.op) or (self.my | self.op) == FULL

    def step(self, col):
        occ = self.my | self.op
        h = bin((occ >> (7 * col)) & 0x7F).count("1")
        g = Game(self.op, self.my | (1 << (h + 7 * col)))
        return g, g.over()

    def reward_for_mover_to_be(self):
        """reward(self.player()): the side to move never has a four; it lost if the other side has one"""
        return F(-1.0) if won(self.op) else F(0.0)


class Stream:
    """RolloutPolicy's `rng`: words of StdRng::seed_from_u64(seed) from position `pos`"""

    def __init__(self, oracle, seed, pos, budget=400000):
        self.words = oracle.stdrng_u32(seed, pos + budget)
        self.pos = pos

    def gen_range_u8(self, n):   # rand 0.8.3 UniformInt<u8>::sample_single: widening multiply with a modulus zone
        zone = 0xFFFFFFFF - (0xFFFFFFFF - n + 1) % n
        while True:
            m = int(self.words[self.pos]) * n
            self.pos += 1
            if (m & 0xFFFFFFFF) <= zone:
                return m >> 32


def rollout_eval(game, rng):
    """policies/rollout.rs:8-31: zero logits; the outcome of one uniformly random playout, seen from the node's mover"""
    g, is_over, leaf_mover_to_move = game, game.over(), True
    while not is_over:
        acts = g.actions()
        g, is_over = g.step(acts[rng.gen_range_u8(len(acts))])
        leaf_mover_to_move = not leaf_mover_to_move
    if won(g.op):                      # the side that just moved made four
        r = -1.0 if leaf_mover_to_move else 1.0
    else:
        r = 0.0
    return [F(0.0)] * W, ([F(0), F(1), F(0)] if r == 0 else ([F(1), F(0), F(0)] if r < 0 else [F(0), F(0), F(1)]))


# Outcome (game.rs:9-66) as (kind, turns), kinds "L" < "D" < "W"; Lose/Draw prefer more turns, Win fewer
def sol_key(s):
    kind, t = s
    return {"L": 0, "D": 1, "W": 2}[kind], (-t if kind == "W" else t)


def sol_reversed(s):
    return {"W": "L", "L": "W", "D": "D"}[s[0]], s[1] + 1


def sol_value(s):
    return {"W": F(1.0), "D": F(0.0), "L": F(-1.0)}[s[0]]


class Node:
    __slots__ = ("parent", "children", "game", "solution", "action", "prob", "cum", "visits")

    def __init__(self, parent, game, solution, action, prob):
        self.parent, self.children, self.game, self.solution = parent, [], game, solution
        self.action, self.prob, self.cum, self.visits = action, prob, F(0.0), F(0.0)

    def unvisited(self):
        return not self.children and self.solution is None


class FrozenPy:
    def __init__(self, oracle, c, fpu, solve, game, rng):
        self.c, self.fpu, self.solve, self.rng, self.oracle = F(c), F(fpu), solve, rng, oracle
        self.root = Node(None, game, None, 0, F(0.0))
        self.count = 1
        v, any_solved = self.visit(self.root)
        self.backprop(self.root, v, any_solved)

    def ln(self, x):
        return self.oracle.det_logf([x])[0]

    def explore(self):
        node = self.root
        while True:
            if node.solution is not None:
                return self.backprop(node, sol_value(node.solution), True)
            if node.unvisited():
                v, any_solved = self.visit(node)
                return self.backprop(node, v, any_solved)
            node = self.select(node)

    def select(self, node):
        best, best_v = None, -INF
        for ch in node.children:
            if ch.unvisited():
                v = self.fpu + ch.prob
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    q = sol_value(sol_reversed(ch.solution)) if ch.solution is not None else (-ch.cum) / ch.visits
                    visits = np.sqrt(self.c * self.ln(node.visits), dtype=F)
                    v = q + visits / np.sqrt(ch.visits, dtype=F)
            if best is None or v > best_v:
                best, best_v = ch, v
        return best

    def visit(self, node):
        logits, dist = rollout_eval(node.game, self.rng)
        any_solved, max_logit = False, -INF
        for a in node.game.actions():
            child_game, is_over = node.game.step(a)
            sol = None
            if is_over:
                any_solved = True
                r = child_game.reward_for_mover_to_be()
                sol = ("W", 0) if r > 0 else (("L", 0) if r < 0 else ("D", 0))
            max_logit = max(max_logit, logits[a])
            node.children.append(Node(node, child_game, sol, a, logits[a]))
            self.count += 1
        total = F(0.0)
        for ch in node.children:
            ch.prob = self.oracle.det_expf([ch.prob - max_logit])[0]
            total = F(total + ch.prob)
        for ch in node.children:
            ch.prob = F(ch.prob / total)
        return F(dist[2] - dist[0]), any_solved

    def backprop(self, node, value, solved):
        while True:
            if self.solve and solved and node.solution is None:
                all_solved, worst = True, None
                for ch in node.children:
                    if ch.unvisited() or ch.solution is None:
                        all_solved = False
                    elif worst is None or sol_key(ch.solution) < sol_key(worst):
                        worst = ch.solution
                if worst is not None and worst[0] == "L":
                    node.solution = ("W", 0)
                    value = F(-node.cum + F(node.visits + F(1.0)))
                elif node.children and all_solved:
                    best_for_me = sol_reversed(worst)
                    node.solution = best_for_me
                    value = F(-node.cum) if best_for_me[0] == "D" else F(-node.cum - F(node.visits + F(1.0)))
                else:
                    solved = False
            node.cum = F(node.cum + value)
            node.visits = F(node.visits + F(1.0))
            value = F(-value)
            if node.parent is None:
                return
            node = node.parent

    def best_action(self, by_q):
        best, best_v = None, -INF
        for ch in self.root.children:
            if ch.unvisited():
                continue
            if ch.solution is not None:
                v = {"W": -INF, "D": F(1e6), "L": INF}[ch.solution[0]]
            else:
                v = (-ch.cum) / ch.visits if by_q else ch.visits
            if best is None or v > best_v:
                best, best_v = ch.action, v
        return best


def frozen_search(oracle, my_bb, op_bb, seed, first_word, explores, c=2.0, fpu=np.inf, solve=True, by_q=False):
    rng = Stream(oracle, seed, first_word)
    t = FrozenPy(oracle, c, fpu, solve, Game(int(my_bb), int(op_bb)), rng)
    for _ in range(explores):
        t.explore()
    kinds = {"L": 0, "D": 1, "W": 2}
    out = dict(child_N=np.zeros(9, F), child_cum=np.zeros(9, F), child_P=np.zeros(9, F), child_sol=np.zeros((9, 3), np.int32),
               root_stat=np.array([t.root.visits, t.root.cum], F), num_nodes=t.count, best_action=t.best_action(by_q),
               rng_words=rng.pos,
               root_sol=np.array([1, kinds[t.root.solution[0]], t.root.solution[1]] if t.root.solution else [0, 0, 0], np.int32))
    for ch in t.root.children:
        out["child_N"][ch.action], out["child_cum"][ch.action], out["child_P"][ch.action] = ch.visits, ch.cum, ch.prob
        if ch.solution:
            out["child_sol"][ch.action] = [1, kinds[ch.solution[0]], ch.solution[1]]
    return out
