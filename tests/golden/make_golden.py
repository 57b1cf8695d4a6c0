#!/usr/bin/env python3
"""Generates the fixtures in tests/golden/ (run once in the authoring container; /root/reference is NOT
available on the GPU box, so only the JSON outputs of this script travel).

What it extracts — data only, never source text:
  * slimnn known-answer vectors: the literal weight / bias / input / expected arrays of
      slimnn/src/conv.rs:92-602 (6 Conv2d tests), slimnn/src/linear.rs:105-112, slimnn/src/activations.rs:70-75
  * Connect4 scripted games: the move sequences and asserted results of study-connect4/src/connect4.rs:299-447
  * the Outcome ordering truth table asserted in synthesis/src/game.rs:94-141
  * the TicTacToe MCTS known answers asserted in synthesis/src/mcts.rs:691-868
  * an independent second opinion for the Connect4Net MLP (no test exists for it in the reference): random
    weights + random reachable positions evaluated with this container's torch (float32 and float64).
"""
import ast
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def rust_array(text):
    """'[[1., -2.5], [3., 4.]]' (Rust literal) -> nested python lists."""
    text = re.sub(r"//[^\n]*", "", text)
    return ast.literal_eval(re.sub(r"\s+", " ", text))


def balanced(src, start):
    """Returns the substring of the bracketed expression starting at src[start] == '['."""
    depth = 0
    for i in range(start, len(src)):
        if src[i] == "[":
            depth += 1
        elif src[i] == "]":
            depth -= 1
            if depth == 0:
                return src[start : i + 1]
    raise ValueError("unbalanced")


def grab(body, marker):
    i = body.index(marker) + len(marker)
    j = body.index("[", i)
    return rust_array(balanced(body, j))


def conv_kats():
    src = open(f"{REF}/slimnn/src/conv.rs").read()
    tests = src[src.index("#[cfg(test)]") :]
    out = []
    for m in re.finditer(r"fn (test_\w+)\(\)", tests):
        name = m.group(1)
        nxt = tests.find("#[test]", m.end())
        body = tests[m.end() : nxt if nxt > 0 else len(tests)]
        g = re.search(r"Conv2d<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", body)
        cin, cout, k, rp, cp, s = map(int, g.groups())
        f = re.search(r"forward::<(\d+), (\d+), (\d+), (\d+)>", body)
        w_in, h_in, w_out, h_out = map(int, f.groups())
        out.append(
            dict(
                name=name, source=f"slimnn/src/conv.rs::{name}",
                cin=cin, cout=cout, k=k, row_pad=rp, col_pad=cp, stride=s,
                w_in=w_in, h_in=h_in, w_out=w_out, h_out=h_out,
                weight=grab(body, "conv.weight ="), bias=grab(body, "conv.bias ="),
                x=grab(body, "let x ="), expected=grab(body, "let t ="),
                tolerance=1e-6,
            )
        )
    return out


def linear_kat():
    src = open(f"{REF}/slimnn/src/linear.rs").read()
    body = src[src.index("fn test_linear") :]
    w = grab(body, "q.weight =")
    b = grab(body, "q.bias =")
    cases = []
    for m in re.finditer(r"q\.forward\(&(\[[^\]]*\])\), (\[[^\]]*\])", body):
        cases.append(dict(x=rust_array(m.group(1)), expected=rust_array(m.group(2))))
    return dict(source="slimnn/src/linear.rs:105-112", weight=w, bias=b, cases=cases)


def relu_kat():
    src = open(f"{REF}/slimnn/src/activations.rs").read()
    body = src[src.index("fn test_relu_1d") :]
    x = grab(body, "let x =")
    m = re.search(r"assert_eq!\(y, (\[[^\]]*\])\)", body)
    return dict(source="slimnn/src/activations.rs:70-75", x=x, expected=rust_array(m.group(1)))


def connect4_games():
    src = open(f"{REF}/study-connect4/src/connect4.rs").read()
    tests = src[src.index("#[cfg(test)]") :]
    games = []
    for name in ("test_first_wins", "test_second_wins", "test_draw"):
        i = tests.index(f"fn {name}")
        nxt = tests.find("#[test]", i)
        body = tests[i:nxt]
        steps = []
        # sequence of events in source order: step asserts and iter_actions position asserts
        for m in re.finditer(
            r"assert!\((!?)game\.step\(&Column\((\d+)\)\)\);|"
            r"assert!\(game\.iter_actions\(\)\.position\(\|c\| c == Column\((\d+)\)\)\.is_(some|none)\(\)\);",
            body,
        ):
            if m.group(2) is not None:
                steps.append(dict(op="step", col=int(m.group(2)), is_over=(m.group(1) != "!")))
            else:
                steps.append(dict(op="legal", col=int(m.group(3)), expected=(m.group(4) == "some")))
        final = {}
        w = re.search(r"assert_eq!\(game\.winner\(\), (Some\(PlayerId::(\w+)\)|None)\)", body)
        final["winner"] = w.group(2) if w.group(2) else None
        p = re.search(r"assert_eq!\(game\.player\(\), PlayerId::(\w+)\)", body)
        final["player"] = p.group(1) if p else None
        for color in ("Red", "Black"):
            r = re.search(rf"assert_eq!\(game\.reward\(PlayerId::{color}\), (-?[\d.]+)\)", body)
            final[f"reward_{color.lower()}"] = float(r.group(1)) if r else None
        r = re.search(r"assert_eq!\(game\.reward\(game\.player\(\)\), (-?[\d.]+)\)", body)
        final["reward_to_move"] = float(r.group(1)) if r else None
        final["is_over"] = "assert!(game.is_over())" in body
        games.append(dict(name=name, source=f"study-connect4/src/connect4.rs::{name}", events=steps, final=final))
    # the four exhaustive won() sweeps (connect4.rs:449-498) restated as explicit bitboard lists
    wins = []
    for row in range(7):
        bb = (1 << row) | (1 << (row + 7)) | (1 << (row + 14)) | (1 << (row + 21))
        for _ in range(6):
            wins.append(dict(kind="horz", bb=bb))
            bb <<= 7
    for col in range(9):
        bb = sum(1 << (7 * col + i) for i in range(4))
        for _ in range(4):
            wins.append(dict(kind="vert", bb=bb))
            bb <<= 1
    for row in range(3, 7):
        bb = (1 << row) | (1 << (row + 6)) | (1 << (row + 12)) | (1 << (row + 18))
        for _ in range(6):
            wins.append(dict(kind="d1", bb=bb))
            bb <<= 7
    for col in range(6):
        bb = sum(1 << (7 * (col + i) + i) for i in range(4))
        for _ in range(4):
            wins.append(dict(kind="d2", bb=bb))
            bb <<= 1
    # sanity: the loops above must be the loops of the reference tests (checked structurally)
    for needle in ("for row in 0..HEIGHT", "for col in 0..WIDTH", "for row in 3..HEIGHT", "for col in 0..6"):
        assert needle in tests, needle
    return dict(games=games, won_bitboards=wins)


def outcome_table():
    src = open(f"{REF}/synthesis/src/game.rs").read()
    tests = src[src.index("#[cfg(test)]") :]
    rows = []
    sym = {"Equal": 0, "Greater": 1, "Less": -1}
    for m in re.finditer(r"Outcome::(\w+)\(0\)\.cmp\(&Outcome::(\w+)\(0\)\), Ordering::(\w+)", tests):
        rows.append(dict(a=m.group(1), b=m.group(2), cmp=sym[m.group(3)]))
    opt = []
    for m in re.finditer(r"assert!\(Some\(Outcome::(\w+)\(0\)\) (==|>|<) (None|Some\(Outcome::(\w+)\(0\)\))\);", tests):
        opt.append(dict(a=m.group(1), op=m.group(2), b=m.group(4)))
    assert len(rows) == 9 and len(opt) == 12, (len(rows), len(opt))
    return dict(source="synthesis/src/game.rs:94-141", cmp=rows, option=opt)


def ttt_kats():
    src = open(f"{REF}/synthesis/src/mcts.rs").read()
    out = []
    for which, name in enumerate(("test_solve_win", "test_solve_loss", "test_solve_draw")):
        i = src.index(f"fn {name}")
        nxt = src.find("#[test]", i)
        body = src[i:nxt]
        moves = [int(r) * 3 + int(c) for r, c in re.findall(r"game\.step\(&Action \{ row: (\d), col: (\d) \}\);", body)]
        active = "\n".join(l for l in body.splitlines() if not l.strip().startswith("//"))
        sol_none = [int(a) for a in re.findall(r"assert_eq!\(mcts\.solution\(&(\d)\.into\(\)\), None\);", active)]
        best = re.search(r"assert_eq!\(mcts\.best_action\(ActionSelection::Q\), (\d)\.into\(\)\);", active)
        nodes = re.search(r"assert_eq!\(mcts\.nodes\.len\(\), (\d+)\);", active)
        seed = re.search(r"StdRng::seed_from_u64\((\d+)\)", body)
        c = re.search(r"PolynomialUct \{ c: ([\d.]+) \}", body)
        out.append(
            dict(
                which=which, name=name, source=f"synthesis/src/mcts.rs::{name}", moves=moves,
                seed=int(seed.group(1)), c=float(c.group(1)),
                solution_none_actions=sol_none,
                best_action_q=int(best.group(1)) if best else None,
                nodes_len=int(nodes.group(1)),
            )
        )
    return out


def mlp_goldens():
    import torch

    dims = [63, 128, 96, 64, 48, 12]
    rng = np.random.RandomState(20211003)
    blob = []
    layers = []
    for i in range(5):
        bound = 1.0 / np.sqrt(dims[i])
        W = rng.uniform(-bound, bound, size=(dims[i + 1], dims[i])).astype(np.float32)
        b = rng.uniform(-bound, bound, size=(dims[i + 1],)).astype(np.float32)
        layers.append((W, b))
        blob += [W.ravel(), b.ravel()]
    blob = np.concatenate(blob)
    assert blob.size == 30492

    # random reachable positions: play random legal moves from the empty board (python mirror of the bit layout)
    def won(bb):
        def run(s, mask):
            return bb & (bb >> s) & (bb >> 2 * s) & (bb >> 3 * s) & mask
        fab_row = sum(1 << (7 * c) for c in range(9))
        cols05 = sum(0x7F << (7 * c) for c in range(6))
        rows = lambda rs: sum(fab_row << r for r in rs)
        return bool(run(6, cols05 & rows([3, 4, 5, 6])) | run(8, cols05 & rows([0, 1, 2, 3])) | run(7, cols05)
                    | run(1, rows([0, 1, 2, 3])))

    positions = []
    while len(positions) < 64:
        my = op = 0
        h = [0] * 9
        nmoves = rng.randint(0, 40)
        ok = True
        for _ in range(nmoves):
            legal = [c for c in range(9) if h[c] < 7]
            if not legal:
                ok = False
                break
            c = legal[rng.randint(len(legal))]
            my ^= 1 << (h[c] + 7 * c)
            h[c] += 1
            my, op = op, my
            if won(op):
                ok = False
                break
        if ok:
            positions.append((my, op, list(h)))

    def features(my, op, h):
        s = np.zeros((7, 9), dtype=np.float32)
        for row in range(7):
            for col in range(9):
                idx = 1 << (row + 7 * col)
                s[row, col] = 1.0 if my & idx else (-1.0 if op & idx else -0.1)
        for col in range(9):
            if h[col] < 7:
                s[h[col], col] = 0.1
        return s.reshape(63)

    X = np.stack([features(*p) for p in positions])

    def fwd(dtype):
        x = torch.tensor(X, dtype=dtype)
        for i, (W, b) in enumerate(layers):
            x = torch.nn.functional.linear(x, torch.tensor(W, dtype=dtype), torch.tensor(b, dtype=dtype))
            if i < 4:
                x = torch.relu(x)
        return x[:, :9].numpy(), torch.softmax(x[:, 9:], dim=-1).numpy()

    l32, v32 = fwd(torch.float32)
    l64, v64 = fwd(torch.float64)
    np.save(os.path.join(OUT, "c4net_blob_f32.npy"), blob)
    return dict(
        source="torch %s nn.functional.linear/relu/softmax on study-connect4/src/policies.rs:20-44 shapes" % torch.__version__,
        weights_file="c4net_blob_f32.npy",
        my_bb=[int(p[0]) for p in positions], op_bb=[int(p[1]) for p in positions],
        features=X.tolist(),
        logits_f32=l32.tolist(), value_f32=v32.tolist(), logits_f64=l64.tolist(), value_f64=v64.tolist(),
    )


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not available; fixtures are already committed")
    dump = lambda name, obj: json.dump(obj, open(os.path.join(OUT, name), "w"), indent=1)
    dump("slimnn_conv_kats.json", conv_kats())
    dump("slimnn_linear_relu_kats.json", dict(linear=linear_kat(), relu=relu_kat()))
    dump("connect4_kats.json", connect4_games())
    dump("outcome_kats.json", outcome_table())
    dump("tictactoe_mcts_kats.json", ttt_kats())
    dump("c4net_torch_goldens.json", mlp_goldens())
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
