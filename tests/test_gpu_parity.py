"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star):
  * integer / index / visit-count work: bit-exact (np.array_equal on f32 visit counts, W sums, priors, solutions);
  * policy/value network: within 1e-5 of the canonical slimnn-order oracle, and EXACTLY equal to the oracle's ACC_FMA
    mode (the fused-multiply-add chain the f32 MFMA computes), which is what makes visit-exact MCTS parity testable.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NN_TOL = 1e-5  # north_star: "policy/value outputs within 1e-5 fp32"


@pytest.fixture(scope="module")
def blob(golden_dir):
    return np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))


@pytest.fixture(scope="module")
def engine(blob):
    import synthesis_amd as sa

    eng = sa.Engine(concurrent_games=256, max_explores=800, device=0)
    eng.load_weights(blob)
    yield eng
    eng.close()


def random_positions(oracle, n, seed, max_moves=45):
    """Random reachable non-terminal positions (uniform random legal playouts of random length)."""
    rng = np.random.RandomState(seed)
    my, op = [], []
    while len(my) < n:
        moves = []
        h = [0] * 9
        ok = True
        for _ in range(rng.randint(0, max_moves)):
            legal = [c for c in range(9) if h[c] < 7]
            if not legal:
                ok = False
                break
            c = legal[rng.randint(len(legal))]
            moves.append(c)
            h[c] += 1
        r = oracle.c4_play(moves) if moves else dict(over=np.zeros(0, bool), my_bb=0, op_bb=0)
        if ok and not np.any(r["over"]):
            my.append(r["my_bb"])
            op.append(r["op_bb"])
    return np.array(my, np.uint64), np.array(op, np.uint64)


# ------------------------------------------------------------------------------------------------ primitives
def test_device_math_is_ieee_exact(engine, oracle):
    rng = np.random.RandomState(1)
    a = np.concatenate([rng.uniform(-104, 1, 200000), rng.uniform(0, 2000, 100000), [0.0, -0.0, 1.0, 800.0, 1601.0]]).astype(np.float32)
    b = np.concatenate([rng.uniform(0.5, 1700, 300000), [1.0, 3.0, 7.0, 1601.0, 0.1]]).astype(np.float32)
    e, d, s = engine.debug_math(a, b)
    assert np.array_equal(e.view(np.uint32), oracle.det_expf(a).view(np.uint32))  # det_expf bit for bit
    assert np.array_equal(d.view(np.uint32), (a / b).view(np.uint32))            # correctly rounded divide
    pos = a >= 0
    assert np.array_equal(s[pos].view(np.uint32), np.sqrt(a[pos]).view(np.uint32))  # correctly rounded sqrt


def test_slimnn_activation_layers_match_oracle(engine, oracle):
    """slimnn/src/activations.rs:31-63 as device layers: ReLU, Tanh (deterministic det_tanhf shared with the oracle) and the
    un-stabilised Softmax::apply_1d (exp / sum in index order, no max subtraction) — bit for bit the oracle's."""
    rng = np.random.RandomState(3)
    x = np.concatenate([rng.uniform(-6, 6, 40000), rng.uniform(-60, 60, 20000), [0.0, -0.0, 0.625, -0.625, 44.5, -44.5]]).astype(np.float32)
    assert np.array_equal(engine.activation(1, x).view(np.uint32), oracle.tanh(x).view(np.uint32))
    assert np.array_equal(engine.activation(0, x), np.maximum(x, np.float32(0)))
    assert np.abs(engine.activation(1, x) - np.tanh(x.astype(np.float64))).max() < 2.5e-7
    rows = rng.uniform(-5, 5, (500, 12)).astype(np.float32)
    rows[7] = [100.0] + [0.0] * 11   # overflow: inf / inf = NaN in the reference too
    got = engine.activation(2, rows)
    ref = np.stack([oracle.softmax_slimnn(r) for r in rows])
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(got[7]).any()   # (NaN sign/payload is not compared)
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok].view(np.uint32), ref[ok].view(np.uint32))
    with pytest.raises(Exception):
        engine.activation(7, rows)


def test_descent_division_and_square_root_are_ieee_exact_by_enumeration(engine):
    """The two shortened sequences of the descent (device_common.cuh) against the device's IEEE operations over EVERY f32
    significand: div2_by_small_int (one residual correction) for every integer divisor 1 .. 8,192 — beyond any 1 + n a tree can hold
    (max_explores <= 7,280) — at three exponents (2.1e11 quotients), plus divisors up to 2^16 in blocks; div2_safe_range (two
    corrections: the softmax sums' form) on the same operands; sqrt_normal_range at both exponent parities and near the ends of its
    range (3.4e7 roots). Not one bit differs."""
    assert engine.debug_small_int_math(1, 8192) == (0, 0, 0)
    for lo in (8193, 20000, 32768, 65000):
        assert engine.debug_small_int_math(lo, lo + 535) == (0, 0, 0)


def test_packed_division_is_ieee_exact_on_its_range(engine):
    """device_common.cuh div2_by_small_int (two quotients per v_pk_fma stream, used by the descent's explore_value): equal to the
    IEEE quotient bit for bit for a = 0 or 2^-60 <= a <= 2^60 and integer-valued 1 <= b <= 2^16 — the only range it is used on
    (numerator c * prior * sqrt(N), denominator 1 + n; priors below 2^-40 are flagged and take the full division)."""
    rng = np.random.RandomState(7)
    n = 4_000_000
    a = np.exp2(rng.uniform(-60, 60, n)).astype(np.float32)
    a[: n // 4] = (rng.uniform(0, 1, n // 4) * rng.uniform(1, 30, n // 4) * 3.0).astype(np.float32)   # the descent's own range
    a[:1000] = 0.0
    b = rng.randint(1, 802, n).astype(np.float32)
    b[n // 2:] = rng.randint(1, 65537, n - n // 2).astype(np.float32)
    fast, full = engine.debug_fast_div(a, b)
    assert np.array_equal(full.view(np.uint32), (a / b).view(np.uint32))
    assert np.array_equal(fast.view(np.uint32), full.view(np.uint32))


def test_device_logf_matches_oracle(engine, oracle):
    a = np.concatenate([np.arange(1, 5000), np.random.RandomState(2).uniform(1e-30, 1e30, 100000)]).astype(np.float32)
    _, _, s = engine.debug_math(a, -np.ones_like(a))
    assert np.array_equal(s.view(np.uint32), oracle.det_logf(a).view(np.uint32))
    assert oracle.det_logf([1.0])[0] == 0.0
    np.testing.assert_allclose(oracle.det_logf(a).astype(np.float64), np.log(a.astype(np.float64)), rtol=3e-7, atol=1e-7)


def test_device_stdrng_matches_oracle(engine, oracle):
    for seed in (0, 1, 12345, 2**63 + 17):
        assert np.array_equal(engine.debug_stdrng_u32(seed, 200), oracle.stdrng_u32(seed, 200))


# ------------------------------------------------------------------------------------------------ leaf evaluation
def test_features_match_oracle(engine, oracle):
    my, op = random_positions(oracle, 300, seed=2, max_moves=63)
    assert np.array_equal(engine.features(my, op), oracle.c4_features(my, op))


def test_policy_eval_parity(engine, oracle, blob):
    my, op = random_positions(oracle, 1000, seed=3)
    logits, value = engine.policy_eval(my, op)
    ref_l, ref_v = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_SLIMNN)
    assert np.abs(logits - ref_l).max() < NN_TOL and np.abs(value - ref_v).max() < NN_TOL
    fma_l, fma_v = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_FMA)
    assert np.array_equal(logits, fma_l)   # MFMA == k-ordered fmaf chain
    assert np.array_equal(value, fma_v)


def test_policy_eval_torch_goldens_and_ragged_sizes(engine, golden_dir, oracle, blob):
    g = json.load(open(os.path.join(golden_dir, "c4net_torch_goldens.json")))
    logits, value = engine.policy_eval(g["my_bb"], g["op_bb"])
    assert np.abs(logits - np.array(g["logits_f64"])).max() < NN_TOL
    assert np.abs(value - np.array(g["value_f64"])).max() < NN_TOL
    my, op = random_positions(oracle, 70, seed=4)
    full = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_FMA)
    for n in (0, 1, 15, 16, 17, 33, 70):  # empty, partial tiles, tile boundaries
        l, v = engine.policy_eval(my[:n], op[:n])
        assert l.shape == (n, 9) and np.array_equal(l, full[0][:n]) and np.array_equal(v, full[1][:n])


def test_policy_eval_large_batch_consistency(engine, oracle, blob):
    """Full-size property: 65,536 positions (4096 games x 16) — every tile of the grid-stride loop gives the same
    answer as the first, and a sample is checked against the oracle."""
    my, op = random_positions(oracle, 512, seed=5)
    big_my, big_op = np.tile(my, 128), np.tile(op, 128)
    logits, value = engine.policy_eval(big_my, big_op)
    assert np.array_equal(logits.reshape(128, 512, 9), np.broadcast_to(logits[:512], (128, 512, 9)))
    assert np.array_equal(value.reshape(128, 512, 3), np.broadcast_to(value[:512], (128, 512, 3)))
    fl, fv = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_FMA)
    assert np.array_equal(logits[:512], fl) and np.array_equal(value[:512], fv)


def test_policy_eval_transfer_paths_and_contexts(engine, oracle, blob):
    """syn_policy_eval_batch reads the positions in pinned host memory in place and, by batch size, writes the results in place
    (<= 4,096), fetches them with one DMA (<= 32,768) or uses the pageable transfers (beyond): every boundary gives the oracle's
    bits. Evaluation contexts (syn_eval_ctx: one worker's policy, alpha_zero.rs:192-198) give the same bits — one batch in flight
    each, four of them from four host threads at once, beside the engine's own call."""
    import threading
    import synthesis_amd as sa

    my, op = random_positions(oracle, 4099, seed=12)
    fl, fv = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_FMA)
    reps = 9   # 36,891 positions: past the pinned path's end
    big_my, big_op = np.tile(my, reps), np.tile(op, reps)
    big_l, big_v = np.tile(fl, (reps, 1)), np.tile(fv, (reps, 1))
    for n in (4096, 4097, 20000, 32768, 32769, len(big_my)):
        l, v = engine.policy_eval(big_my[:n], big_op[:n])
        assert np.array_equal(l.view(np.uint32), big_l[:n].view(np.uint32)) and np.array_equal(v.view(np.uint32), big_v[:n].view(np.uint32)), n
    ctxs = [engine.eval_context() for _ in range(4)]
    for n in (0, 1, 17, 4096, 4097, 33000):
        l, v = ctxs[0].eval(big_my[:n], big_op[:n])
        assert l.shape == (n, 9) and np.array_equal(l.view(np.uint32), big_l[:n].view(np.uint32)) and np.array_equal(v.view(np.uint32), big_v[:n].view(np.uint32))
    # submit ... wait with other work on the engine's own stream in between; a second submit before the wait is refused
    ctxs[1].submit(my[:700], op[:700])
    with pytest.raises(sa.SynthesisAmdError) as e:
        ctxs[1].submit(my[:5], op[:5])
    assert e.value.code == -1 and "waited" in str(e.value)
    l0, v0 = engine.policy_eval(my[:300], op[:300])
    l1, v1 = ctxs[1].wait()
    assert np.array_equal(l1, fl[:700]) and np.array_equal(v1, fv[:700]) and np.array_equal(l0, fl[:300]) and np.array_equal(v0, fv[:300])
    # a context of an engine without weights: the engine's error code, the text in the context's own slot; wait() without a batch: nothing
    bare = sa.Engine(concurrent_games=64, max_explores=8)
    c0 = bare.eval_context()
    with pytest.raises(sa.SynthesisAmdError) as e:
        c0.eval(my[:3], op[:3])
    assert e.value.code == -4 and "syn_load_weights" in str(e.value)
    with pytest.raises(sa.SynthesisAmdError):
        c0.wait()   # the submission failed: there is no batch to wait for (and no zeros posing as results)
    bare.load_weights(blob)
    l, v = c0.eval(my[:3], op[:3])
    assert np.array_equal(l, fl[:3]) and np.array_equal(v, fv[:3])
    with pytest.raises(sa.SynthesisAmdError):
        c0.wait()   # ... nor a second time for a batch already collected
    bare.close()   # (closes its contexts first)
    errors = []

    def worker(k):
        try:
            rs = np.random.RandomState(k)
            for _ in range(150):
                lo = int(rs.randint(0, 3000)); n = int(rs.randint(1, 1000))
                l, v = ctxs[k].eval(my[lo:lo + n], op[lo:lo + n])
                if not (np.array_equal(l, fl[lo:lo + n]) and np.array_equal(v, fv[lo:lo + n])):
                    errors.append((k, lo, n))
        except Exception as ex:   # noqa: BLE001
            errors.append((k, repr(ex)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for _ in range(50):   # the engine's own entry point beside them
        l, v = engine.policy_eval(my[:513], op[:513])
        assert np.array_equal(l, fl[:513]) and np.array_equal(v, fv[:513])
    for t in threads:
        t.join()
    assert errors == []
    for c in ctxs:
        c.close()


def test_a_new_context_is_ready_for_its_first_call_while_the_null_stream_is_busy(engine, oracle, blob):
    """Contexts created and used at once, 150 times, while torch's default stream is kept busy. Background: a context's workgroup
    counter used to be zeroed with hipMemset on the null stream — asynchronous to the host and not ordered against the context's
    non-blocking stream — and inside the full GPU suite (never in a test alone, and not under this load either) the zeroing landed
    in the context's first kernel: "call 1, workgroups counted 0", the completion word never written. The zeroing is now
    stream-ordered; the full suite is the regression test, this one keeps the create -> first call path under load."""
    import torch

    my, op = random_positions(oracle, 128, seed=21)
    fl, fv = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_FMA)
    x = torch.randn(4096, 4096, device="cuda")
    for _ in range(150):
        y = x @ x
        y = y @ x   # a few milliseconds of work queued on the (blocking) default stream
        c = engine.eval_context()
        l, v = c.eval(my, op)
        assert np.array_equal(l, fl) and np.array_equal(v, fv)
        c.close()
    torch.cuda.synchronize()
    del x, y


def test_a_missing_completion_word_costs_time_not_answers(engine, oracle, blob, monkeypatch):
    """The polled completion's way out: with the context's workgroup counter started off wrong (debug knob) the kernel never writes
    the completion word; syn_eval_ctx_wait gives up polling after ~2 ms, drains the stream — the kernel has finished, its results
    are complete — returns them, leaves a note in the context's error slot and waits on the stream from then on."""
    my, op = random_positions(oracle, 300, seed=23)
    fl, fv = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_FMA)
    monkeypatch.setenv("SYN_DEBUG", "1")
    monkeypatch.setenv("SYN_EVAL_BREAK_COUNTER", "1")
    c = engine.eval_context()
    monkeypatch.delenv("SYN_EVAL_BREAK_COUNTER")
    for n in (128, 5, 300, 128):
        l, v = c.eval(my[:n], op[:n])
        assert np.array_equal(l, fl[:n]) and np.array_equal(v, fv[:n])
    note = (engine._lib.syn_eval_ctx_last_error(c._c) or b"").decode()
    assert "without reporting completion" in note and "call 1" in note
    c.close()
    c2 = engine.eval_context()   # a healthy context leaves no note
    l, v = c2.eval(my[:128], op[:128])
    assert np.array_equal(l, fl[:128]) and (engine._lib.syn_eval_ctx_last_error(c2._c) or b"") == b""
    c2.close()


# ------------------------------------------------------------------------------------------------ slimnn layers
def test_slimnn_layer_kats_on_gpu(engine, oracle, golden_dir):
    """slimnn/src/conv.rs:92-602, linear.rs:105-112 through the GPU kernels; also bit-exact vs the oracle's slimnn mode."""
    for k in json.load(open(os.path.join(golden_dir, "slimnn_conv_kats.json"))):
        W = np.array(k["weight"], np.float32)
        x = np.array(k["x"], np.float32)
        y = engine.conv2d(W, k["bias"], x, k["row_pad"], k["col_pad"], k["stride"])[0]
        assert np.abs(y - np.array(k["expected"], np.float32)).max() < k["tolerance"], k["name"]
        assert np.array_equal(y, oracle.conv2d(W, k["bias"], x, k["row_pad"], k["col_pad"], k["stride"])[0])
    lk = json.load(open(os.path.join(golden_dir, "slimnn_linear_relu_kats.json")))
    for case in lk["linear"]["cases"]:
        assert engine.linear(lk["linear"]["weight"], lk["linear"]["bias"], case["x"])[0].tolist() == case["expected"]
    rng = np.random.RandomState(6)
    W = rng.randn(96, 128).astype(np.float32); b = rng.randn(96).astype(np.float32); x = rng.randn(37, 128).astype(np.float32)
    assert np.array_equal(engine.linear(W, b, x), oracle.linear(W, b, x))
    assert np.array_equal(engine.linear(W, b, x, relu=True), np.maximum(oracle.linear(W, b, x), 0))
    Wc = rng.randn(8, 2, 3, 3).astype(np.float32); bc = rng.randn(8).astype(np.float32); xc = rng.randn(5, 2, 7, 9).astype(np.float32)
    assert np.array_equal(engine.conv2d(Wc, bc, xc, 1, 0, 3), oracle.conv2d(Wc, bc, xc, 1, 0, 3))
    assert np.array_equal(engine.conv2d(Wc, bc, xc, 1, 1, 1), oracle.conv2d(Wc, bc, xc, 1, 1, 1))
    # the LDS-tiled kernels (layer_kernels.cuh) take over from 64 samples: ragged last tiles, narrow and wide layers, strides
    for (O, I, n) in ((128, 63, 1000), (12, 48, 777), (96, 128, 64), (5, 7, 4097)):
        W = rng.randn(O, I).astype(np.float32); b = rng.randn(O).astype(np.float32); x = rng.randn(n, I).astype(np.float32)
        assert np.array_equal(engine.linear(W, b, x), oracle.linear(W, b, x)), (O, I, n)
        assert np.array_equal(engine.linear(W, b, x, relu=True), np.maximum(oracle.linear(W, b, x), 0)), (O, I, n)
    xc = rng.randn(333, 2, 7, 9).astype(np.float32)
    for (rp, cp, st) in ((1, 1, 1), (1, 0, 3), (0, 0, 2), (2, 1, 1)):
        assert np.array_equal(engine.conv2d(Wc, bc, xc, rp, cp, st), oracle.conv2d(Wc, bc, xc, rp, cp, st)), (rp, cp, st)
    W5 = rng.randn(3, 4, 5, 5).astype(np.float32); b5 = rng.randn(3).astype(np.float32); x5 = rng.randn(70, 4, 11, 6).astype(np.float32)
    assert np.array_equal(engine.conv2d(W5, b5, x5, 2, 2, 1, relu=True), np.maximum(oracle.conv2d(W5, b5, x5, 2, 2, 1), 0))
    my, op = random_positions(oracle, 1001, seed=9)   # 63 * 1001 floats: not a multiple of the 4-float store groups
    assert np.array_equal(engine.features(my, op), oracle.c4_features(my, op))
    import synthesis_amd as sa
    with pytest.raises(sa.SynthesisAmdError):  # conv.rs:50-51 asserts on inconsistent output dims
        engine.conv2d(Wc, bc, xc, 1, 0, 3, out_hw=(4, 3))


# ------------------------------------------------------------------------------------------------ MCTS
SEARCH_KEYS = ("child_N", "child_W", "child_P", "child_sol", "root_stat", "root_sol", "num_nodes", "best_action",
               "target_pi", "target_q")


def assert_search_equal(got, ref, ctx=""):
    for k in SEARCH_KEYS:
        a, b = np.asarray(got[k]), np.asarray(ref[k])
        assert a.shape == b.shape, (ctx, k, a.shape, b.shape)
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            raise AssertionError(f"{ctx}: {k} differs at {bad[:5].tolist()} got {a[tuple(bad[0])]} ref {b[tuple(bad[0])]}")


def test_mcts_search_visit_exact(engine, oracle, blob):
    """Visit counts, W sums, priors, solver marks, node counts, best action and both targets are identical to the
    oracle MCTS on the same roots (parity config of study-connect4/src/main.rs:58-66)."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config

    my, op = random_positions(oracle, 96, seed=7)
    my[0] = 0; op[0] = 0  # the empty board
    for explores in (0, 1, 37, 200):
        got = engine.mcts_search(sa.parity_mcts_config(), my, op, explores)
        ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, explores, nn_mode=oracle.ACC_FMA)
        assert_search_equal(got, ref, f"explores={explores}")
    got = engine.mcts_search(sa.parity_mcts_config(), my[:24], op[:24], 800)
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my[:24], op[:24], 800, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, "explores=800")
    assert got["num_nodes"].max() <= 1 + 9 * 801


@pytest.mark.parametrize("shape", ["rows", "lanes8", "lanes16", "pc2"])
def test_trained_checkpoint_deep_trees_match_oracle(golden_dir, oracle, monkeypatch, shape):
    """A TRAINED network (tests/golden/c4net_trained_f32.npy, produced on the device by examples/train_connect4.py; see
    tests/golden/README.md) makes sharp priors, hence deep narrow trees with long backprop paths and near-ties in PUCT — the
    regime the random-init blob never reaches. Searches at 800 explores and whole self-play games stay identical to
    the oracle on every launch shape, and the trees really are deeper than with the random-init network."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    path = os.path.join(golden_dir, "c4net_trained_f32.npy")
    if not os.path.exists(path):
        pytest.skip("no trained checkpoint fixture yet")
    trained = np.load(path)
    if shape != "rows":
        monkeypatch.setenv("SYN_DEBUG", "1")  # developer knobs are honoured only with SYN_DEBUG=1
        if shape == "pc2":
            from tests.conftest import debug_shapes_built
            if not debug_shapes_built():
                pytest.skip("library built without DEBUG_SHAPES=1")
            monkeypatch.setenv("SYN_PC", "2")
        else:
            monkeypatch.setenv("SYN_LANES", shape[5:])
    eng = sa.Engine(concurrent_games=1100, max_explores=800)
    eng.load_weights(trained)
    my, op = random_positions(oracle, 64, seed=19, max_moves=40)
    my[0] = 0; op[0] = 0
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 800)
    ref = oracle.c4_mcts_search(parity_mcts_config(), trained, my, op, 800, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, f"trained weights, {shape}, 800 explores")
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=3, n_games=8, counters=True)
    ref = oracle.c4_selfplay(parity_rollout_config(800), trained, 3, 8, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, f"trained weights, {shape}, self-play")
    for k in ("explores", "select_levels", "backprop_levels", "policy_evals", "max_depth"):
        assert got["counters"][k] == ref["counters"][k], k
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    rnd = oracle.c4_selfplay(parity_rollout_config(800), blob, 3, 8, threads=8, nn_mode=oracle.ACC_FMA)["counters"]
    deep, shallow = got["counters"]["select_levels"] / got["counters"]["explores"], rnd["select_levels"] / rnd["explores"]
    assert deep > shallow + 0.5, (deep, shallow)
    eng.close()


def test_mcts_search_reference_default_1600_explores(oracle, blob):
    """The reference's own default budget (num_explores: 1600, study-connect4/src/main.rs:30): slab of 1 + 9*1601 nodes
    per tree, still visit-exact; plus one full self-play game at 1600 explores."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    eng = sa.Engine(concurrent_games=64, max_explores=1600)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 12, seed=41, max_moves=30)
    my[0] = 0; op[0] = 0
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 1600)
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 1600, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, "explores=1600")
    assert got["num_nodes"].max() <= 1 + 9 * 1601 and got["num_nodes"].max() > 7210
    sp = eng.selfplay(sa.parity_rollout_config(1600), base_seed=9, n_games=2)
    rf = oracle.c4_selfplay(parity_rollout_config(1600), blob, 9, 2, threads=2, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(sp, rf, "1600 explores")
    eng.close()


def test_mcts_search_late_game_solver_and_auto_extend(engine, oracle, blob):
    """Nearly full boards: few legal columns (auto-extend chains), many terminal children (solver + value correction),
    roots that get solved before the explore budget is used."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config

    rng = np.random.RandomState(8)
    my, op = [], []
    while len(my) < 64:
        h = [0] * 9
        moves = []
        target = rng.randint(50, 62)
        for _ in range(target):
            legal = [c for c in range(9) if h[c] < 7]
            c = legal[rng.randint(len(legal))]
            moves.append(c); h[c] += 1
        r = oracle.c4_play(moves)
        if not r["over"].any():
            my.append(r["my_bb"]); op.append(r["op_bb"])
    my = np.array(my, np.uint64); op = np.array(op, np.uint64)
    got = engine.mcts_search(sa.parity_mcts_config(), my, op, 300)
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 300, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, "late game")
    assert ref["root_sol"][:, 0].sum() > 10  # the case really exercises solved roots


def test_mcts_config_variants(engine, oracle, blob):
    """Remaining deterministic MCTSConfig variants: Fpu::ParentQ, no solver, no value correction, no auto-extend,
    select_solved_nodes = false, ActionSelection::Q, other c, PolicyNoise::Equal (mcts.rs:258-269)."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config

    my, op = random_positions(oracle, 48, seed=9, max_moves=60)
    variants = [
        dict(fpu=1), dict(solve=0), dict(correct_values_on_solve=0), dict(auto_extend=0), dict(select_solved_nodes=0),
        dict(c=1.25, fpu_value=0.0), dict(noise=1, noise_weight=0.25), dict(noise=1, noise_weight=0.6, fpu=1),
        # Exploration::Uct as the reference configures its rollout MCTS (study-connect4/src/main.rs:74-82)
        dict(exploration=0, c=2.0, auto_extend=0, fpu_value=float("inf")), dict(exploration=0, c=0.7, fpu_value=0.5),
    ]
    for v in variants:
        ocfg = parity_mcts_config(**v)
        scfg = sa.MCTSConfig(exploration=sa.Exploration(ocfg.exploration), c=ocfg.c, solve=bool(ocfg.solve),
                             correct_values_on_solve=bool(ocfg.correct_values_on_solve),
                             select_solved_nodes=bool(ocfg.select_solved_nodes), auto_extend=bool(ocfg.auto_extend),
                             fpu=sa.Fpu(ocfg.fpu), fpu_value=ocfg.fpu_value,
                             root_policy_noise=sa.PolicyNoise(ocfg.noise), noise_weight=ocfg.noise_weight)
        for sel in (0, 1):
            got = engine.mcts_search(scfg, my, op, 150, action_selection=sel)
            ref = oracle.c4_mcts_search(ocfg, blob, my, op, 150, action_selection=sel, nn_mode=oracle.ACC_FMA)
            assert_search_equal(got, ref, f"variant {v} sel {sel}")


def test_deep_narrow_trees_exercise_long_paths(oracle, blob):
    """Low FPU + small c make the search dive: descent paths far longer than one 16-lane register chunk (the path lives
    on lane L&15 of register L>>4) must still select / backprop exactly like the oracle."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    eng = sa.Engine(concurrent_games=128, max_explores=800)
    eng.load_weights(blob)
    kw = dict(c=0.35, fpu_value=-1.0)
    my, op = random_positions(oracle, 40, seed=51, max_moves=12)
    my[0] = 0; op[0] = 0
    got = eng.mcts_search(sa.MCTSConfig(**kw), my, op, 800)
    ref = oracle.c4_mcts_search(parity_mcts_config(**kw), blob, my, op, 800, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, "deep trees")
    sp = eng.selfplay(sa.parity_rollout_config(300, mcts_cfg=sa.MCTSConfig(**kw)), base_seed=13, n_games=48, counters=True)
    rf = oracle.c4_selfplay(parity_rollout_config(300, mcts=parity_mcts_config(**kw)), blob, 13, 48, threads=8,
                            nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(sp, rf, "deep trees self-play")
    assert sp["counters"]["max_depth"] == rf["counters"]["max_depth"]
    assert sp["counters"]["max_depth"] > 34, sp["counters"]["max_depth"]  # more than two register chunks deep
    eng.close()


def test_engine_error_behaviour(blob):
    """Errors are codes + messages, never a silent fallback (SURVEY §8b 'Errors')."""
    import synthesis_amd as sa

    eng = sa.Engine(concurrent_games=16, max_explores=32)
    with pytest.raises(sa.SynthesisAmdError) as e:
        eng.policy_eval([0], [0])
    assert e.value.code == -4  # SYN_ERR_NO_WEIGHTS
    with pytest.raises(sa.SynthesisAmdError):
        eng.load_weights(blob[:-1])
    eng.load_weights(blob)
    with pytest.raises(sa.SynthesisAmdError) as e:
        eng.mcts_search(sa.parity_mcts_config(), [0], [0], 33)
    assert e.value.code == -6  # SYN_ERR_CAPACITY
    # Fpu::Func = Normal(mean, std) and PolicyNoise::Dirichlet run on the device (round 2); malformed parameters are rejected
    with pytest.raises(sa.SynthesisAmdError) as e:
        eng.mcts_search(sa.MCTSConfig(fpu=sa.Fpu.Func, fpu_std=-1.0), [0], [0], 8)
    assert e.value.code == -1  # SYN_ERR_INVALID_ARGUMENT (Normal::new rejects std < 0)
    with pytest.raises(sa.SynthesisAmdError) as e:
        eng.mcts_search(sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=0.0, noise_weight=0.25), [0], [0], 8)
    assert e.value.code == -1  # Dirichlet::new_with_size rejects alpha <= 0
    eng.mcts_search(sa.MCTSConfig(fpu=sa.Fpu.Func, fpu_value=1.0, fpu_std=0.1), [0], [0], 8)
    eng.mcts_search(sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=0.3, noise_weight=0.25), [0], [0], 8)
    r = eng.selfplay(sa.parity_rollout_config(8), 0, 0)
    assert r["plies"].size == 0
    # roots the reference itself could not search: overlapping stones, a floating stone, a full board
    for my, op in ((1, 1), (2, 0), ((1 << 63) - 1 - 0x2AAAAAAAAAAAAAAA & ((1 << 63) - 1), 0x2AAAAAAAAAAAAAAA)):
        with pytest.raises(sa.SynthesisAmdError) as e:
            eng.mcts_search(sa.parity_mcts_config(), [my], [op], 8)
        assert e.value.code == -1
    eng.close()


# ------------------------------------------------------------------------------------------------ self-play
SELFPLAY_KEYS = ("plies", "states_bb", "pis", "vs", "actions", "root_nodes", "final_kind")


def assert_selfplay_equal(got, ref, ctx=""):
    assert np.array_equal(got["plies"], ref["plies"]), (ctx, got["plies"], ref["plies"])
    for g in range(len(ref["plies"])):
        n = ref["plies"][g]
        for k in ("states_bb", "pis", "vs", "actions", "root_nodes"):
            if not np.array_equal(got[k][g, :n], ref[k][g, :n]):
                ply = int(np.argwhere((got[k][g, :n] != ref[k][g, :n]).reshape(n, -1).any(axis=1))[0][0])
                raise AssertionError(f"{ctx}: game {g} {k} differs first at ply {ply}: {got[k][g, ply]} vs {ref[k][g, ply]}")
    assert np.array_equal(got["final_kind"], ref["final_kind"])


def test_selfplay_matches_oracle_800_explores(engine, oracle, blob):
    """BASELINE config (800 explores, parity MCTS config): whole games — every position, visit distribution, value
    target, sampled action and tree size — identical to the oracle's run_game with the same per-game StdRng."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_rollout_config

    got = engine.selfplay(sa.parity_rollout_config(800), base_seed=11, n_games=12, counters=True)
    ref = oracle.c4_selfplay(parity_rollout_config(800), blob, 11, 12, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "800 explores")
    for k in ("explores", "select_levels", "children_scanned", "expansions", "new_nodes", "policy_evals",
              "backprop_levels", "solver_children", "solved_hits"):
        assert got["counters"][k] == ref["counters"][k], k
    assert got["counters"]["games"] == 12 and got["counters"]["moves"] == int(ref["plies"].sum())


def test_selfplay_many_games_and_refill(engine, oracle, blob):
    """More games than tree slots (256 slots, 700 games): finished games hand their slot to the next game index;
    results depend only on the game index. Also first_game offsets (the multi-GPU shard parameter)."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_rollout_config

    cfg = sa.parity_rollout_config(40)
    got = engine.selfplay(cfg, base_seed=3, n_games=700)
    ref = oracle.c4_selfplay(parity_rollout_config(40), blob, 3, 700, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, "700 games")
    part = engine.selfplay(cfg, base_seed=3, n_games=100, first_game=350)
    sub = {k: got[k][350:450] for k in SELFPLAY_KEYS}  # entries past a game's last ply are unspecified
    assert_selfplay_equal(part, sub, "first_game offset")


def test_selfplay_value_targets_and_action_variants(engine, oracle, blob):
    """ValueTarget::{Z, QZaverage, QtoZ}, stop_games_when_solved, ActionSelection::Q, other sampling horizons
    (alpha_zero.rs:270-338)."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_rollout_config

    variants = [
        (dict(value_target=sa.ValueTarget.Z), dict(value_target=0)),
        (dict(value_target=sa.ValueTarget.QZaverage, value_target_p=0.3), dict(value_target=2, vt_p=0.3)),
        (dict(value_target=sa.ValueTarget.QtoZ, value_target_from=0.1, value_target_to=0.9), dict(value_target=3, vt_from=0.1, vt_to=0.9)),
        (dict(stop_games_when_solved=True), dict(stop_games_when_solved=1)),
        (dict(action=sa.ActionSelection.Q, random_actions_until=3, sample_actions_until=10),
         dict(action=0, random_actions_until=3, sample_actions_until=10)),
        (dict(random_actions_until=0, sample_actions_until=0), dict(random_actions_until=0, sample_actions_until=0)),
    ]
    for sv, ov in variants:
        got = engine.selfplay(sa.parity_rollout_config(60, **sv), base_seed=21, n_games=40)
        ref = oracle.c4_selfplay(parity_rollout_config(60, **ov), blob, 21, 40, threads=8, nn_mode=oracle.ACC_FMA)
        assert_selfplay_equal(got, ref, str(sv))


@pytest.mark.parametrize("quads", [2, 3, 4])
def test_quad_async_kernel_matches_oracle(blob, oracle, monkeypatch, quads):
    """The many-trees-per-CU kernel (NQ quads of 16 trees per workgroup sharing one LDS weight image, quad-level spin
    barriers, pooled exchange buffers) is normally chosen above 32 trees per CU; force it on a small engine and hold it
    to the same bit-exact bar: searches incl. late-game solver positions, and whole self-play games with refill."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    monkeypatch.setenv("SYN_DEBUG", "1")  # developer knobs are honoured only with SYN_DEBUG=1
    monkeypatch.setenv("SYN_QUADS", str(quads))
    eng = sa.Engine(concurrent_games=208, max_explores=800)  # 13 quads: partial last workgroup for every NQ
    eng.load_weights(blob)
    my, op = random_positions(oracle, 300, seed=31, max_moves=60)
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 120)
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 120, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, f"quads={quads} search")
    got = eng.selfplay(sa.parity_rollout_config(50), base_seed=77, n_games=500, counters=True)
    ref = oracle.c4_selfplay(parity_rollout_config(50), blob, 77, 500, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, f"quads={quads} self-play")
    assert got["counters"]["policy_evals"] == ref["counters"]["policy_evals"]
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=5, n_games=6)
    ref = oracle.c4_selfplay(parity_rollout_config(800), blob, 5, 6, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, f"quads={quads} 800 explores")
    eng.close()


@pytest.mark.parametrize("waves", [4, 8, 12, 16])
def test_lane_per_tree_kernel_matches_oracle(blob, oracle, monkeypatch, waves):
    """The high-concurrency launch shape (lane_kernel.cuh: one tree per lane, own record layout, path-buffer backprop,
    free-running waves) is normally chosen from 65,536 concurrent games; force it on a small engine (partial last
    wave, partial last workgroup) and hold it to the same bit-exact bar as the row-per-tree kernels: searches incl.
    late-game solver positions, every config family, whole self-play games with refill and all value targets."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    monkeypatch.setenv("SYN_DEBUG", "1")  # developer knobs are honoured only with SYN_DEBUG=1
    monkeypatch.setenv("SYN_LANES", str(waves))
    eng = sa.Engine(concurrent_games=1100, max_explores=800)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 300, seed=31, max_moves=60)
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 120)
    assert eng.last_launch_shape()[0] == 4 and eng.last_launch_shape()[2] == 64 * waves
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 120, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, f"lanes={waves} search")
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
        assert_search_equal(got, ref, f"lanes={waves} variant {kw}")
    got = eng.selfplay(sa.parity_rollout_config(50), base_seed=77, n_games=2500, counters=True)
    assert eng.last_launch_shape()[0] == 4
    ref = oracle.c4_selfplay(parity_rollout_config(50), blob, 77, 2500, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, f"lanes={waves} self-play")
    for k in ("explores", "select_levels", "children_scanned", "expansions", "new_nodes", "policy_evals",
              "backprop_levels", "solver_children", "solved_hits", "max_depth"):
        assert got["counters"][k] == ref["counters"][k], (waves, k)
    assert got["counters"]["games"] == 2500 and got["counters"]["moves"] == int(got["plies"].sum())
    for sv, ov in ((dict(value_target=sa.ValueTarget.Z), dict(value_target=0)),
                   (dict(value_target=sa.ValueTarget.QZaverage, value_target_p=0.3), dict(value_target=2, vt_p=0.3)),
                   (dict(value_target=sa.ValueTarget.QtoZ, value_target_from=0.1, value_target_to=0.9),
                    dict(value_target=3, vt_from=0.1, vt_to=0.9)),
                   (dict(stop_games_when_solved=True, action=sa.ActionSelection.Q, random_actions_until=3),
                    dict(stop_games_when_solved=1, action=0, random_actions_until=3))):
        got = eng.selfplay(sa.parity_rollout_config(40, **sv), base_seed=9, n_games=64)
        ref = oracle.c4_selfplay(parity_rollout_config(40, **ov), blob, 9, 64, threads=8, nn_mode=oracle.ACC_FMA)
        assert_selfplay_equal(got, ref, f"lanes={waves} {sv}")
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=5, n_games=6)
    ref = oracle.c4_selfplay(parity_rollout_config(800), blob, 5, 6, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, f"lanes={waves} 800 explores")
    eng.close()


@pytest.mark.parametrize("nv", [1, 2, 3])
def test_producer_consumer_kernel_matches_oracle(blob, oracle, monkeypatch, nv, debug_shapes):
    """The producer/consumer debug shape (pc_kernel.cuh: 12 tree waves time-slicing `nv` virtual waves of 64 trees each, 4 matrix
    waves fed through an LDS ring, per-tree state parked in global memory between visits) is never chosen automatically
    (measured no faster than the symmetric kernel); force it on a small engine (partial virtual waves, idle tree waves) and hold it to the same
    bit-exact bar as every other shape: searches incl. late-game solver positions, every config family, whole self-play
    games with refill and all value targets."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    monkeypatch.setenv("SYN_DEBUG", "1")  # developer knobs are honoured only with SYN_DEBUG=1
    monkeypatch.setenv("SYN_PC", str(nv))
    eng = sa.Engine(concurrent_games=1100, max_explores=800)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 300, seed=31, max_moves=60)
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 120)
    assert eng.last_launch_shape()[0] == 5 and eng.last_launch_shape()[2] == 1024
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 120, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, f"pc nv={nv} search")
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
        assert_search_equal(got, ref, f"pc nv={nv} variant {kw}")
    got = eng.selfplay(sa.parity_rollout_config(50), base_seed=77, n_games=2500, counters=True)
    assert eng.last_launch_shape()[0] == 5
    ref = oracle.c4_selfplay(parity_rollout_config(50), blob, 77, 2500, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, f"pc nv={nv} self-play")
    for k in ("explores", "select_levels", "children_scanned", "expansions", "new_nodes", "policy_evals",
              "backprop_levels", "solver_children", "solved_hits", "max_depth"):
        assert got["counters"][k] == ref["counters"][k], (nv, k)
    assert got["counters"]["games"] == 2500 and got["counters"]["moves"] == int(got["plies"].sum())
    for sv, ov in ((dict(value_target=sa.ValueTarget.Z), dict(value_target=0)),
                   (dict(value_target=sa.ValueTarget.QZaverage, value_target_p=0.3), dict(value_target=2, vt_p=0.3)),
                   (dict(value_target=sa.ValueTarget.QtoZ, value_target_from=0.1, value_target_to=0.9),
                    dict(value_target=3, vt_from=0.1, vt_to=0.9)),
                   (dict(stop_games_when_solved=True, action=sa.ActionSelection.Q, random_actions_until=3),
                    dict(stop_games_when_solved=1, action=0, random_actions_until=3))):
        got = eng.selfplay(sa.parity_rollout_config(40, **sv), base_seed=9, n_games=64)
        ref = oracle.c4_selfplay(parity_rollout_config(40, **ov), blob, 9, 64, threads=8, nn_mode=oracle.ACC_FMA)
        assert_selfplay_equal(got, ref, f"pc nv={nv} {sv}")
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=5, n_games=6)
    ref = oracle.c4_selfplay(parity_rollout_config(800), blob, 5, 6, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, f"pc nv={nv} 800 explores")
    eng.close()


def test_producer_consumer_kernel_with_policy_cache(blob, oracle, monkeypatch, debug_shapes):
    """PolicyWithCache on the producer/consumer kernel (hits skip the ring; their outputs wait in the virtual wave's own
    buffer): results are the oracle's bit for bit, with a tiny contended table and with one that hits."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    monkeypatch.setenv("SYN_DEBUG", "1")  # developer knobs are honoured only with SYN_DEBUG=1
    monkeypatch.setenv("SYN_PC", "2")
    my, op = random_positions(oracle, 300, seed=31, max_moves=60)
    sref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 120, nn_mode=oracle.ACC_FMA)
    gref = oracle.c4_selfplay(parity_rollout_config(50), blob, 77, 2500, threads=8, nn_mode=oracle.ACC_FMA)
    for log2 in (10, 22):
        eng = sa.Engine(concurrent_games=1100, max_explores=800, policy_cache_log2=log2)
        eng.load_weights(blob)
        for rep in range(2):
            got = eng.mcts_search(sa.parity_mcts_config(), my, op, 120)
            assert_search_equal(got, sref, f"pc cache 2^{log2} search pass {rep}")
        assert eng.last_launch_shape()[0] == 5
        hits, misses = eng.last_cache_stats()
        assert hits + misses > 0 and (log2 == 10 or hits > misses)
        got = eng.selfplay(sa.parity_rollout_config(50), base_seed=77, n_games=2500, counters=True)
        assert_selfplay_equal(got, gref, f"pc cache 2^{log2} self-play")
        hits, misses = eng.last_cache_stats()
        assert hits + misses == got["counters"]["policy_evals"] == gref["counters"]["policy_evals"]
        if log2 == 22:
            assert hits > 0.3 * (hits + misses)
        eng.close()


@pytest.mark.parametrize("conc", [300, 1100])
def test_fpu_normal_and_dirichlet_noise_match_oracle(blob, oracle, conc):
    """The reference's own self-play MCTS configuration (study-connect4/src/main.rs:37-49: Fpu::Func(|| Normal(1.0, 0.1)))
    and PolicyNoise::Dirichlet (mcts.rs:241-256) on the device: rand_distr's samplers (ziggurat normal, Marsaglia-Tsang gamma)
    on a per-tree StdRng stream, bit for bit the oracle's — searches, whole self-play games, and the reference's property that
    the root's priors still sum to 1 after the noise (mcts.rs:834-868). Neither configuration returns SYN_ERR_UNSUPPORTED."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    eng = sa.Engine(concurrent_games=conc, max_explores=400)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 120, seed=77, max_moves=50)
    my[0] = 0; op[0] = 0
    variants = [
        (sa.reference_selfplay_mcts_config(), dict(fpu=2, fpu_value=1.0, fpu_std=0.1)),
        (sa.MCTSConfig(fpu=sa.Fpu.Func, fpu_value=0.5, fpu_std=0.3, exploration=sa.Exploration.Uct, c=1.4),
         dict(fpu=2, fpu_value=0.5, fpu_std=0.3, exploration=0, c=1.4)),
        (sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=0.3, noise_weight=0.25),
         dict(noise=2, noise_alpha=0.3, noise_weight=0.25)),
        (sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=1.0, noise_weight=0.5, fpu=sa.Fpu.Func, fpu_value=1.0,
                       fpu_std=0.1), dict(noise=2, noise_alpha=1.0, noise_weight=0.5, fpu=2, fpu_value=1.0, fpu_std=0.1)),
        (sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=2.5, noise_weight=0.25, auto_extend=False),
         dict(noise=2, noise_alpha=2.5, noise_weight=0.25, auto_extend=0)),
    ]
    for scfg, okw in variants:
        for explores in (0, 150):
            got = eng.mcts_search(scfg, my, op, explores)
            assert eng.last_launch_shape()[0] == 4   # the draws live in the lane-per-tree kernels
            ref = oracle.c4_mcts_search(parity_mcts_config(**okw), blob, my, op, explores, nn_mode=oracle.ACC_FMA)
            assert_search_equal(got, ref, f"noise variant {okw} explores {explores}")
            if explores == 0:
                np.testing.assert_allclose(got["child_P"].sum(axis=1), 1.0, atol=1e-6)
    for scfg, okw in variants[:1] + variants[3:4]:
        got = eng.selfplay(sa.parity_rollout_config(60, mcts_cfg=scfg), base_seed=31, n_games=300, first_game=7, counters=True)
        ref = oracle.c4_selfplay(parity_rollout_config(60, mcts=parity_mcts_config(**okw)), blob, 31, 300, first_game=7, threads=8,
                                 nn_mode=oracle.ACC_FMA)
        assert_selfplay_equal(got, ref, f"self-play with {okw}")
        assert got["counters"]["policy_evals"] == ref["counters"]["policy_evals"]
    eng.close()


@pytest.mark.parametrize("log2", [10, 22])
def test_policy_cache_is_semantics_neutral(blob, oracle, monkeypatch, log2):
    """PolicyWithCache on the device (policies/cache.rs:19-32; lane-per-tree kernel): searches and whole self-play games
    are bit-identical to the oracle whether a position's evaluation came from the table or from the network — also with
    a 1,024-entry table where almost every slot is fought over by concurrent games (torn / overwritten entries must
    read as misses) — and the big table does hit."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    monkeypatch.setenv("SYN_DEBUG", "1")  # developer knobs are honoured only with SYN_DEBUG=1
    monkeypatch.setenv("SYN_LANES", "8")
    eng = sa.Engine(concurrent_games=1100, max_explores=800, policy_cache_log2=log2)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 300, seed=31, max_moves=60)
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob, my, op, 120, nn_mode=oracle.ACC_FMA)
    for rep in range(2):  # the second pass finds the first pass's entries
        got = eng.mcts_search(sa.parity_mcts_config(), my, op, 120)
        assert_search_equal(got, ref, f"cache 2^{log2} search pass {rep}")
    hits, misses = eng.last_cache_stats()
    assert hits + misses > 0 and (log2 == 10 or hits > misses)
    ref = oracle.c4_selfplay(parity_rollout_config(50), blob, 77, 2500, threads=8, nn_mode=oracle.ACC_FMA)
    got = eng.selfplay(sa.parity_rollout_config(50), base_seed=77, n_games=2500, counters=True)
    assert eng.last_launch_shape()[0] == 4
    assert_selfplay_equal(got, ref, f"cache 2^{log2} self-play")
    hits, misses = eng.last_cache_stats()
    assert hits + misses == got["counters"]["policy_evals"] == ref["counters"]["policy_evals"]
    if log2 == 22:
        assert hits > 0.3 * (hits + misses)  # openings and re-searched subtrees repeat
    got = eng.selfplay(sa.parity_rollout_config(800), base_seed=5, n_games=6)
    ref = oracle.c4_selfplay(parity_rollout_config(800), blob, 5, 6, threads=8, nn_mode=oracle.ACC_FMA)
    assert_selfplay_equal(got, ref, f"cache 2^{log2} 800 explores")
    # a new network must not see the old network's cached evaluations
    blob2 = (blob * np.float32(0.5)).astype(np.float32)
    eng.load_weights(blob2)
    got = eng.mcts_search(sa.parity_mcts_config(), my[:100], op[:100], 120)
    ref = oracle.c4_mcts_search(parity_mcts_config(), blob2, my[:100], op[:100], 120, nn_mode=oracle.ACC_FMA)
    assert_search_equal(got, ref, f"cache 2^{log2} after new weights")
    eng.close()
    with pytest.raises(sa.SynthesisAmdError):
        sa.Engine(concurrent_games=64, max_explores=64, policy_cache_log2=5)


def test_selfplay_full_size_properties(blob, oracle, monkeypatch):
    """BASELINE full size: 4096 concurrent games x 800 explores. Too big for the oracle, so size-independent
    properties: replaying the recorded actions with the oracle's Connect4 reproduces every recorded position and ends
    exactly at the recorded ply; visit distributions sum to 1 and are zero on illegal columns; value targets are
    distributions; tree sizes respect 1 + 9*(explores+1); counters are self-consistent; a re-run is bit-identical."""
    import synthesis_amd as sa

    eng = sa.Engine(concurrent_games=4096, max_explores=800)
    eng.load_weights(blob)
    cfg = sa.parity_rollout_config(800)
    r = eng.selfplay(cfg, base_seed=2024, n_games=4096, counters=True)
    plies = r["plies"]
    assert plies.min() >= 7 and plies.max() <= 63
    mask = np.arange(63)[None, :] < plies[:, None]
    np.testing.assert_allclose(r["pis"].sum(axis=2)[mask], 1.0, atol=2e-6)
    np.testing.assert_allclose(r["vs"].sum(axis=2)[mask], 1.0, atol=2e-6)
    assert r["root_nodes"][mask].max() <= 1 + 9 * 801
    occ = r["states_bb"][..., 0] | r["states_bb"][..., 1]
    for c in range(9):  # full column => zero visit probability
        full = ((occ >> np.uint64(7 * c)) & np.uint64(0x7F)) == np.uint64(0x7F)
        assert np.all(r["pis"][..., c][mask & full] == 0)
    for g in range(0, 4096, 37):  # replay a sample of games on the oracle's rules
        n = plies[g]
        rep = oracle.c4_play(list(r["actions"][g, :n]))
        assert rep["over"][-1] and not rep["over"][:-1].any()
        assert (rep["my_bb"], rep["op_bb"]) != (0, 0)
        pre = oracle.c4_play(list(r["actions"][g, : n - 1]))
        assert (pre["my_bb"], pre["op_bb"]) == (int(r["states_bb"][g, n - 1, 0]), int(r["states_bb"][g, n - 1, 1]))
    c = r["counters"]
    assert c["games"] == 4096 and c["moves"] == int(plies.sum())
    assert c["policy_evals"] <= c["expansions"] <= c["explores"]
    assert c["new_nodes"] <= 9 * c["expansions"]
    r2 = eng.selfplay(cfg, base_seed=2024, n_games=4096)
    assert_selfplay_equal(r2, r, "re-run")
    eng.close()
    # the same games on the 16,384-slot engine (quad-async kernel): results depend only on the game index
    big = sa.Engine(concurrent_games=16384, max_explores=800)
    big.load_weights(blob)
    r3 = big.selfplay(cfg, base_seed=2024, n_games=4096)
    assert_selfplay_equal(r3, r, "16384-slot engine")
    big.close()
    # ... and on the lane-per-tree kernel (different record layout, different backprop): bit-identical again
    monkeypatch.setenv("SYN_DEBUG", "1")  # developer knobs are honoured only with SYN_DEBUG=1
    monkeypatch.setenv("SYN_LANES", "12")
    lanes = sa.Engine(concurrent_games=4096, max_explores=800)
    lanes.load_weights(blob)
    r4 = lanes.selfplay(cfg, base_seed=2024, n_games=4096, counters=True)
    assert lanes.last_launch_shape()[0] == 4
    assert_selfplay_equal(r4, r, "lane-per-tree kernel")
    assert r4["counters"] == r["counters"]
    lanes.close()


def test_vanilla_mcts_with_rollout_policy_matches_oracle(oracle):
    """MCTS over RolloutPolicy (rollout.rs:8-31; the pairing of the reference's MCTS tests, mcts.rs:691-868): leaf
    evaluations — random playouts on the tree's own StdRng stream, uniform priors — as the reference configures it
    (Uct, no auto-extend, fpu = inf, study-connect4/src/main.rs:74-82) and with the AlphaZero-style config; no network
    weights are loaded. Visit counts, sums, solutions and targets bit-identical to the oracle."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config

    eng = sa.Engine(concurrent_games=600, max_explores=800)
    my, op = random_positions(oracle, 500, seed=71, max_moves=50)
    my[0] = 0; op[0] = 0
    for kw, explores in ((dict(exploration=0, c=2.0, auto_extend=0, fpu_value=float("inf")), 400),
                         (dict(exploration=0, c=0.7, fpu_value=0.5), 150), (dict(), 300)):
        ocfg = parity_mcts_config(**kw)
        scfg = sa.MCTSConfig(exploration=sa.Exploration(ocfg.exploration), c=ocfg.c, solve=bool(ocfg.solve),
                             correct_values_on_solve=bool(ocfg.correct_values_on_solve),
                             select_solved_nodes=bool(ocfg.select_solved_nodes), auto_extend=bool(ocfg.auto_extend),
                             fpu=sa.Fpu(ocfg.fpu), fpu_value=ocfg.fpu_value)
        for sel in (0, 1):
            got = eng.mcts_search(scfg, my, op, explores, action_selection=sel, rollout_seed=1234)
            ref = oracle.c4_mcts_search_rollout(ocfg, 1234, my, op, explores, action_selection=sel)
            assert_search_equal(got, ref, f"rollout {kw} sel {sel}")
    assert eng.last_launch_shape()[0] == 4
    with pytest.raises(sa.SynthesisAmdError):
        eng.mcts_search(sa.parity_mcts_config(), my[:4], op[:4], 50)  # the network path still needs weights
    eng.close()


def test_match_play_reproduces_the_oracle_move_by_move(oracle, blob):
    """synthesis_amd.match (the evaluator's per-move `MCTS::exploit`, evaluator.rs:129-161, batched over games): a match
    network-MCTS vs rollout-MCTS replayed with the oracle — every move of every game is the oracle's best_action for that
    position, and the rewards follow the oracle's rules."""
    import synthesis_amd as sa
    from synthesis_amd import match
    from tests.oracle_lib import parity_mcts_config

    eng = sa.Engine(concurrent_games=256, max_explores=200)
    eng.load_weights(blob)
    me = match.Player("net", 120)
    opp = match.rollout_player(150)
    n, seed = 48, 17
    # replay the lockstep on the host with the oracle choosing every move
    my = np.zeros(n, np.uint64); op = np.zeros(n, np.uint64); alive = np.ones(n, bool); ref_reward = np.zeros(n, np.float32)
    ocfg_net = parity_mcts_config()
    ocfg_roll = parity_mcts_config(exploration=0, c=2.0, auto_extend=0, fpu_value=float("inf"))
    for ply in range(63):
        idx = np.nonzero(alive)[0]
        if idx.size == 0:
            break
        if ply % 2 == 0:
            a = oracle.c4_mcts_search(ocfg_net, blob, my[idx], op[idx], 120, nn_mode=oracle.ACC_FMA)["best_action"]
        else:
            a = oracle.c4_mcts_search_rollout(ocfg_roll, seed + ply * n, my[idx], op[idx], 150)["best_action"]
        nmy, nop, over, w = match.step(my[idx], op[idx], a)
        my[idx], op[idx] = nmy, nop
        ref_reward[idx[over & w]] = 1.0 if ply % 2 == 0 else -1.0
        alive[idx[over]] = False
    got, plies = match.play_match(eng, me, opp, n, seed=seed)
    assert np.array_equal(got, ref_reward)
    assert plies.min() >= 7 and plies.max() <= 63
    w, d, l, s, elo = match.score(got)
    assert w + d + l == n and 0.0 <= s <= 1.0
    eng.close()


def test_every_launch_shape_plays_the_same_games():
    """Results depend only on (config, seed, game index): five configuration families (Uct without auto-extend, ParentQ +
    Equal noise, solver off + stop_games_when_solved + ActionSelection::Q, deep trees + QtoZ targets, parity), 3,000 games
    each, on the row-per-tree kernel, the lane-per-tree kernel and the lane kernel with the policy cache — every output
    array and every counter identical (tools/stress_cross_kernel.py)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_cross_kernel.py")], capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0 and "ALL IDENTICAL" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
