"""The C++ host mirror of the reference's plug-in surface (include/synthesis_amd.hpp) driven by a C++ caller standing in
for the Rust host (tests/cpp/host_harness.cpp): Policy::eval through the adaptor, run_n_games into a
ReplayBuffer, deduplicate, one learner step — every printed value compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory, golden_dir):
    d = tmp_path_factory.mktemp("cpp")
    exe = str(d / "host_harness")
    lib = os.path.join(ROOT, "synthesis_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_harness.cpp"), "-o", exe, "-L" + lib,
                           "-lsynthesis_amd", "-Wl,-rpath," + lib, "-pthread"])
    blob = str(d / "blob.f32")
    np.load(os.path.join(golden_dir, "c4net_blob_f32.npy")).astype("<f4").tofile(blob)
    return exe, blob


def f32s(tokens):
    return np.array([struct.unpack("<f", struct.pack("<I", int(t, 16)))[0] for t in tokens], np.float32)


def test_cpp_host_fails_loudly_without_a_gpu(harness):
    """No CPU fallback anywhere above the boundary either: the C++ Engine constructor throws the ABI's status."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    exe, blob = harness
    p = subprocess.run([exe, blob], capture_output=True, text=True)
    assert p.returncode == 3 and p.stdout.startswith("error -2 ")


@pytest.mark.gpu
def test_cpp_host_matches_oracle(harness, oracle, golden_dir):
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    exe, blobf = harness
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    p = subprocess.run([exe, blobf], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l.split() for l in p.stdout.splitlines()]
    by = {}
    for l in lines:
        by.setdefault(l[0], []).append(l[1:])

    assert by["two_policies_batching"] == [["0", "0"]]   # two HipPolicy objects of one Engine, eval_batch from two threads at once
    # Policy::eval (batch of one and batched) + Game::features + Game::player on the scripted line 4 4 3 5 2 4 0 8 8
    moves = [4, 4, 3, 5, 2, 4, 0, 8, 8]
    assert len(by["pos"]) == len(moves) + 1
    for i, pos in enumerate(by["pos"]):
        r = oracle.c4_play(moves[:i]) if i else dict(my_bb=0, op_bb=0, player=0)
        assert (int(pos[1]), int(pos[2]), int(pos[3])) == (r["my_bb"], r["op_bb"], r["player"])
        my = np.array([r["my_bb"]], np.uint64); op = np.array([r["op_bb"]], np.uint64)
        fl, fv = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_FMA)
        assert np.array_equal(f32s(by["logits"][i]), fl[0]) and np.array_equal(f32s(by["value"][i]), fv[0])
        assert np.array_equal(f32s(by["batch_logits"][i]), fl[0])
        assert np.array_equal(f32s(by["features"][i]), oracle.c4_features(my, op)[0])

    # run_n_games -> ReplayBuffer: positions, visit distributions and value targets of 12 games, in order
    ref = oracle.c4_selfplay(parity_rollout_config(64), blob, 5, 16, nn_mode=oracle.ACC_FMA)
    steps12 = int(ref["plies"][:12].sum())
    assert [int(x) for x in by["buffer"][0][:3]] == [12, 12, steps12]
    assert int(by["buffer"][0][3]) == int(oracle.c4_selfplay(parity_rollout_config(64), blob, 5, 12, nn_mode=oracle.ACC_FMA,
                                                            outputs=False)["counters"]["policy_evals"])
    k = 0
    for g in range(12):
        for t in range(ref["plies"][g]):
            assert (int(by["step"][k][0]), int(by["step"][k][1])) == (int(ref["states_bb"][g, t, 0]), int(ref["states_bb"][g, t, 1]))
            assert np.array_equal(f32s(by["pi"][k]), ref["pis"][g, t]) and np.array_equal(f32s(by["v"][k]), ref["vs"][g, t])
            k += 1
    # extend with games 12..15, keep_last_n_games(10), deduplicate. Game ids start at 1 (new_game() precedes the adds,
    # data.rs:131-133) and the cut is `game_id < total - n` (data.rs:171-180), so ids 6..16 survive: 11 games.
    kept_steps = int(ref["plies"][5:16].sum())
    assert [int(x) for x in by["kept"][0]] == [16, 11, kept_steps]
    my = np.concatenate([ref["states_bb"][g, : ref["plies"][g], 0] for g in range(5, 16)])
    op = np.concatenate([ref["states_bb"][g, : ref["plies"][g], 1] for g in range(5, 16)])
    pi = np.concatenate([ref["pis"][g, : ref["plies"][g]] for g in range(5, 16)])
    v = np.concatenate([ref["vs"][g, : ref["plies"][g]] for g in range(5, 16)])
    dd = oracle.dedup(my, op, pi, v)
    assert int(by["dedup"][0][0]) == dd["num"].size

    # search of the empty board
    s = oracle.c4_mcts_search(parity_mcts_config(), blob, np.zeros(1, np.uint64), np.zeros(1, np.uint64), 64, nn_mode=oracle.ACC_FMA)
    assert np.array_equal(f32s(by["child_N"][0]), s["child_N"][0])
    assert (int(by["best"][0][0]), int(by["best"][0][2])) == (int(s["best_action"][0]), int(s["num_nodes"][0]))

    # evaluator.rs:200-228 / 163-198 through the C++ mirror: rewards of whole matches
    rcfg = parity_mcts_config(exploration=0, c=2.0, auto_extend=0, fpu_value=float("inf"))
    want = [oracle.c4_mcts_vs_mcts(rcfg, 0, 60, 30, sd)[0] for sd in (1, 2, 3, 4)]
    assert list(f32s(by["vanilla_rewards"][0])) == want
    want = [oracle.c4_eval_against_rollout(parity_mcts_config(), 48, blob, rcfg, 1, 40, sd, nn_mode=oracle.ACC_FMA)[0] for sd in (5, 6, 7)]
    assert list(f32s(by["versus_rewards"][0])) == want

    other = blob.copy(); other[:200] = -other[:200]
    want = [oracle.c4_eval_against_old(parity_mcts_config(), 40, blob, other, nn_mode=oracle.ACC_FMA)[0],
            oracle.c4_eval_against_old(parity_mcts_config(), 40, other, blob, nn_mode=oracle.ACC_FMA)[0]]
    assert list(f32s(by["old_rewards"][0])) == want

    # one learner step on the first 32 unique states, published to the self-play network
    from tests.oracle_lib import default_train_hyper

    X = oracle.c4_features(dd["my_bb"][:32], dd["op_bb"][:32])
    w, _, _, _, lo = oracle.train_steps(blob, default_train_hyper(), X[None], dd["pis"][:32][None], dd["vs"][:32][None], [1e-3])
    assert np.array_equal(f32s(by["losses"][0]), lo[0])
    fl, _ = oracle.c4net_eval(w, np.zeros(1, np.uint64), np.zeros(1, np.uint64), mode=oracle.ACC_FMA)
    assert np.array_equal(f32s(by["logits_after_step"][0]), fl[0])

    # error behaviour: wrong blob size -> SYN_ERR_INVALID_ARGUMENT; an eighth stone in a column -> the same code
    assert [c[0] for c in by["caught"]] == ["-1", "-1"] and "no_error" not in by


@pytest.fixture(scope="module")
def c_example(tmp_path_factory, golden_dir):
    """examples/policy_eval_worker.c built with the command line its header documents (plain C against include/synthesis_amd.h)."""
    d = tmp_path_factory.mktemp("cexample")
    exe = str(d / "policy_eval_worker")
    lib = os.path.join(ROOT, "synthesis_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "policy_eval_worker.c"), "-o", exe, "-L" + lib, "-lsynthesis_amd",
                           "-Wl,-rpath," + lib])
    blob = str(d / "blob.f32")
    np.load(os.path.join(golden_dir, "c4net_blob_f32.npy")).astype("<f4").tofile(blob)
    return exe, blob


def test_c_example_builds_and_fails_loudly_without_a_gpu(c_example):
    import torch

    exe, blob = c_example
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    p = subprocess.run([exe, blob], capture_output=True, text=True)
    assert p.returncode == 1 and p.stderr.strip() != ""


@pytest.mark.gpu
def test_c_example_workers_get_the_oracles_bits(c_example, oracle, golden_dir):
    """Three worker threads, each with its own evaluation context on one engine: Policy::eval one position per call and as a
    submitted batch — every printed float is the oracle's, bit for bit."""
    exe, blobf = c_example
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    p = subprocess.run([exe, blobf], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 15
    for line in lines:
        tok = line.split()
        w, pos = int(tok[1]), int(tok[3].rstrip(":"))
        r = oracle.c4_play([(w + i) % 9 for i in range(pos + 1)])
        fl, fv = oracle.c4net_eval(blob, [r["my_bb"]], [r["op_bb"]], mode=oracle.ACC_FMA)
        got = np.array([int(t, 16) for t in tok[4:16]], np.uint32)
        assert np.array_equal(got[:9], fl[0].view(np.uint32)) and np.array_equal(got[9:], fv[0].view(np.uint32))
        assert tok[-2:] == ["batch", "same"]
