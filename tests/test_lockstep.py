"""The host-tree, batched-policy form of the search (include/synthesis_amd_lockstep.hpp: mcts.rs's MCTS<G, P, N> restated over
any Game with Policy::eval taken out of visit(); BASELINE.json configs[1] as worded). CPU: the header driven by a C++ harness
whose batch policy is the oracle's network — every tree must equal oracle/mcts.hpp's sequential search, field for field — and by a
second Game. GPU: syn_mcts_search_lockstep (trees on the host, syn_policy_eval_batch per round) against the fused
syn_mcts_search and the oracle."""
import os
import subprocess

import numpy as np
import pytest

from tests.test_gpu_parity import SEARCH_KEYS, assert_search_equal, random_positions

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESULT_DTYPE = np.dtype([("child_N", np.float32, (9,)), ("child_W", np.float32, (9, 3)), ("child_P", np.float32, (9,)),
                         ("child_sol", np.int32, (9, 3)), ("root_N", np.float32), ("root_W", np.float32, (3,)),
                         ("root_sol", np.int32, (3,)), ("num_nodes", np.uint32), ("best_action", np.int32),
                         ("target_pi", np.float32, (9,)), ("target_q", np.float32, (3,))])
# the harness's configuration variants (tests/cpp/lockstep_harness.cpp) as oracle configurations
VARIANTS = {
    0: dict(),
    1: dict(exploration=0, c=1.4, fpu=1),
    2: dict(noise=1, noise_weight=0.25, auto_extend=0, select_solved_nodes=0),
    3: dict(solve=0, fpu_value=0.5),
    4: dict(correct_values_on_solve=0, c=1.5),
}


@pytest.fixture(scope="module")
def harness(tmp_path_factory, oracle, golden_dir):
    d = tmp_path_factory.mktemp("lockstep")
    exe = str(d / "lockstep_harness")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-ffp-contract=off", "-pthread",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "lockstep_harness.cpp"),
                           "-o", exe, "-L" + odir, "-loracle", "-Wl,-rpath," + odir])
    blob = str(d / "blob.f32")
    np.load(os.path.join(golden_dir, "c4net_blob_f32.npy")).astype("<f4").tofile(blob)
    return exe, blob, d


def records(path, n):
    rec = np.fromfile(path, RESULT_DTYPE)
    assert rec.shape == (n,)
    out = {k: rec[k].copy() for k in RESULT_DTYPE.names}
    out["root_stat"] = np.concatenate([out.pop("root_N")[:, None], out.pop("root_W")], axis=1)
    return out


def test_lockstep_trees_equal_the_sequential_oracle(harness, oracle, golden_dir):
    from tests.oracle_lib import parity_mcts_config

    exe, blobf, d = harness
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    my, op = random_positions(oracle, 80, seed=2024, max_moves=52)
    my[0] = 0; op[0] = 0
    roots = str(d / "roots.u64")
    np.concatenate([my, op]).astype("<u8").tofile(roots)
    for variant, explores, threads in ((0, 0, 1), (0, 1, 1), (0, 150, 4), (1, 90, 1), (2, 120, 3), (3, 100, 1), (4, 100, 2)):
        out = str(d / f"out{variant}_{explores}.bin")
        p = subprocess.run([exe, "c4", blobf, roots, str(explores), str(variant), str(threads), out], capture_output=True,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        got = records(out, len(my))
        ref = oracle.c4_mcts_search(parity_mcts_config(**VARIANTS[variant]), blob, my, op, explores, nn_mode=oracle.ACC_FMA)
        assert_search_equal(got, ref, f"lockstep variant {variant} explores {explores}")
        rounds, evals, calls = [int(x) for x in p.stdout.split()[1::2]]
        # one batched call per round, and rounds are bounded by the deepest tree: construction + explores
        assert calls == rounds <= explores + 1 and evals <= len(my) * (explores + 1)


def test_lockstep_driver_is_generic_over_the_game(harness):
    """A three-action subtraction game (take 1-3 stones, the last stone wins) under the same driver with a uniform policy: the
    solver proves every root (a multiple of four loses, anything else wins by moving to one) and best_action plays the proof."""
    exe, _, _ = harness
    p = subprocess.run([exe, "nim"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0
    rows = [[int(x) for x in l.split()[1:]] for l in p.stdout.splitlines()]
    for stones, solved, kind, turns, best in rows:
        if stones > 13:
            continue
        assert solved == 1
        if stones % 4 == 0:
            assert kind == 0 and turns == stones // 2          # Lose(turns): the opponent always answers to the next multiple
        else:
            assert kind == 2 and best == stones % 4 - 1 and turns == 2 * (stones // 4) + 1


def test_lockstep_pool_hands_a_worker_threads_exception_to_the_caller(harness):
    """Game::step throwing on a pool thread: the search ends with that exception on the calling thread (no terminate, no hang), and
    the next search starts its own pool."""
    exe, _, _ = harness
    p = subprocess.run([exe, "nimthrow"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.splitlines() == ["caught seven stones", "second search 128 trees, root solved 1"]


@pytest.mark.gpu
def test_lockstep_search_on_the_gpu_equals_the_fused_search(oracle, golden_dir):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config

    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    eng = sa.Engine(concurrent_games=4096, max_explores=200)
    eng.load_weights(blob)
    my, op = random_positions(oracle, 4096, seed=99, max_moves=50)
    my[0] = 0; op[0] = 0
    for scfg, okw, explores in ((sa.parity_mcts_config(), dict(), 64),
                                (sa.MCTSConfig(exploration=sa.Exploration.Uct, c=1.4, fpu=sa.Fpu.ParentQ), dict(exploration=0, c=1.4, fpu=1), 40)):
        got = eng.mcts_search_lockstep(scfg, my, op, explores)
        stats = got.pop("stats")
        fused = eng.mcts_search(scfg, my, op, explores)
        assert_search_equal(got, fused, "lockstep vs fused")
        ref = oracle.c4_mcts_search(parity_mcts_config(**okw), blob, my[:256], op[:256], explores, nn_mode=oracle.ACC_FMA)
        assert_search_equal({k: got[k][:256] for k in SEARCH_KEYS}, ref, "lockstep vs oracle")
        assert 1 <= stats["rounds"] <= explores + 1 and stats["positions_evaluated"] <= 4096 * (explores + 1)
    # the draws of Fpu::Func / Dirichlet live on the device path only
    with pytest.raises(sa.SynthesisAmdError) as e:
        eng.mcts_search_lockstep(sa.reference_selfplay_mcts_config(), my[:4], op[:4], 8)
    assert e.value.code == -5
    eng.close()
