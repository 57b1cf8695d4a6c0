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
    5: dict(fpu=2, fpu_value=1.0, fpu_std=0.1),   # the reference's own self-play configuration (study-connect4/src/main.rs:37-49)
    6: dict(exploration=0, c=1.4, fpu=2, fpu_value=0.5, fpu_std=0.3),
    7: dict(noise=2, noise_alpha=0.3, noise_weight=0.25),     # PolicyNoise::Dirichlet (mcts.rs:241-256): alpha < 1, = 1, > 1
    8: dict(noise=2, noise_alpha=1.0, noise_weight=0.5, fpu=2, fpu_value=1.0, fpu_std=0.1),
    9: dict(noise=2, noise_alpha=2.5, noise_weight=0.25, auto_extend=0),
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
    for variant, explores, threads in ((0, 0, 1), (0, 1, 1), (0, 150, 4), (1, 90, 1), (2, 120, 3), (3, 100, 1), (4, 100, 2),
                                      (5, 0, 1), (5, 150, 3), (6, 90, 2),
                                      # one policy per host thread (lockstep_search_sharded): 3 shards of the 80 roots
                                      (0, 120, -3), (5, 100, -4), (1, 60, -40),
                                      (7, 0, 1), (7, 120, 2), (8, 100, -3), (9, 90, 1)):
        out = str(d / f"out{variant}_{explores}_{threads}.bin")
        # (-40 stands for: four workers sharing ONE policy through a CombiningPolicy)
        p = subprocess.run([exe, "c4", blobf, roots, str(explores), str(variant), str(max(threads, -4)), out], capture_output=True,
                           text=True, timeout=600, env=dict(os.environ, LS_COMBINE="1" if threads == -40 else "0"))
        assert p.returncode == 0, p.stdout + p.stderr
        got = records(out, len(my))
        ref = oracle.c4_mcts_search(parity_mcts_config(**VARIANTS[variant]), blob, my, op, explores, nn_mode=oracle.ACC_FMA)
        assert_search_equal(got, ref, f"lockstep variant {variant} explores {explores}")
        rounds, evals, calls = [int(x) for x in p.stdout.split()[1:6:2]]
        # one batched call per round, and rounds are bounded by the deepest tree: construction + explores — per group of trees that
        # shares its calls: all of them on a pool, two halves taking turns on one thread, two halves per shard
        groups = 1 if threads > 1 else 2 * abs(max(threads, -4))
        assert calls <= rounds <= groups * (explores + 1) and evals <= len(my) * (explores + 1)


def counter_fpu_fn():
    """The counter-based Fpu::Func of the harness's variant 10 (tests/cpp/lockstep_harness.cpp::counter_fpu): call k returns
    0.5 + (k * 2654435761 mod 2^32 >> 16) / 65536. Returns (callable, calls) — calls[0] counts."""
    calls = [0]

    def fn():
        h = (calls[0] * 2654435761) & 0xFFFFFFFF
        calls[0] += 1
        return 0.5 + (h >> 16) / 65536.0

    return fn, calls


def test_fpu_func_pointer_on_the_host_trees_equals_the_oracle(harness, oracle, golden_dir):
    """Fpu::Func(fn() -> f32) as the reference types it (config.rs:25; called once per unexpanded child per scan, mcts.rs:351-356) on
    the host trees: a function with state (a counter) gives a tree the same numbers as the sequential oracle gives it when both call
    it in the same order — one root, one thread, the same explores; the call counts agree as well."""
    from tests.oracle_lib import parity_mcts_config

    exe, blobf, d = harness
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    my, op = random_positions(oracle, 6, seed=77, max_moves=30)
    my[0] = 0; op[0] = 0
    for i in range(len(my)):
        roots = str(d / f"root_fn{i}.u64")
        np.array([my[i], op[i]]).astype("<u8").tofile(roots)
        out = str(d / f"out_fn{i}.bin")
        p = subprocess.run([exe, "c4", blobf, roots, "200", "10", "1", out], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        got = records(out, 1)
        fn, calls = counter_fpu_fn()
        oracle.set_fpu_fn(fn)
        try:
            ref = oracle.c4_mcts_search(parity_mcts_config(fpu=3), blob, my[i:i + 1], op[i:i + 1], 200, nn_mode=oracle.ACC_FMA)
        finally:
            oracle.set_fpu_fn(None)
        assert_search_equal(got, ref, f"Fpu::Func(fn) root {i}")
        assert int(p.stdout.split()[-1]) == calls[0] > 0


SELFPLAY_DTYPE = np.dtype([("bb", np.uint64, (2,)), ("pi", np.float32, (9,)), ("v", np.float32, (3,)), ("action", np.uint32),
                           ("root_nodes", np.uint32)])
GAME_DTYPE = np.dtype([("plies", np.int32), ("final_kind", np.int32), ("pos", SELFPLAY_DTYPE, (63,))])
# the harness's rollout variants as oracle rollout configurations (tests/oracle_lib.py::parity_rollout_config keywords)
SELFPLAY_VARIANTS = {
    0: dict(),
    1: dict(value_target=0, stop_games_when_solved=1, action=0),
    2: dict(value_target=2, vt_p=0.25, random_actions_until=3, sample_actions_until=10, mcts=dict(exploration=0, c=1.4, fpu=1)),
    3: dict(value_target=3, vt_from=0.1, vt_to=0.9, mcts=dict(noise=1, noise_weight=0.25)),
    4: dict(mcts=dict(fpu=2, fpu_value=1.0, fpu_std=0.1)),
    5: dict(mcts=dict(noise=2, noise_alpha=0.3, noise_weight=0.25, fpu=2, fpu_value=1.0, fpu_std=0.1)),
}


def assert_games_equal(got, ref, what):
    """plies, final outcome and every recorded position (state, pi, v, action, nodes.len()) of every game, bit for bit"""
    np.testing.assert_array_equal(got["plies"], ref["plies"], err_msg=what + ": plies")
    np.testing.assert_array_equal(got["final_kind"], ref["final_kind"], err_msg=what + ": final_kind")
    for g in range(len(ref["plies"])):
        n = int(ref["plies"][g])
        for k in ("states_bb", "actions", "root_nodes"):
            np.testing.assert_array_equal(got[k][g, :n], ref[k][g, :n], err_msg=f"{what}: game {g} {k}")
        for k in ("pis", "vs"):
            np.testing.assert_array_equal(got[k][g, :n].view(np.uint32), ref[k][g, :n].view(np.uint32), err_msg=f"{what}: game {g} {k}")


def test_gather_experience_with_the_references_per_worker_rng(harness, oracle, golden_dir):
    """The reference's own RNG discipline (alpha_zero.rs:120-209): num_workers + 1 workers, worker i plays games_to_schedule /
    workers_left games one after another on ONE StdRng::seed_from_u64(seed * (num_workers + 1) + i_worker) that runs through all of them —
    the host-tree driver (gather_experience_host_trees: one thread, one policy, one game in flight per worker) against the oracle's
    sequential restatement, game for game in worker order; and the games DO depend on the number of workers, as in the reference."""
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    exe, blobf, d = harness
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    runs = {}
    for variant, games, explores, workers_plus_1, seed in ((0, 23, 40, 7, 3), (0, 23, 40, 3, 3), (4, 10, 48, 4, 0), (1, 5, 30, 6, 11)):
        out = str(d / f"gather{variant}_{games}_{workers_plus_1}.bin")
        p = subprocess.run([exe, "gather", blobf, str(games), str(explores), str(variant), str(-workers_plus_1), str(seed), "0", out],
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout + p.stderr
        raw = np.fromfile(out, GAME_DTYPE)
        got = dict(plies=raw["plies"], final_kind=raw["final_kind"].astype(np.uint8), states_bb=raw["pos"]["bb"], pis=raw["pos"]["pi"],
                   vs=raw["pos"]["v"], actions=raw["pos"]["action"].astype(np.uint8), root_nodes=raw["pos"]["root_nodes"])
        kw = dict(SELFPLAY_VARIANTS[variant])
        mk = kw.pop("mcts", None)
        cfg = parity_rollout_config(explores, mcts=parity_mcts_config(**mk) if mk else None, **kw)
        ref = oracle.c4_gather_experience(cfg, blob, seed, games, workers_plus_1 - 1, use_cache=True)
        assert_games_equal(got, ref, f"gather_experience variant {variant}, {workers_plus_1} workers")
        runs[(variant, workers_plus_1)] = got
    a, b = runs[(0, 7)], runs[(0, 3)]
    assert not (np.array_equal(a["plies"], b["plies"]) and np.array_equal(a["actions"], b["actions"]))   # 7 workers != 3 workers


def test_lockstep_stdrng_is_the_oracles_stream(harness, oracle):
    """The host StdRng of the lock-step self-play driver (ChaCha12 keyed by seed_from_u64) against the oracle's, word for word
    across several blocks, and against rand's own value-stability constant through it."""
    exe, _, _ = harness
    for seed in (0, 1, 20211003, 2 ** 63 + 12345):
        p = subprocess.run([exe, "rng", str(seed), "70"], capture_output=True, text=True, timeout=60)
        assert p.returncode == 0
        got = np.array([int(x) for x in p.stdout.split()], np.uint32)
        ref = oracle.stdrng_u32(seed, 70)
        np.testing.assert_array_equal(got, ref)


def test_lockstep_selfplay_equals_the_sequential_oracle(harness, oracle, golden_dir):
    """run_n_games over host trees (lockstep_selfplay): every game — states, targets after store_rewards, actions, tree sizes —
    equals oracle/selfplay.hpp's sequential run_game with the same per-game seeding, for every ValueTarget, both action
    selections, stop_games_when_solved, Uct / ParentQ and the equalising root noise."""
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    exe, blobf, d = harness
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    runs = [(0, 12, 60, 3, 7, 0, {}), (1, 8, 50, 1, 99, 5, {}), (2, 8, 40, 2, 3, 0, {}), (3, 8, 40, 4, 11, 2, {}), (4, 10, 60, 3, 31, 7, {}),
            # many games: a pool over all of them; one thread, two halves taking turns; a policy whose begin() really returns early
            (0, 700, 6, 4, 13, 0, {}), (4, 515, 9, 1, 2, 40, {}), (2, 300, 8, 1, 21, 0, dict(LS_ASYNC="1")),
            # one policy per host thread (lockstep_selfplay_sharded), and fewer slots than games: a finished game's slot takes the
            # next game index, whichever worker gets there first — every game still depends on its index only
            (0, 400, 8, -3, 5, 9, {}), (4, 300, 8, -4, 77, 0, dict(LS_CONCURRENT="64", LS_ASYNC="1")),
            (1, 90, 10, 1, 8, 0, dict(LS_CONCURRENT="20")), (3, 200, 6, 3, 8, 3, dict(LS_CONCURRENT="70")),
            (5, 40, 50, 2, 17, 4, {}), (5, 260, 10, -3, 3, 0, dict(LS_COMBINE="1")),
            # the workers share one policy: their batches go to it combined (CombiningPolicy — what syn_selfplay_run_lockstep runs)
            (0, 500, 8, -4, 6, 0, dict(LS_COMBINE="1")), (4, 400, 8, -5, 9, 2, dict(LS_COMBINE="1", LS_ASYNC="1", LS_CONCURRENT="150"))]
    for variant, games, explores, threads, seed, first, env in runs:
        out = str(d / f"sp{variant}_{games}_{threads}.bin")
        p = subprocess.run([exe, "selfplay", blobf, str(games), str(explores), str(variant), str(threads), str(seed), str(first), out],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stdout + p.stderr
        rec = np.fromfile(out, GAME_DTYPE)
        assert rec.shape == (games,)
        got = dict(plies=rec["plies"], final_kind=rec["final_kind"].astype(np.uint8), states_bb=rec["pos"]["bb"], pis=rec["pos"]["pi"],
                   vs=rec["pos"]["v"], actions=rec["pos"]["action"].astype(np.uint8), root_nodes=rec["pos"]["root_nodes"])
        kw = dict(SELFPLAY_VARIANTS[variant])
        mcts = kw.pop("mcts", {})
        ref = oracle.c4_selfplay(parity_rollout_config(explores, mcts=parity_mcts_config(**mcts), **kw), blob, seed, games, first_game=first,
                                 nn_mode=oracle.ACC_FMA)
        assert_games_equal(got, ref, f"lockstep self-play variant {variant}")
        rounds, evals, calls = [int(x) for x in p.stdout.split()[1::2]]
        assert (calls <= rounds if env.get("LS_COMBINE") else calls == rounds) and evals <= int(ref["plies"].sum()) * (explores + 1)


def test_policy_cache_wrappers_change_nothing_but_the_policys_work(harness, oracle):
    """policies/cache.rs:5-59 on the host: BatchPolicyWithCache in front of the batch policy of every worker leaves every tree as it
    was (same records, byte for byte) while the inner policy sees fewer positions — each position once per worker — and the
    single-position PolicyWithCache / OwnedPolicyWithCache call their policy once per distinct key (the side to move is part of it)."""
    exe, blobf, d = harness
    my, op = random_positions(oracle, 24, seed=77, max_moves=40)
    my = np.concatenate([my, my[:8]]); op = np.concatenate([op, op[:8]])      # repeated roots: the same positions come up again
    roots = str(d / "roots_cache.u64")
    np.concatenate([my, op]).astype("<u8").tofile(roots)
    outs = {}
    for cache in ("0", "1"):
        for threads in ("1", "-2"):
            out = str(d / f"cache_{cache}_{threads}.bin")
            txt = subprocess.check_output([exe, "c4", blobf, roots, "150", "0", threads, out], env=dict(os.environ, LS_CACHE=cache)).decode()
            f = dict(zip(txt.split()[0::2], map(int, txt.split()[1::2])))
            outs[cache, threads] = (open(out, "rb").read(), f)
    for threads in ("1", "-2"):
        plain, cached = outs["0", threads], outs["1", threads]
        assert plain[0] == cached[0]
        assert plain[1]["positions"] == plain[1]["evals"] and plain[1]["hits"] == 0
        assert cached[1]["evals"] == plain[1]["evals"]                       # the trees ask the same questions
        assert cached[1]["hits"] + cached[1]["misses"] == cached[1]["evals"] and cached[1]["positions"] == cached[1]["misses"]
        assert cached[1]["hits"] > 0                                         # transpositions
        if threads == "1":   # one worker, one map: the eight repeated roots alone are a quarter of the work
            assert cached[1]["hits"] > 0.2 * cached[1]["evals"]
    txt = subprocess.check_output([exe, "cachepolicy"]).decode().splitlines()
    assert txt[0] == "borrowed calls 11 entries 11 owned calls 10 entries 10 wrong 0"
    assert txt[1] == "connect4 calls 2 entries 2 first 1 128"


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


def test_example_for_a_callers_own_game_builds_and_solves(tmp_path):
    """examples/host_trees_custom_game.cpp — what a caller with another Game writes: builds with the documented command line, the
    single-thread and the sharded driver agree, and every root is solved the way the game's theory says (a multiple of four is
    lost; otherwise take stones mod 4)."""
    exe = str(tmp_path / "host_trees")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-ffp-contract=off", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "host_trees_custom_game.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout.splitlines()
    assert len(out) == 13
    for line in out[:12]:
        stones = int(line.split()[0])
        take = int(line.split("take")[1].split()[0])
        again = int(line.rsplit("take", 1)[1].strip(" )"))
        assert ("lost" in line) == (stones % 4 == 0) and take == again
        if stones % 4:
            assert take == stones % 4


def test_lockstep_default_thread_count_respects_the_cpu_quota(harness):
    """threads = 0: the hardware concurrency cut to a cgroup CPU quota (more runnable threads than quota get throttled mid-round), at
    most 32."""
    exe, _, _ = harness
    n = int(subprocess.run([exe, "threads"], capture_output=True, text=True, timeout=60).stdout)
    limit = min(32, os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            limit = min(limit, max(1, -(-int(quota) // int(period))))
    except OSError:
        pass
    assert 1 <= n <= limit


def test_lockstep_policy_failure_reaches_the_caller(harness):
    """A BatchPolicy that throws — in eval_batch_begin() or in eval_batch_end() — under the pool driver, the two-halves driver, the
    sharded driver and the sharded driver over one CombiningPolicy: the caller gets the exception (no worker is left waiting for
    answers that never come), and the same policy object serves the next search."""
    exe, _, _ = harness
    for extra in ([], ["end"]):
        p = subprocess.run([exe, "policythrow"] + extra, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        where = "end" if extra else "begin"
        assert p.stdout.splitlines() == [ln for k in range(4) for ln in (f"shape {k}: caught policy failed in {where}", f"shape {k}: again 600 trees")]


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
                                (sa.MCTSConfig(exploration=sa.Exploration.Uct, c=1.4, fpu=sa.Fpu.ParentQ), dict(exploration=0, c=1.4, fpu=1), 40),
                                # the reference's own configuration: Fpu::Func(|| Normal(1.0, 0.1)) on the host trees, the same draws
                                (sa.reference_selfplay_mcts_config(), dict(fpu=2, fpu_value=1.0, fpu_std=0.1), 64)):
        got = eng.mcts_search_lockstep(scfg, my, op, explores)
        stats = got.pop("stats")
        fused = eng.mcts_search(scfg, my, op, explores)
        assert_search_equal(got, fused, "lockstep vs fused")
        ref = oracle.c4_mcts_search(parity_mcts_config(**okw), blob, my[:256], op[:256], explores, nn_mode=oracle.ACC_FMA)
        assert_search_equal({k: got[k][:256] for k in SEARCH_KEYS}, ref, "lockstep vs oracle")
        # (launches: at most explores + 1 per half of a worker's trees, 32 workers at most)
        assert 1 <= stats["rounds"] <= 64 * (explores + 1) and stats["positions_evaluated"] <= 4096 * (explores + 1)
    # PolicyNoise::Dirichlet on the host trees: the device path's sample for the same tree
    dcfg = sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=0.3, noise_weight=0.25, fpu=sa.Fpu.Func, fpu_value=1.0, fpu_std=0.1)
    got = eng.mcts_search_lockstep(dcfg, my, op, 48)
    got.pop("stats")
    assert_search_equal(got, eng.mcts_search(dcfg, my, op, 48), "lockstep vs fused, Dirichlet + Fpu::Func")
    with pytest.raises(sa.SynthesisAmdError) as e:
        eng.mcts_search_lockstep(sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=0.0, noise_weight=0.25), my[:4], op[:4], 8)
    assert e.value.code == -1
    # Fpu::Func(fn() -> f32) as the reference types it (config.rs:25): a function pointer in syn_mcts_config, called by the host trees
    # (mcts.rs:354). One root on one thread calls a counter-based function in the oracle's order: same tree, same number of calls.
    for i in (0, 5, 17):
        fn, calls = counter_fpu_fn()
        got = eng.mcts_search_lockstep(sa.MCTSConfig(fpu=sa.Fpu.FuncPtr, fpu_fn=fn), my[i:i + 1], op[i:i + 1], 150, host_threads=1)
        got.pop("stats")
        ofn, ocalls = counter_fpu_fn()
        oracle.set_fpu_fn(ofn)
        try:
            ref = oracle.c4_mcts_search(parity_mcts_config(fpu=3), blob, my[i:i + 1], op[i:i + 1], 150, nn_mode=oracle.ACC_FMA)
        finally:
            oracle.set_fpu_fn(None)
        assert_search_equal(got, ref, f"Fpu::Func(fn) through the C ABI, root {i}")
        assert calls[0] == ocalls[0] > 0
    # ... a function without state (any fn() -> f32) serves every tree of a many-thread search: Func(|| 1.0) is Fpu::Const(1.0)
    got = eng.mcts_search_lockstep(sa.MCTSConfig(fpu=sa.Fpu.FuncPtr, fpu_fn=lambda: 1.0), my[:512], op[:512], 40)
    got.pop("stats")
    assert_search_equal(got, eng.mcts_search(sa.parity_mcts_config(), my[:512], op[:512], 40), "Func(|| 1.0) == Const(1.0)")
    # the pointer is required, and a device kernel cannot call into the host: the fused entry points say so
    with pytest.raises(sa.SynthesisAmdError) as e:
        eng.mcts_search_lockstep(sa.MCTSConfig(fpu=sa.Fpu.FuncPtr), my[:4], op[:4], 8)
    assert e.value.code == -1 and "fpu_fn" in str(e.value)
    with pytest.raises(sa.SynthesisAmdError) as e:
        eng.mcts_search(sa.MCTSConfig(fpu=sa.Fpu.FuncPtr, fpu_fn=lambda: 1.0), my[:4], op[:4], 8)
    assert e.value.code == -5 and "host function" in str(e.value)
    with pytest.raises(sa.SynthesisAmdError) as e:
        eng.selfplay(sa.parity_rollout_config(8, mcts_cfg=sa.MCTSConfig(fpu=sa.Fpu.FuncPtr, fpu_fn=lambda: 1.0)), base_seed=1, n_games=2)
    assert e.value.code == -5
    eng.close()


@pytest.mark.gpu
def test_lockstep_selfplay_on_the_gpu_equals_the_fused_selfplay(oracle, golden_dir):
    """BASELINE configs[1] as worded — 4,096 concurrent games, host-side MCTS, batched HIP inference — against the fused kernel's
    games (syn_selfplay_run) game for game, and a sample of them against the oracle."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_rollout_config

    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    eng = sa.Engine(concurrent_games=4096, max_explores=64)
    eng.load_weights(blob)
    cfg = sa.parity_rollout_config(48)
    got = eng.selfplay_lockstep(cfg, 77, 4096, first_game=3)
    stats = got.pop("stats")
    fused = eng.selfplay(cfg, 77, 4096, first_game=3)
    assert_games_equal(got, fused, "lockstep self-play vs fused")
    ref = oracle.c4_selfplay(parity_rollout_config(48), blob, 77, 64, first_game=3, nn_mode=oracle.ACC_FMA)
    assert_games_equal({k: v[:64] for k, v in got.items()}, ref, "lockstep self-play vs oracle")
    assert stats["rounds"] >= 49 and stats["positions_evaluated"] <= int(got["plies"].sum()) * 49
    # the reference's own self-play configuration (study-connect4/src/main.rs:37-49) on the host trees
    rcfg = sa.parity_rollout_config(48, mcts_cfg=sa.reference_selfplay_mcts_config())
    got = eng.selfplay_lockstep(rcfg, 5, 1024, first_game=11)
    got.pop("stats")
    assert_games_equal(got, eng.selfplay(rcfg, 5, 1024, first_game=11), "lockstep self-play vs fused, Fpu::Func")
    dcfg = sa.parity_rollout_config(32, mcts_cfg=sa.MCTSConfig(root_policy_noise=sa.PolicyNoise.Dirichlet, noise_alpha=0.3, noise_weight=0.25))
    got = eng.selfplay_lockstep(dcfg, 8, 512)
    got.pop("stats")
    assert_games_equal(got, eng.selfplay(dcfg, 8, 512), "lockstep self-play vs fused, Dirichlet root noise")
    eng.close()


@pytest.mark.gpu
def test_lockstep_selfplay_reference_configuration_at_the_bench_shape(golden_dir):
    """The host-tree leg at the shape bench.py times it — 4,096 slots, 800 explores, more games than slots (a finished game's slot
    takes the next game) — in the configuration the reference itself plays (trained checkpoint, Fpu::Func(Normal(1.0, 0.1)),
    study-connect4/src/main.rs:37-49): every game equals the fused kernel's, which test_gpu_bench_shape.py holds to the oracle."""
    import synthesis_amd as sa

    blob = np.load(os.path.join(golden_dir, "c4net_trained_f32.npy"))
    eng = sa.Engine(concurrent_games=4096, max_explores=800)
    eng.load_weights(blob)
    cfg = sa.parity_rollout_config(800, mcts_cfg=sa.reference_selfplay_mcts_config())
    got = eng.selfplay_lockstep(cfg, 2024, 6144, first_game=100)
    stats = got.pop("stats")
    assert_games_equal(got, eng.selfplay(cfg, 2024, 6144, first_game=100), "host trees vs fused kernel, reference configuration, bench shape")
    assert stats["positions_evaluated"] <= int(got["plies"].sum()) * 801
    eng.close()


def test_lockstep_host_code_is_clean_under_asan_and_ubsan(oracle, golden_dir, tmp_path):
    """The host side above the C ABI (product code: include/synthesis_amd_lockstep.hpp — trees, worker pool, StdRng, the self-play
    loop) under AddressSanitizer + UndefinedBehaviorSanitizer: the harness rebuilt with both, searches on 4 threads and whole games
    on 3; any report aborts the binary. (GPU sanitizers are unavailable on the pool; the oracle side has its own such test.)"""
    exe = str(tmp_path / "lockstep_harness_san")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
                           "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "lockstep_harness.cpp"),
                           "-o", exe, "-L" + odir, "-loracle", "-Wl,-rpath," + odir])
    blob = str(tmp_path / "blob.f32")
    np.load(os.path.join(golden_dir, "c4net_blob_f32.npy")).astype("<f4").tofile(blob)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    from tests.test_gpu_parity import random_positions
    my, op = random_positions(oracle, 24, seed=5, max_moves=40)
    roots = str(tmp_path / "roots.u64")
    np.concatenate([my, op]).astype("<u8").tofile(roots)
    for args in (["c4", blob, roots, "60", "0", "4", str(tmp_path / "o1.bin")], ["c4", blob, roots, "40", "1", "2", str(tmp_path / "o2.bin")],
                 ["selfplay", blob, "6", "40", "0", "3", "9", "0", str(tmp_path / "o3.bin")],
                 ["selfplay", blob, "4", "30", "3", "2", "1", "7", str(tmp_path / "o4.bin")],
                 ["selfplay", blob, "4", "40", "4", "2", "2", "0", str(tmp_path / "o5.bin")],
                 ["selfplay", blob, "530", "4", "2", "3", "5", "0", str(tmp_path / "o6.bin")],
                 ["selfplay", blob, "200", "5", "4", "-3", "5", "0", str(tmp_path / "o7.bin")],
                 ["c4", blob, roots, "30", "5", "-2", str(tmp_path / "o8.bin")], ["nim"], ["nimthrow"]):
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600, env=dict(env, LS_CONCURRENT="48", LS_ASYNC="1"))
        assert p.returncode == 0, " ".join(args) + "\n" + p.stdout[-1500:] + p.stderr[-3000:]


def test_lockstep_host_threads_are_race_free_under_tsan(oracle, golden_dir, tmp_path):
    """The worker pool (blocks of trees from a shared counter), the sharded drivers (one policy and one thread per shard, game indices
    from a shared counter), a policy that computes on another thread between eval_batch_begin() and _end() while its caller advances
    the other half of its trees, and the exception hand-over, under ThreadSanitizer: any report fails the run."""
    exe = str(tmp_path / "lockstep_harness_tsan")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off", "-pthread",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "lockstep_harness.cpp"),
                           "-o", exe, "-L" + odir, "-loracle", "-Wl,-rpath," + odir])
    blob = str(tmp_path / "blob.f32")
    np.load(os.path.join(golden_dir, "c4net_blob_f32.npy")).astype("<f4").tofile(blob)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    from tests.test_gpu_parity import random_positions
    my, op = random_positions(oracle, 200, seed=6, max_moves=40)
    roots = str(tmp_path / "roots.u64")
    np.concatenate([my, op]).astype("<u8").tofile(roots)
    for args in (["c4", blob, roots, "30", "5", "4", str(tmp_path / "o1.bin")],
                 ["selfplay", blob, "530", "4", "2", "3", "5", "0", str(tmp_path / "o2.bin")],
                 ["selfplay", blob, "140", "6", "4", "4", "1", "0", str(tmp_path / "o3.bin")],
                 ["selfplay", blob, "300", "5", "0", "-4", "1", "0", str(tmp_path / "o4.bin")],
                 ["c4", blob, roots, "30", "0", "-3", str(tmp_path / "o5.bin")], ["nimthrow"], ["policythrow"], ["policythrow", "end"]):
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900, env=dict(env, LS_CONCURRENT="90", LS_ASYNC="1", LS_COMBINE="1" if args[0] == "selfplay" else "0"))
        assert p.returncode == 0, " ".join(args) + "\n" + p.stdout[-1500:] + p.stderr[-3000:]
