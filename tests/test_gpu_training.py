"""GPU parity of the learner step and the replay de-duplication against the oracle (bit-exact: same
fixed-order f32 arithmetic) and against the torch float64 goldens (1e-5; no reference test covers this step)."""
import json
import os

import numpy as np
import pytest

from tests.conftest import free_port

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def blob(golden_dir):
    return np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "train_torch_goldens.npz"))


@pytest.fixture(scope="module")
def engine(blob):
    import synthesis_amd as sa

    eng = sa.Engine(concurrent_games=256, max_explores=64)
    eng.load_weights(blob)
    yield eng
    eng.close()


def test_train_steps_match_oracle_and_torch(engine, oracle, blob, gold):
    from tests.oracle_lib import default_train_hyper

    S, B = gold["my_bb"].shape
    X = np.stack([oracle.c4_features(gold["my_bb"][s], gold["op_bb"][s]) for s in range(S)])
    engine.trainer_init(blob)
    losses = np.stack([engine.train_step(gold["my_bb"][s], gold["op_bb"][s], gold["target_pi"][s], gold["target_v"][s],
                                         float(gold["lrs"][s])) for s in range(S)])
    st = engine.trainer_state()
    wo, mo, vo, step, lo = oracle.train_steps(blob, default_train_hyper(), X, gold["target_pi"], gold["target_v"], gold["lrs"])
    assert st["step"] == step == S
    assert np.array_equal(losses, lo)
    assert np.array_equal(st["weights"], wo) and np.array_equal(st["m"], mo) and np.array_equal(st["v"], vo)
    # independent second opinion (torch float64): losses, final weights
    assert np.abs(losses - gold["losses_f64"]).max() < 1e-5
    assert np.abs(st["weights"] - gold["weights_f64"]).max() < 1e-5
    # first-step gradient vs torch
    engine.trainer_init(blob)
    engine.train_step(gold["my_bb"][0], gold["op_bb"][0], gold["target_pi"][0], gold["target_v"][0], 1e-3)
    assert np.abs(engine.trainer_state()["grads"] - gold["grad0_f64"]).max() < 1e-6


def test_train_ragged_batches_and_hyper(engine, oracle, blob, gold):
    """Batch sizes around the 32-sample LDS chunk (1, 31, 33, 64, 100) and non-default weights / weight decay."""
    from tests.oracle_lib import default_train_hyper

    my = gold["my_bb"].reshape(-1); op = gold["op_bb"].reshape(-1)
    tpi = gold["target_pi"].reshape(-1, 9); tv = gold["target_v"].reshape(-1, 3)
    X = oracle.c4_features(my, op)
    hp = default_train_hyper(weight_decay=1e-3, policy_weight=0.7, value_weight=1.9)
    for B in (1, 31, 32, 33, 64, 100):
        engine.trainer_init(blob, weight_decay=1e-3, policy_weight=0.7, value_weight=1.9)
        l = engine.train_step(my[:B], op[:B], tpi[:B], tv[:B], 2e-3)
        st = engine.trainer_state()
        g, lo = oracle.train_gradients(blob, hp, X[:B], tpi[:B], tv[:B])
        assert np.array_equal(st["grads"], g), B
        assert np.array_equal(l, lo), B
        wo, mo, vo, _, _ = oracle.train_steps(blob, hp, X[:B][None], tpi[:B][None], tv[:B][None], [2e-3])
        assert np.array_equal(st["weights"], wo), B


def test_device_resident_epoch_equals_step_by_step(engine, oracle, blob, gold):
    """syn_train_set_data + syn_train_epoch (one call per epoch, batches gathered on the device by the sampler's
    permutation) give the bits of the same steps issued one by one — and of the oracle."""
    from tests.oracle_lib import default_train_hyper

    my = gold["my_bb"].reshape(-1); op = gold["op_bb"].reshape(-1)
    tpi = gold["target_pi"].reshape(-1, 9); tv = gold["target_v"].reshape(-1, 3)
    n, B = my.size, 32
    perm = np.random.default_rng(11).permutation(n).astype(np.int32)
    steps = n // B
    engine.trainer_init(blob)
    engine.train_set_data(my, op, tpi, tv)
    losses = engine.train_epoch(perm, B, 1e-3)
    st = engine.trainer_state()
    assert losses.shape == (steps, 2) and st["step"] == steps
    engine.trainer_init(blob)
    ref_losses = np.stack([engine.train_step(my[perm[s * B:(s + 1) * B]], op[perm[s * B:(s + 1) * B]],
                                             tpi[perm[s * B:(s + 1) * B]], tv[perm[s * B:(s + 1) * B]], 1e-3) for s in range(steps)])
    st2 = engine.trainer_state()
    assert np.array_equal(losses, ref_losses)
    assert np.array_equal(st["weights"], st2["weights"]) and np.array_equal(st["m"], st2["m"]) and np.array_equal(st["v"], st2["v"])
    X = oracle.c4_features(my, op)
    idx = perm[: steps * B].reshape(steps, B)
    wo, mo, vo, _, lo = oracle.train_steps(blob, default_train_hyper(), X[idx], tpi[idx], tv[idx], [1e-3] * steps)
    assert np.array_equal(st["weights"], wo) and np.array_equal(losses, lo)
    with pytest.raises(Exception):
        engine.train_epoch(np.array([n] * B, np.int32), B, 1e-3)  # index outside the uploaded buffer


@pytest.mark.parametrize("B,steps_per_epoch", [(32, (5, 4, 1)), (7, (3, 2)), (1, (2,))])
def test_persistent_epoch_kernel_chains_epochs(engine, oracle, blob, gold, B, steps_per_epoch):
    """The one-launch epoch kernel (train_epoch.cuh): consecutive epochs of odd and even length (the final network ends in either
    image buffer), short batches, non-default hyper-parameters and a changing learning rate — weights, moments, losses and the
    last step's gradients equal the oracle's, and the published network is the trained one."""
    from tests.oracle_lib import default_train_hyper

    my = gold["my_bb"].reshape(-1); op = gold["op_bb"].reshape(-1)
    tpi = gold["target_pi"].reshape(-1, 9); tv = gold["target_v"].reshape(-1, 3)
    n = my.size
    hp = default_train_hyper(weight_decay=1e-3, policy_weight=0.7, value_weight=1.9)
    engine.trainer_init(blob, weight_decay=1e-3, policy_weight=0.7, value_weight=1.9)
    engine.train_set_data(my, op, tpi, tv)
    X = oracle.c4_features(my, op)
    rng = np.random.default_rng(100 + B)
    all_idx, lrs, losses = [], [], []
    for e, steps in enumerate(steps_per_epoch):
        perm = rng.integers(0, n, size=steps * B).astype(np.int32)
        lr = 1e-3 * (e + 1)
        losses.append(engine.train_epoch(perm, B, lr))
        all_idx.append(perm.reshape(steps, B)); lrs += [lr] * steps
    idx = np.concatenate(all_idx)
    st = engine.trainer_state()
    wo, mo, vo, _, lo = oracle.train_steps(blob, hp, X[idx], tpi[idx], tv[idx], lrs)
    w_prev = oracle.train_steps(blob, hp, X[idx[:-1]], tpi[idx[:-1]], tv[idx[:-1]], lrs[:-1])[0] if len(lrs) > 1 else blob
    go, _ = oracle.train_gradients(w_prev, hp, X[idx[-1]], tpi[idx[-1]], tv[idx[-1]])
    assert st["step"] == len(lrs)
    assert np.array_equal(np.concatenate(losses), lo)
    assert np.array_equal(st["weights"], wo) and np.array_equal(st["m"], mo) and np.array_equal(st["v"], vo)
    assert np.array_equal(st["grads"], go)
    # the step-by-step path continues from the same state (it reads the fragment images the epoch kernel left behind)
    l = engine.train_step(my[:B], op[:B], tpi[:B], tv[:B], 2e-3)
    wo2, _, _, _, lo2 = oracle.train_steps(blob, hp, np.concatenate([X[idx], X[:B][None]]), np.concatenate([tpi[idx], tpi[:B][None]]),
                                           np.concatenate([tv[idx], tv[:B][None]]), lrs + [2e-3])
    assert np.array_equal(engine.trainer_state()["weights"], wo2) and np.array_equal(l, lo2[-1])
    engine.load_weights(blob)


def test_epoch_kernel_recovery_keeps_the_learner(engine, oracle, blob, gold, monkeypatch):
    """syn_train_epoch's persistent kernel needs its 16 workgroups resident together and gives up after ~10 s when another kernel
    holds the CUs. The call must not lose the learner then: the state snapshotted before the launch is put back and the epoch
    runs through the queued per-step launches — same bits. SYN_TRAIN_FORCE_ABORT takes that path on purpose."""
    from tests.oracle_lib import default_train_hyper

    my = gold["my_bb"].reshape(-1); op = gold["op_bb"].reshape(-1)
    tpi = gold["target_pi"].reshape(-1, 9); tv = gold["target_v"].reshape(-1, 3)
    B, steps = 32, 5
    hp = default_train_hyper()
    engine.trainer_init(blob)
    engine.train_set_data(my, op, tpi, tv)
    X = oracle.c4_features(my, op)
    perm = np.random.default_rng(7).integers(0, my.size, size=2 * steps * B).astype(np.int32)
    idx = perm.reshape(2 * steps, B)
    l0 = engine.train_epoch(perm[: steps * B], B, 1e-3)               # the persistent kernel
    monkeypatch.setenv("SYN_DEBUG", "1")
    monkeypatch.setenv("SYN_TRAIN_FORCE_ABORT", "1")
    l1 = engine.train_epoch(perm[steps * B:], B, 1e-3)                # persistent kernel, thrown away, restored, queued launches
    monkeypatch.delenv("SYN_TRAIN_FORCE_ABORT")
    st = engine.trainer_state()
    wo, mo, vo, _, lo = oracle.train_steps(blob, hp, X[idx], tpi[idx], tv[idx], [1e-3] * (2 * steps))
    assert st["step"] == 2 * steps
    assert np.array_equal(np.concatenate([l0, l1]), lo)
    assert np.array_equal(st["weights"], wo) and np.array_equal(st["m"], mo) and np.array_equal(st["v"], vo)
    # and the persistent kernel continues from what the fallback left behind (both fragment images are current)
    l2 = engine.train_epoch(perm[: 3 * B], B, 2e-3)
    wo2, _, _, _, lo2 = oracle.train_steps(blob, hp, np.concatenate([X[idx], X[idx[:3]]]), np.concatenate([tpi[idx], tpi[idx[:3]]]),
                                           np.concatenate([tv[idx], tv[idx[:3]]]), [1e-3] * (2 * steps) + [2e-3] * 3)
    assert np.array_equal(engine.trainer_state()["weights"], wo2) and np.array_equal(l2, lo2[-3:])
    engine.load_weights(blob)


def test_data_parallel_gradient_path(engine, oracle, blob, gold):
    """configs[4] plumbing on one GPU: two 'ranks' compute gradients of their half-batches into caller-owned device
    buffers, the sum is applied with grad_scale = 1/2 — equals (to f32 rounding) one step on the combined batch, and is
    bit-identical to the oracle doing the same two-gradient average."""
    import torch
    from tests.oracle_lib import default_train_hyper

    B = 32
    my = torch.from_numpy(gold["my_bb"][:2].astype(np.int64)).cuda()
    op = torch.from_numpy(gold["op_bb"][:2].astype(np.int64)).cuda()
    tpi = torch.from_numpy(gold["target_pi"][:2]).cuda()
    tv = torch.from_numpy(gold["target_v"][:2]).cuda()
    g = [torch.zeros(30492, dtype=torch.float32, device="cuda") for _ in range(2)]
    engine.trainer_init(blob)
    for r in range(2):
        engine.train_gradients_device(my[r].data_ptr(), op[r].data_ptr(), tpi[r].data_ptr(), tv[r].data_ptr(), B, g[r].data_ptr())
    total = g[0] + g[1]  # what an all-reduce(sum) leaves on every rank
    engine.train_apply_device(total.data_ptr(), 1e-3, grad_scale=0.5)
    st = engine.trainer_state()
    hp = default_train_hyper()
    X = np.stack([oracle.c4_features(gold["my_bb"][s], gold["op_bb"][s]) for s in range(2)])
    g0, _ = oracle.train_gradients(blob, hp, X[0], gold["target_pi"][0], gold["target_v"][0])
    g1, _ = oracle.train_gradients(blob, hp, X[1], gold["target_pi"][1], gold["target_v"][1])
    assert np.array_equal(g[0].cpu().numpy(), g0) and np.array_equal(g[1].cpu().numpy(), g1)
    wo, mo, vo, _ = oracle.train_adam(blob, hp, (g0 + g1) * np.float32(0.5), 1e-3, np.zeros_like(blob), np.zeros_like(blob), 0)
    assert np.array_equal(st["weights"], wo)
    # the same update as one 64-sample step, up to f32 summation order
    engine.trainer_init(blob)
    engine.train_step(gold["my_bb"][:2].reshape(-1), gold["op_bb"][:2].reshape(-1), gold["target_pi"][:2].reshape(-1, 9),
                      gold["target_v"][:2].reshape(-1, 3), 1e-3)
    assert np.abs(engine.trainer_state()["weights"] - wo).max() < 1e-5


def test_publish_weights_feeds_selfplay(engine, oracle, blob, gold):
    engine.trainer_init(blob)
    for s in range(4):
        engine.train_step(gold["my_bb"][s], gold["op_bb"][s], gold["target_pi"][s], gold["target_v"][s], 1e-3)
    w = engine.trainer_state()["weights"]
    engine.trainer_publish_weights()
    logits, value = engine.policy_eval(gold["my_bb"][5], gold["op_bb"][5])
    fl, fv = oracle.c4net_eval(w, gold["my_bb"][5], gold["op_bb"][5], mode=oracle.ACC_FMA)
    assert np.array_equal(logits, fl) and np.array_equal(value, fv)
    engine.load_weights(blob)


def test_replay_deduplicate_matches_oracle(engine, oracle, blob):
    """data.rs:196-235 on real self-play output: openings repeat across games (the empty board appears in every game),
    targets of identical states are averaged in buffer order."""
    import synthesis_amd as sa

    r = engine.selfplay(sa.parity_rollout_config(24), base_seed=77, n_games=400)
    my, op, pi, v = [], [], [], []
    for g in range(400):
        n = r["plies"][g]
        my.append(r["states_bb"][g, :n, 0]); op.append(r["states_bb"][g, :n, 1]); pi.append(r["pis"][g, :n]); v.append(r["vs"][g, :n])
    my = np.concatenate(my); op = np.concatenate(op); pi = np.concatenate(pi); v = np.concatenate(v)
    got = engine.replay_deduplicate(my, op, pi, v)
    ref = oracle.dedup(my, op, pi, v)
    assert got["num"].size == ref["num"].size < my.size
    assert int(ref["num"].max()) == 400  # the empty board
    for k in ("my_bb", "op_bb", "num", "pis", "vs"):
        assert np.array_equal(got[k], ref[k]), k
    assert int(got["num"].sum()) == my.size
    empty = engine.replay_deduplicate(my[:0], op[:0], pi[:0], v[:0])
    assert empty["num"].size == 0
    one = engine.replay_deduplicate(my[:1], op[:1], pi[:1], v[:1])
    assert one["num"].tolist() == [1] and np.array_equal(one["pis"][0], pi[0])


def test_data_parallel_learner_two_ranks(oracle, blob, gold, tmp_path):
    """BASELINE configs[4] end to end on one box: two processes (torch.distributed.run, gloo, both on GPU 0) each train on
    half of every batch through DataParallelLearner; both ranks end with identical weights, equal to the oracle applying
    Adam to the mean of the two shard gradients, and within float rounding of the single-process full-batch steps."""
    import subprocess
    import sys

    from tests.oracle_lib import default_train_hyper

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "tests", "dp_learner_worker.py"), str(tmp_path)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in ("weights", "m", "v", "losses"):
        assert np.array_equal(r0[k], r1[k]), k
    assert int(r0["step"]) == 4
    hp = default_train_hyper()
    w, m, v, step = blob.copy(), np.zeros_like(blob), np.zeros_like(blob), 0
    B = gold["my_bb"].shape[1]; h = B // 2
    for s in range(4):
        gs, ls = [], []
        for r in range(2):
            sl = slice(r * h, (r + 1) * h)
            X = oracle.c4_features(gold["my_bb"][s, sl], gold["op_bb"][s, sl])
            g, l = oracle.train_gradients(w, hp, X, gold["target_pi"][s, sl], gold["target_v"][s, sl])
            gs.append(g); ls.append(l)
        w, m, v, step = oracle.train_adam(w, hp, (gs[0] + gs[1]) * np.float32(0.5), float(gold["lrs"][s]), m, v, step)
        assert np.array_equal(r0["losses"][s], (ls[0] + ls[1]) / np.float32(2)), s
    assert np.array_equal(r0["weights"], w) and np.array_equal(r0["m"], m) and np.array_equal(r0["v"], v)
    # against the single-process run on the full batches (torch float64 goldens after 4 steps are not stored: use the oracle)
    X = np.stack([oracle.c4_features(gold["my_bb"][s], gold["op_bb"][s]) for s in range(4)])
    wf, _, _, _, _ = oracle.train_steps(blob, hp, X, gold["target_pi"][:4], gold["target_v"][:4], gold["lrs"][:4])
    assert np.abs(r0["weights"] - wf).max() < 1e-5


def test_data_parallel_step_through_rccl_at_world_size_1(oracle, blob, gold, tmp_path):
    """BASELINE configs[4]'s literal step — gradients -> RCCL all-reduce -> Adam, enqueued on one stream with no host wait in between
    (syn_train_gradients_enqueue / syn_train_apply_enqueue, DataParallelLearner._step_device) — on the process group a one-GPU box can
    build: backend nccl (= RCCL), world size 1. The message really crosses RCCL (30,494 floats: gradients + the two loss sums); the
    result equals the oracle's full-batch steps bit for bit."""
    import subprocess
    import sys

    from tests.conftest import free_port
    from tests.oracle_lib import default_train_hyper

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DP_BACKEND="nccl")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "tests", "dp_learner_worker.py"), str(tmp_path)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    r0 = np.load(tmp_path / "rank0.npz")
    assert int(r0["step"]) == 4
    hp = default_train_hyper()
    X = np.stack([oracle.c4_features(gold["my_bb"][s], gold["op_bb"][s]) for s in range(4)])
    wf, mf, vf, _, lf = oracle.train_steps(blob, hp, X, gold["target_pi"][:4], gold["target_v"][:4], gold["lrs"][:4])
    assert np.array_equal(r0["weights"], wf) and np.array_equal(r0["m"], mf) and np.array_equal(r0["v"], vf)
    assert np.array_equal(r0["losses"], lf)


@pytest.mark.parametrize("net,mode", [("mlp", "loop"), ("conv", "loop"), ("mlp", "data-parallel"), ("conv", "data-parallel")])
def test_training_example_keeps_two_ranks_identical(tmp_path, net, mode):
    """examples/train_connect4.py with two ranks (gloo, both on GPU 0), both networks, both readings of BASELINE configs[4].
    loop (synthesis_amd.learner.LearningLoop): every rank plays its shard, rank 0 gathers / de-duplicates / trains with the
      persistent epoch kernel, the weights are broadcast once per iteration — both ranks end with the same weights, and they are
      bit-identical to the ONE-rank run's (the loop does not depend on the number of ranks).
    data-parallel (DataParallelLearner): the new games are all-gathered into ONE global replay buffer, de-duplicated identically on
      both ranks, every global batch is split evenly, one fused all-reduce (gradients + losses) per step — bit-identical weights
      on both ranks. Uneven game counts (601 games over 2 ranks) exercise the remainder handling."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    port = free_port()
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    common = ["--iterations", "2", "--games-per-train", "601", "--explores", "40", "--epochs", "1", "--concurrent", "512",
              "--dist-backend", "gloo", "--net", net] + (["--data-parallel"] if mode == "data-parallel" else [])
    prefix = str(tmp_path / "w")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(root, "examples", "train_connect4.py")] + common + ["--dump-weights", prefix],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0, out.stdout.decode()[-3000:]
    w0, w1 = np.load(prefix + ".rank0.npy"), np.load(prefix + ".rank1.npy")
    assert np.array_equal(w0.view(np.uint32), w1.view(np.uint32))
    from bench import make_conv_weights, make_weights
    assert not np.array_equal(w0, make_conv_weights(20260101) if net == "conv" else make_weights(20211003))   # the weights did train
    lines = [json.loads(l) for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 2 and lines[0]["games"] == 601 and lines[1]["optimiser_steps"] > 0
    if mode == "loop":
        single = str(tmp_path / "s")
        one = subprocess.run([sys.executable, os.path.join(root, "examples", "train_connect4.py")] + common + ["--dump-weights", single],
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert one.returncode == 0, one.stdout.decode()[-3000:]
        ws = np.load(single + ".rank0.npy")
        assert np.array_equal(ws.view(np.uint32), w0.view(np.uint32)), "one rank and two ranks train the same network"
        l1 = [json.loads(l) for l in one.stdout.decode().splitlines() if l.startswith("{")]
        assert [r["unique"] for r in l1] == [r["unique"] for r in lines] and l1[1]["epoch_losses"] == lines[1]["epoch_losses"]


def test_epoch_kernel_modes_agree(tmp_path, golden_dir):
    """The three ways an epoch can run — the persistent kernel with its workers on one XCD (L2-coherent step barrier), the
    same kernel with the device-scope barrier it falls back to when the workers are spread over XCDs, and the queued
    two-kernels-per-step path — leave bit-identical weights, moments and losses. The modes are debug knobs read once per
    process, hence one child process each."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, os, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "import synthesis_amd as sa\n"
        f"g = np.load(os.path.join({golden_dir!r}, 'train_torch_goldens.npz')); blob = np.load(os.path.join({golden_dir!r}, 'c4net_blob_f32.npy'))\n"
        "my = g['my_bb'].reshape(-1); op = g['op_bb'].reshape(-1); tpi = g['target_pi'].reshape(-1, 9); tv = g['target_v'].reshape(-1, 3)\n"
        "eng = sa.Engine(concurrent_games=64, max_explores=16); eng.load_weights(blob); eng.trainer_init(blob)\n"
        "eng.train_set_data(my, op, tpi, tv)\n"
        "perm = np.random.default_rng(5).integers(0, my.size, size=11 * 32).astype(np.int32)\n"
        "l1 = eng.train_epoch(perm, 32, 1e-3); l2 = eng.train_epoch(perm[: 6 * 32], 32, 5e-4)\n"
        "st = eng.trainer_state()\n"
        "np.savez(sys.argv[1], w=st['weights'], m=st['m'], v=st['v'], g=st['grads'], l=np.concatenate([l1, l2]))\n")
    outs = {}
    for name, knobs in (("one_xcd", {}), ("device_scope", {"SYN_DEBUG": "1", "SYN_TRAIN_DEVICE_SCOPE": "1"}),
                        ("queued", {"SYN_DEBUG": "1", "SYN_TRAIN_QUEUED": "1"})):
        env = {k: v for k, v in os.environ.items() if not k.startswith("SYN_")}
        env.update(knobs)
        path = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", script, path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        outs[name] = np.load(path)
    for name in ("device_scope", "queued"):
        for k in ("w", "m", "v", "g", "l"):
            assert np.array_equal(outs["one_xcd"][k].view(np.uint32), outs[name][k].view(np.uint32)), (name, k)


def test_learner_epochs_are_bit_reproducible_under_load(blob, gold):
    """Long-run determinism stress of the three persistent learner kernels (train_epoch_kernel for Connect4Net,
    train_conv_epoch_kernel_mw in f32 and in bf16 for Connect4ConvNet; loop body = alpha_zero.rs:72-94): 500 epochs in all, back to
    back, WHILE self-play launches of a second engine keep arriving on the same GPU (half of the CUs taken, L2 / HBM shared — the
    neighbours a learner has in the N-rank loop), and once per learner behind a launch that holds EVERY CU (the epoch kernel's
    workers cannot all be resident: it waits, or falls back to queued launches from its snapshot). Every epoch must leave exactly
    the bits of the undisturbed one — weights, both Adam moments, last gradient and per-step losses. A missing hazard wait, an LDS
    reuse without a barrier or a stale cross-workgroup read shows up here as a flipped bit in some of the runs."""
    import threading

    import synthesis_amd as sa
    from tests.test_gpu_convnet import conv_blob

    my = gold["my_bb"].reshape(-1); op = gold["op_bb"].reshape(-1)
    tpi = gold["target_pi"].reshape(-1, 9); tv = gold["target_v"].reshape(-1, 3)
    # a few thousand positions: the goldens' positions repeated with rotated targets (the learner only sees arrays)
    reps = max(1, 4096 // my.size)
    my = np.tile(my, reps); op = np.tile(op, reps)
    tpi = np.concatenate([np.roll(tpi, k, axis=0) for k in range(reps)]); tv = np.concatenate([np.roll(tv, k, axis=0) for k in range(reps)])
    perm = np.random.default_rng(11).permutation(my.size).astype(np.int32)[: 192 * 32]

    learn = sa.Engine(concurrent_games=64, max_explores=16)
    learn.load_weights(blob)
    small = sa.Engine(concurrent_games=2048, max_explores=200)     # 128 workgroups: half of the CUs
    small.load_weights(blob)
    big = sa.Engine(concurrent_games=65536, max_explores=200)      # one workgroup on every CU
    big.load_weights(blob)
    cfg = sa.parity_rollout_config(200)
    stop = threading.Event()

    def neighbours():
        g = 0
        while not stop.is_set():
            small.selfplay(cfg, base_seed=5, n_games=4096, first_game=g, outputs=False)
            g += 4096

    def epoch(kind):
        if kind == "mlp":
            learn.trainer_init(blob)
        else:
            learn.trainer_init_conv(conv_blob())
            if kind == "conv_bf16":
                learn.trainer_set_precision("bf16")
        learn.train_set_data(my, op, tpi, tv)
        losses = learn.train_epoch(perm, 32, 1e-3)
        st = learn.trainer_state()
        return [st[k].view(np.uint32).copy() for k in ("weights", "m", "v", "grads")] + [losses.view(np.uint32).copy()]

    refs = {kind: epoch(kind) for kind in ("mlp", "conv", "conv_bf16")}
    th = threading.Thread(target=neighbours)
    th.start()
    try:
        for kind, n in (("mlp", 168), ("conv", 166), ("conv_bf16", 166)):
            for i in range(n):
                got = epoch(kind)
                for a, b, name in zip(got, refs[kind], ("weights", "m", "v", "grads", "losses")):
                    assert np.array_equal(a, b), f"{kind}: epoch {i} differs from the undisturbed run in {name}"
    finally:
        stop.set()
        th.join()
    # behind a launch that holds every CU
    for kind in ("mlp", "conv", "conv_bf16"):
        box = {}
        t2 = threading.Thread(target=lambda: box.update(r=big.selfplay(cfg, base_seed=9, n_games=196608, outputs=False)))
        t2.start()
        import time
        time.sleep(0.3)
        got = epoch(kind)
        t2.join()
        for a, b, name in zip(got, refs[kind], ("weights", "m", "v", "grads", "losses")):
            assert np.array_equal(a, b), f"{kind}: epoch beside a full-chip launch differs in {name}"
    for e in (learn, small, big):
        e.close()
