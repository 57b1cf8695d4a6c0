"""GPU parity of the conv policy/value network (Connect4ConvNet: slimnn Conv2d over the 2x7x9 bitplanes + Linear heads; the layers
are the reference's — slimnn/src/conv.rs:45-85, linear.rs:17-25 — the architecture is this build's, oracle/nn.hpp) behind the same
Policy::eval surface: batched evaluation, MCTS searches and whole self-play games against the CPU oracle.

Bars: network outputs EXACTLY equal to the oracle's ACC_FMA mode (the fma chains the matrix cores compute) and within 1e-5 of
the canonical slimnn loop order (north_star); visit counts / trajectories bit-exact."""
import numpy as np
import pytest

from tests.test_gpu_parity import NN_TOL, assert_search_equal, assert_selfplay_equal, random_positions

pytestmark = pytest.mark.gpu


def conv_blob(seed=20260101):
    """Fixed-seed initialisation (bench.make_conv_weights: U(+-1/sqrt(fan_in)) per layer, head weights x4)."""
    from bench import make_conv_weights
    from synthesis_amd.engine import CONV_NUM_PARAMS

    blob = make_conv_weights(seed)
    assert blob.size == CONV_NUM_PARAMS
    return blob


@pytest.fixture(scope="module")
def cblob():
    return conv_blob()


@pytest.fixture(scope="module")
def engine(cblob):
    import synthesis_amd as sa

    eng = sa.Engine(concurrent_games=1100, max_explores=800, device=0)
    eng.load_weights_conv(cblob)
    yield eng
    eng.close()


def test_conv_policy_eval_parity(engine, oracle, cblob):
    my, op = random_positions(oracle, 3000, seed=5, max_moves=62)
    my[0] = 0; op[0] = 0
    # edge and corner stones exercise the padding taps
    my[1] = (1 << 0) | (1 << 6) | (1 << 56) | (1 << 62); op[1] = (1 << 1) | (1 << 55)
    for n in (1, 15, 16, 17, 3000):
        lg, v = engine.policy_eval(my[:n], op[:n])
        ref_l, ref_v = oracle.c4conv_eval(cblob, my[:n], op[:n], mode=oracle.ACC_FMA)
        assert np.array_equal(lg.view(np.uint32), ref_l.view(np.uint32)), n
        assert np.array_equal(v.view(np.uint32), ref_v.view(np.uint32)), n
    can_l, can_v = oracle.c4conv_eval(cblob, my, op, mode=oracle.ACC_SLIMNN)
    assert np.abs(lg - can_l).max() <= NN_TOL and np.abs(v - can_v).max() <= NN_TOL
    assert np.abs(lg).max() > 0.05  # a real network, not zeros


def test_conv_mcts_search_visit_exact(engine, oracle, cblob):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config

    my, op = random_positions(oracle, 200, seed=17, max_moves=60)
    my[0] = 0; op[0] = 0
    for explores in (0, 1, 37, 200):
        got = engine.mcts_search(sa.parity_mcts_config(), my, op, explores)
        assert engine.last_launch_shape()[0] == 4  # lane-per-tree kernel
        ref = oracle.c4_mcts_search(parity_mcts_config(), cblob, my, op, explores, nn_mode=oracle.ACC_FMA, net="conv")
        assert_search_equal(got, ref, f"conv explores={explores}")
    got = engine.mcts_search(sa.parity_mcts_config(), my[:16], op[:16], 800)
    ref = oracle.c4_mcts_search(parity_mcts_config(), cblob, my[:16], op[:16], 800, nn_mode=oracle.ACC_FMA, net="conv")
    assert_search_equal(got, ref, "conv explores=800")
    # a non-FAST config family (ParentQ first-play urgency, plain UCT) goes through the general kernel variant
    ocfg = parity_mcts_config(exploration=0, c=1.4, fpu=1)
    scfg = sa.MCTSConfig(exploration=sa.Exploration(0), c=1.4, fpu=sa.Fpu(1))
    got = engine.mcts_search(scfg, my[:64], op[:64], 90, action_selection=0)
    ref = oracle.c4_mcts_search(ocfg, cblob, my[:64], op[:64], 90, action_selection=0, nn_mode=oracle.ACC_FMA, net="conv")
    assert_search_equal(got, ref, "conv uct/parent-q")


def test_conv_selfplay_matches_oracle(engine, oracle, cblob):
    import synthesis_amd as sa
    from tests.oracle_lib import parity_rollout_config

    got = engine.selfplay(sa.parity_rollout_config(60), base_seed=21, n_games=2300, counters=True)
    assert engine.last_launch_shape()[0] == 4 and engine.last_launch_shape()[2] == 256   # 4 waves up to 256 trees per CU
    ref = oracle.c4_selfplay(parity_rollout_config(60), cblob, 21, 2300, threads=8, nn_mode=oracle.ACC_FMA, net="conv")
    assert_selfplay_equal(got, ref, "conv self-play")
    for k in ("explores", "select_levels", "children_scanned", "expansions", "new_nodes", "policy_evals", "backprop_levels",
              "solver_children", "solved_hits", "max_depth"):
        assert got["counters"][k] == ref["counters"][k], k
    got = engine.selfplay(sa.parity_rollout_config(800), base_seed=4, n_games=4)
    ref = oracle.c4_selfplay(parity_rollout_config(800), cblob, 4, 4, threads=4, nn_mode=oracle.ACC_FMA, net="conv")
    assert_selfplay_equal(got, ref, "conv 800 explores")


def test_conv_16_wave_shape_and_policy_cache(oracle, cblob):
    """The 16- and 8-wave workgroup shapes (more than 512 / 256 trees per CU) — sampled games equal the oracle's; and the policy
    cache stays semantics-neutral with the conv network."""
    import synthesis_amd as sa
    from tests.oracle_lib import parity_rollout_config

    for slots, threads in ((140000, 1024), (70000, 512)):   # > 512 trees per CU: 16 waves; > 256: 8 waves
        eng = sa.Engine(concurrent_games=slots, max_explores=24)
        eng.load_weights_conv(cblob)
        got = eng.selfplay(sa.parity_rollout_config(24), base_seed=8, n_games=slots)
        assert eng.last_launch_shape()[0] == 4 and eng.last_launch_shape()[2] == threads
        for first in (0, slots // 2, slots - 32):
            ref = oracle.c4_selfplay(parity_rollout_config(24), cblob, 8, 32, first_game=first, threads=8, nn_mode=oracle.ACC_FMA,
                                     net="conv")
            sub = {k: got[k][first:first + 32] for k in ("plies", "final_kind", "states_bb", "pis", "vs", "actions", "root_nodes")}
            assert_selfplay_equal(sub, ref, f"conv {threads // 64} waves, games {first}..")
        eng.close()
    eng = sa.Engine(concurrent_games=1100, max_explores=100, policy_cache_log2=14)
    eng.load_weights_conv(cblob)
    got = eng.selfplay(sa.parity_rollout_config(100), base_seed=5, n_games=1500)
    ref = oracle.c4_selfplay(parity_rollout_config(100), cblob, 5, 1500, threads=8, nn_mode=oracle.ACC_FMA, net="conv")
    assert_selfplay_equal(got, ref, "conv with policy cache")
    hits, misses = eng.last_cache_stats()
    assert hits > 0 and misses > 0
    eng.close()


def test_conv_and_mlp_networks_switch(engine, oracle, cblob, golden_dir):
    """syn_load_weights / syn_load_weights_conv replace the engine's network in place; wrong sizes are rejected."""
    import os
    import synthesis_amd as sa

    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    my, op = random_positions(oracle, 64, seed=2)
    engine.load_weights(blob)
    lg, v = engine.policy_eval(my, op)
    ref_l, ref_v = oracle.c4net_eval(blob, my, op, mode=oracle.ACC_FMA)
    assert np.array_equal(lg, ref_l) and np.array_equal(v, ref_v)
    with pytest.raises(sa.SynthesisAmdError):
        engine.load_weights_conv(blob)  # 30,492 floats is not a Connect4ConvNet
    engine.load_weights_conv(cblob)
    lg, v = engine.policy_eval(my, op)
    ref_l, ref_v = oracle.c4conv_eval(cblob, my, op, mode=oracle.ACC_FMA)
    assert np.array_equal(lg, ref_l) and np.array_equal(v, ref_v)
    # the conv network lives in the lane-per-tree kernels: node pools past their 16-bit block addressing are refused, not mis-run
    big = sa.Engine(concurrent_games=64, max_explores=8000)
    with pytest.raises(sa.SynthesisAmdError) as ei:
        big.load_weights_conv(cblob)
    assert ei.value.code == -5   # SYN_ERR_UNSUPPORTED
    with pytest.raises(sa.SynthesisAmdError):
        big.trainer_init_conv(blob)  # wrong parameter count
    big.close()


def test_conv_learner_matches_oracle_and_feeds_selfplay(oracle, cblob, golden_dir):
    """The learner step of Connect4ConvNet (train_conv.cuh + adam_kernel): gradients, losses, weights and Adam moments after
    several steps bit-identical to oracle/train.hpp::ConvTrainer (itself checked against torch float64 goldens) — through
    syn_train_step with full and ragged batches and through syn_train_set_data + syn_train_epoch; the trained network,
    published on the device, then plays the oracle's games."""
    import os
    import synthesis_amd as sa
    from tests.oracle_lib import default_train_hyper, parity_rollout_config

    g = np.load(os.path.join(golden_dir, "conv_train_torch_goldens.npz"))
    my, op, tpi, tv, lrs = g["my_bb"], g["op_bb"], g["target_pi"], g["target_v"], g["lrs"]
    eng = sa.Engine(concurrent_games=256, max_explores=64)
    eng.load_weights_conv(cblob)
    hp = default_train_hyper(weight_decay=1e-3, policy_weight=0.7, value_weight=1.9)
    for B in (32, 31, 5, 1):
        eng.trainer_init_conv(cblob, weight_decay=1e-3, policy_weight=0.7, value_weight=1.9)
        l = eng.train_step(my[0][:B], op[0][:B], tpi[0][:B], tv[0][:B], 2e-3)
        st = eng.trainer_state()
        go, lo = oracle.convtrain_gradients(cblob, hp, my[0][:B], op[0][:B], tpi[0][:B], tv[0][:B])
        assert st["weights"].size == 12412
        assert np.array_equal(st["grads"].view(np.uint32), go.view(np.uint32)), B
        assert np.array_equal(l, lo), B
        wo, mo, vo, _, _ = oracle.convtrain_steps(cblob, hp, my[0][None, :B], op[0][None, :B], tpi[0][None, :B], tv[0][None, :B], [2e-3])
        assert np.array_equal(st["weights"], wo) and np.array_equal(st["m"], mo) and np.array_equal(st["v"], vo), B
    with pytest.raises(sa.SynthesisAmdError):
        eng.train_step(np.zeros(33, np.uint64), np.zeros(33, np.uint64), np.zeros((33, 9), np.float32), np.zeros((33, 3), np.float32), 1e-3)
    # the 8 golden steps one by one, then as one device-resident epoch
    eng.trainer_init_conv(cblob)
    ls = np.stack([eng.train_step(my[s], op[s], tpi[s], tv[s], float(lrs[s])) for s in range(8)])
    st = eng.trainer_state()
    wo, mo, vo, so, lo = oracle.convtrain_steps(cblob, default_train_hyper(), my, op, tpi, tv, lrs)
    assert st["step"] == 8 and np.array_equal(ls, lo)
    assert np.array_equal(st["weights"], wo) and np.array_equal(st["m"], mo) and np.array_equal(st["v"], vo)
    assert np.abs(st["weights"] - g["final_weights_f64"]).max() <= 1e-5   # and the torch float64 run
    eng.trainer_init_conv(cblob)
    eng.train_set_data(my.reshape(-1), op.reshape(-1), tpi.reshape(-1, 9), tv.reshape(-1, 3))
    le = eng.train_epoch(np.arange(128, dtype=np.int32), 32, 1e-3)
    wo4, _, _, _, lo4 = oracle.convtrain_steps(cblob, default_train_hyper(), my[:4], op[:4], tpi[:4], tv[:4], [1e-3] * 4)
    assert np.array_equal(le, lo4) and np.array_equal(eng.trainer_state()["weights"], wo4)
    # publish: the trained conv network becomes the self-play policy without a host round trip
    eng.trainer_publish_weights()
    got = eng.selfplay(sa.parity_rollout_config(48), base_seed=6, n_games=40)
    ref = oracle.c4_selfplay(parity_rollout_config(48), wo4, 6, 40, threads=8, nn_mode=oracle.ACC_FMA, net="conv")
    assert_selfplay_equal(got, ref, "self-play with the published trained conv network")
    # switching the trainer back to Connect4Net keeps working
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    eng.trainer_init(blob)
    assert eng.trainer_state()["weights"].size == 30492
    eng.close()


def test_conv_epoch_kernel_modes_agree(tmp_path, oracle, cblob, golden_dir):
    """The ways a conv epoch can run — the step spread over four workgroups of one XCD (L2-coherent barrier), the same kernel with
    the device-scope barrier, the one-workgroup kernel, and a four-workgroup launch that is thrown away and redone by the
    one-workgroup kernel — leave bit-identical weights, moments, gradients and losses, equal to the oracle's, for full and ragged
    minibatches. The modes are debug knobs, hence one child process each."""
    import os
    import subprocess
    import sys
    from tests.oracle_lib import default_train_hyper

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bpath = str(tmp_path / "cblob.npy")
    np.save(bpath, cblob)
    script = (
        "import sys, os, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "import synthesis_amd as sa\n"
        f"g = np.load(os.path.join({golden_dir!r}, 'conv_train_torch_goldens.npz')); blob = np.load({bpath!r})\n"
        "my = g['my_bb'].reshape(-1); op = g['op_bb'].reshape(-1); tpi = g['target_pi'].reshape(-1, 9); tv = g['target_v'].reshape(-1, 3)\n"
        "eng = sa.Engine(concurrent_games=64, max_explores=16); eng.load_weights_conv(blob); eng.trainer_init_conv(blob)\n"
        "if len(sys.argv) > 2: eng.trainer_set_precision(sys.argv[2])\n"
        "eng.train_set_data(my, op, tpi, tv)\n"
        "perm = np.random.default_rng(5).integers(0, my.size, size=9 * 32).astype(np.int32)\n"
        "l1 = eng.train_epoch(perm, 32, 1e-3); l2 = eng.train_epoch(perm[: 6 * 31], 31, 5e-4); l3 = eng.train_epoch(perm[: 7 * 5], 5, 2e-3)\n"
        "st = eng.trainer_state()\n"
        "np.savez(sys.argv[1], w=st['weights'], m=st['m'], v=st['v'], g=st['grads'], l=np.concatenate([l1, l2, l3]), perm=perm)\n")
    outs = {}
    for name, knobs in (("four_wgs", {}), ("four_wgs_device_scope", {"SYN_DEBUG": "1", "SYN_TRAIN_DEVICE_SCOPE": "1"}),
                        ("one_wg", {"SYN_DEBUG": "1", "SYN_TRAIN_CONV_MW": "0"}),
                        ("thrown_away", {"SYN_DEBUG": "1", "SYN_TRAIN_FORCE_ABORT": "1"})):
        env = {k: v for k, v in os.environ.items() if not k.startswith("SYN_")}
        env.update(knobs)
        path = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", script, path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        outs[name] = np.load(path)
    for name in ("four_wgs_device_scope", "one_wg", "thrown_away"):
        for k in ("w", "m", "v", "g", "l"):
            assert np.array_equal(outs["four_wgs"][k].view(np.uint32), outs[name][k].view(np.uint32)), (name, k)
    # the bf16 variant: four workgroups and one leave the same bits as well (same operands, same instruction per chain element)
    bouts = {}
    for name, knobs in (("four_wgs", {}), ("one_wg", {"SYN_DEBUG": "1", "SYN_TRAIN_CONV_MW": "0"})):
        env = {k: v for k, v in os.environ.items() if not k.startswith("SYN_")}
        env.update(knobs)
        path = str(tmp_path / ("bf16_" + name + ".npz"))
        r = subprocess.run([sys.executable, "-c", script, path, "bf16"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        bouts[name] = np.load(path)
    for k in ("w", "m", "v", "g", "l"):
        assert np.array_equal(bouts["four_wgs"][k].view(np.uint32), bouts["one_wg"][k].view(np.uint32)), ("bf16", k)
    assert not np.array_equal(bouts["four_wgs"]["w"], outs["four_wgs"]["w"])   # (and it IS the other arithmetic)
    # ... and the oracle's
    g = np.load(os.path.join(golden_dir, "conv_train_torch_goldens.npz"))
    my = g["my_bb"].reshape(-1); op = g["op_bb"].reshape(-1); tpi = g["target_pi"].reshape(-1, 9); tv = g["target_v"].reshape(-1, 3)
    perm = outs["four_wgs"]["perm"]
    hp = default_train_hyper()
    w, m, v, step = cblob, None, None, 0
    losses = []
    for B, n, lr in ((32, 9, 1e-3), (31, 6, 5e-4), (5, 7, 2e-3)):
        idx = perm[: n * B].reshape(n, B)
        w, m, v, step, lo = oracle.convtrain_steps(w, hp, my[idx], op[idx], tpi[idx], tv[idx], [lr] * n, m=m, v=v, step=step)
        losses.append(lo)
    assert np.array_equal(outs["four_wgs"]["l"], np.concatenate(losses))
    assert np.array_equal(outs["four_wgs"]["w"], w) and np.array_equal(outs["four_wgs"]["m"], m) and np.array_equal(outs["four_wgs"]["v"], v)


def test_conv_learner_bf16_variant_stays_within_bf16_error(oracle, cblob, golden_dir):
    """BASELINE configs[4] words the training step "bf16 conv": SYN_TRAIN_BF16 rounds every matrix operand of the conv learner to
    bf16 (bf16 matrix cores, f32 accumulation, f32 master weights / Adam). It is not bit-exact with anything — the bar is bf16's
    own error against the f32 learner (which is bit-identical to oracle/train.hpp::ConvTrainer): first-step gradients, losses, and
    the torch float64 goldens after the 8-step run; and the switch is refused for Connect4Net."""
    import os
    import synthesis_amd as sa
    from tests.oracle_lib import default_train_hyper

    g = np.load(os.path.join(golden_dir, "conv_train_torch_goldens.npz"))
    my, op, tpi, tv, lrs = g["my_bb"], g["op_bb"], g["target_pi"], g["target_v"], g["lrs"]
    eng = sa.Engine(concurrent_games=256, max_explores=64)
    eng.load_weights_conv(cblob)
    hp = default_train_hyper()
    for B in (32, 7):
        go, lo = oracle.convtrain_gradients(cblob, hp, my[0][:B], op[0][:B], tpi[0][:B], tv[0][:B])
        eng.trainer_init_conv(cblob)
        eng.trainer_set_precision("bf16")
        l = eng.train_step(my[0][:B], op[0][:B], tpi[0][:B], tv[0][:B], 1e-3)
        gb = eng.trainer_state()["grads"]
        assert not np.array_equal(gb, go)                                   # it really is another arithmetic
        cos = float(np.dot(gb.astype(np.float64), go.astype(np.float64)) / (np.linalg.norm(gb) * np.linalg.norm(go)))
        assert cos > 0.9995, (B, cos)
        assert np.abs(gb - go).max() <= 0.02 * np.abs(go).max(), B          # bf16: 8 bits of significand per operand
        assert np.abs(l - lo).max() <= 5e-3 * max(1.0, np.abs(lo).max()), (l, lo)
    # the 8 golden steps: bf16 gradients, f32 Adam — stays close to the float64 run (Adam normalises the step to ~lr per parameter)
    eng.trainer_init_conv(cblob)
    assert np.array_equal(eng.train_step(my[0], op[0], tpi[0], tv[0], 0.0),  # (f32 again after a re-init: the oracle's losses)
                          oracle.convtrain_gradients(cblob, hp, my[0], op[0], tpi[0], tv[0])[1])
    eng.trainer_init_conv(cblob)
    eng.trainer_set_precision("bf16")
    ls = np.stack([eng.train_step(my[s], op[s], tpi[s], tv[s], float(lrs[s])) for s in range(8)])
    st = eng.trainer_state()
    _, _, _, _, lo = oracle.convtrain_steps(cblob, hp, my, op, tpi, tv, lrs)
    assert st["step"] == 8 and np.abs(ls - lo).max() <= 2e-2
    assert np.abs(st["weights"] - g["final_weights_f64"]).max() <= 8 * float(np.max(lrs)) + 1e-6
    assert np.median(np.abs(st["weights"] - g["final_weights_f64"])) <= 2e-4
    # and as one device-resident epoch: the persistent kernel's bf16 instantiation gives the queued bf16 steps' bits
    eng.trainer_init_conv(cblob)
    eng.trainer_set_precision("bf16")
    eng.train_set_data(my.reshape(-1), op.reshape(-1), tpi.reshape(-1, 9), tv.reshape(-1, 3))
    le = eng.train_epoch(np.arange(128, dtype=np.int32), 32, 1e-3)
    we = eng.trainer_state()["weights"]
    eng.trainer_init_conv(cblob)
    eng.trainer_set_precision("bf16")
    lq = np.stack([eng.train_step(my[s], op[s], tpi[s], tv[s], 1e-3) for s in range(4)])
    assert np.array_equal(le, lq) and np.array_equal(we, eng.trainer_state()["weights"])
    # Connect4Net trains in f32 only
    eng.trainer_init(np.load(os.path.join(golden_dir, "c4net_blob_f32.npy")))
    with pytest.raises(sa.SynthesisAmdError) as ei:
        eng.trainer_set_precision("bf16")
    assert ei.value.code == -5   # SYN_ERR_UNSUPPORTED
    eng.close()


def test_conv_trained_checkpoint_matches_oracle(oracle, golden_dir):
    """A TRAINED conv network (tests/golden/c4conv_trained_f32.npy, produced by examples/train_connect4.py --net conv): sharp priors,
    deep narrow trees, many solved lines — searches and whole games still equal the oracle's."""
    import os
    import synthesis_amd as sa
    from tests.oracle_lib import parity_mcts_config, parity_rollout_config

    w = np.load(os.path.join(golden_dir, "c4conv_trained_f32.npy"))
    eng = sa.Engine(concurrent_games=1100, max_explores=800)
    eng.load_weights_conv(w)
    my, op = random_positions(oracle, 96, seed=23, max_moves=50)
    my[0] = 0; op[0] = 0
    got = eng.mcts_search(sa.parity_mcts_config(), my, op, 400)
    ref = oracle.c4_mcts_search(parity_mcts_config(), w, my, op, 400, nn_mode=oracle.ACC_FMA, net="conv")
    assert_search_equal(got, ref, "trained conv, 400 explores")
    got = eng.selfplay(sa.parity_rollout_config(200), base_seed=2, n_games=48, counters=True)
    ref = oracle.c4_selfplay(parity_rollout_config(200), w, 2, 48, threads=8, nn_mode=oracle.ACC_FMA, net="conv")
    assert_selfplay_equal(got, ref, "trained conv self-play")
    assert got["counters"]["max_depth"] == ref["counters"]["max_depth"] and got["counters"]["max_depth"] >= 12
    eng.close()
