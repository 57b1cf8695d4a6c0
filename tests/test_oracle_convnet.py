"""CPU checks of the oracle's Connect4ConvNet (oracle/nn.hpp): it is the composition of the slimnn layers the reference's own KATs
pin (Conv2d: conv.rs:92-602 replayed in test_oracle_kats.py; Linear: linear.rs:105-112), equals an independent float64 numpy
evaluation to 1e-5, and its two accumulation orders (slimnn's loops / the matrix-core chains) agree to 1e-5."""
import numpy as np

from tests.test_gpu_convnet import conv_blob


def planes_of(my, op):
    x = np.zeros((2, 7, 9), np.float32)
    for pl, bb in enumerate((int(my), int(op))):
        for r in range(7):
            for c in range(9):
                x[pl, r, c] = (bb >> (r + 7 * c)) & 1
    return x


def test_convnet_is_the_composition_of_the_pinned_layers(oracle):
    blob = conv_blob()
    rng = np.random.RandomState(3)
    my = np.zeros(40, np.uint64); op = np.zeros(40, np.uint64)
    for i in range(40):
        cells = rng.permutation(63)[: rng.randint(0, 40)]
        for k, cell in enumerate(cells):
            if k % 2 == 0: my[i] |= np.uint64(1) << np.uint64(cell)
            else: op[i] |= np.uint64(1) << np.uint64(cell)
    my[1] = (1 << 0) | (1 << 6) | (1 << 56) | (1 << 62)  # the four corners: every padding tap
    lg, v, raw = oracle.c4conv_eval(blob, my, op, mode=oracle.ACC_SLIMNN, raw=True)
    lg_f, v_f, raw_f = oracle.c4conv_eval(blob, my, op, mode=oracle.ACC_FMA, raw=True)
    assert np.abs(raw - raw_f).max() <= 1e-5 and np.abs(v - v_f).max() <= 1e-5
    cw = blob[:288].reshape(16, 2, 3, 3); cb = blob[288:304]; hw = blob[304:304 + 12 * 1008].reshape(12, 1008); hb = blob[-12:]
    for i in range(40):
        x = planes_of(my[i], op[i])
        # the layer entry points the KAT tests exercise
        y = oracle.conv2d(cw, cb, x, 1, 1, 1, mode=oracle.ACC_SLIMNN)[0]
        y = np.maximum(y, 0).reshape(-1)
        out = oracle.linear(hw, hb, y[None], mode=oracle.ACC_SLIMNN)[0]
        assert np.array_equal(out.view(np.uint32), raw[i].view(np.uint32)), i
        # independent float64 evaluation
        xp = np.zeros((2, 9, 11)); xp[:, 1:8, 1:10] = x
        y64 = np.zeros((16, 7, 9))
        for co in range(16):
            acc = np.full((7, 9), float(cb[co]))
            for ci in range(2):
                for k1 in range(3):
                    for k2 in range(3):
                        acc += float(cw[co, ci, k1, k2]) * xp[ci, k1:k1 + 7, k2:k2 + 9]
            y64[co] = acc
        out64 = hw.astype(np.float64) @ np.maximum(y64, 0).reshape(-1) + hb
        assert np.abs(out64 - raw[i]).max() <= 1e-5, i
        e = np.exp(out64[9:] - out64[9:].max())
        assert np.abs(e / e.sum() - v[i]).max() <= 1e-5 and np.array_equal(lg[i], raw[i][:9])


def test_conv_trainer_matches_torch_float64_goldens(oracle, golden_dir):
    """oracle/train.hpp::ConvTrainer (forward, log_softmax + kl_div, backward, Adam — all in fixed f32 orders) against this
    container's torch in float64 (tests/golden/make_conv_train_golden.py): first gradient, per-step losses, weights after 8 steps."""
    import os
    from tests.oracle_lib import default_train_hyper

    g = np.load(os.path.join(golden_dir, "conv_train_torch_goldens.npz"))
    blob = conv_blob()
    hp = default_train_hyper()
    grad, losses = oracle.convtrain_gradients(blob, hp, g["my_bb"][0], g["op_bb"][0], g["target_pi"][0], g["target_v"][0])
    assert np.abs(grad - g["first_grad_f64"]).max() <= 1e-6 and np.abs(losses - g["losses_f64"][0]).max() <= 1e-6
    w, m, v, step, ls = oracle.convtrain_steps(blob, hp, g["my_bb"], g["op_bb"], g["target_pi"], g["target_v"], g["lrs"])
    assert step == 8 and np.abs(ls - g["losses_f64"]).max() <= 1e-6
    assert np.abs(w - g["final_weights_f64"]).max() <= 1e-5
    assert np.abs(w - blob).max() > 1e-3  # the weights did move
