"""ctypes binding of oracle/liboracle.so — the CPU restatement of the reference (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module. The product package
(synthesis_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


class MctsConfig(C.Structure):
    _fields_ = [
        ("exploration", C.c_int), ("c", C.c_float),
        ("solve", C.c_int), ("correct_values_on_solve", C.c_int), ("select_solved_nodes", C.c_int),
        ("auto_extend", C.c_int),
        ("fpu", C.c_int), ("fpu_value", C.c_float),
        ("noise", C.c_int), ("noise_alpha", C.c_float), ("noise_weight", C.c_float),
        ("fpu_std", C.c_float),
    ]


class RolloutConfig(C.Structure):
    _fields_ = [
        ("num_explores", C.c_int), ("random_actions_until", C.c_int), ("sample_actions_until", C.c_int),
        ("stop_games_when_solved", C.c_int),
        ("value_target", C.c_int), ("vt_p", C.c_float), ("vt_from", C.c_float), ("vt_to", C.c_float),
        ("action", C.c_int),
        ("mcts", MctsConfig),
    ]


def parity_mcts_config(**kw):
    """policy_mcts_cfg of study-connect4/src/main.rs:58-66 (the deterministic variant of the self-play config)."""
    d = dict(exploration=1, c=3.0, solve=1, correct_values_on_solve=1, select_solved_nodes=1, auto_extend=1,
             fpu=0, fpu_value=1.0, noise=0, noise_alpha=0.0, noise_weight=0.0, fpu_std=0.0)
    d.update(kw)
    return MctsConfig(**d)


def parity_rollout_config(num_explores=800, **kw):
    """rollout_cfg of study-connect4/src/main.rs:28-50 with the deterministic FPU; explores per BASELINE.json."""
    mc = kw.pop("mcts", None) or parity_mcts_config()
    d = dict(num_explores=num_explores, random_actions_until=1, sample_actions_until=30, stop_games_when_solved=0,
             value_target=1, vt_p=0.0, vt_from=0.0, vt_to=0.0, action=1, mcts=mc)
    d.update(kw)
    return RolloutConfig(**d)


class TrainHyper(C.Structure):
    _fields_ = [("weight_decay", C.c_float), ("policy_weight", C.c_float), ("value_weight", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]


def default_train_hyper(**kw):
    """study-connect4/src/main.rs:20-24 + Adam::default()."""
    d = dict(weight_decay=1e-6, policy_weight=1.0, value_weight=1.0, beta1=0.9, beta2=0.999, eps=1e-8)
    d.update(kw)
    return TrainHyper(**d)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    ACC_SLIMNN = 0
    ACC_FMA = 1
    ACC_F16X2 = 2   # the engine's SYN_NET_ARITH_F16X2 arithmetic (oracle/nn_f16x2.hpp)

    def __init__(self, lib):
        self.lib = lib
        lib.orc_c4_selfplay.restype = C.c_double
        lib.orc_c4_gather_experience.restype = C.c_double
        lib.orc_c4conv_selfplay.restype = C.c_double
        lib.orc_c4net_num_params.restype = C.c_size_t
        lib.orc_c4conv_num_params.restype = C.c_size_t
        lib.orc_outcome_value.restype = C.c_float

    # ---- Fpu::Func(fn() -> f32) (config.rs:25): the function configurations with fpu = 3 call (None uninstalls it)
    FPU_FN = C.CFUNCTYPE(C.c_float)

    def set_fpu_fn(self, fn):
        self._fpu_fn = None if fn is None else (fn if isinstance(fn, self.FPU_FN) else self.FPU_FN(fn))   # keeps the callback alive
        self.lib.orc_set_fpu_fn(self._fpu_fn if self._fpu_fn is not None else self.FPU_FN())

    # ---- outcome
    def outcome_cmp(self, a, b):
        """a, b: None or (kind, turns) with kind in 'Lose','Draw','Win'."""
        k = {"Lose": 0, "Draw": 1, "Win": 2}
        aa = (0, 0, 0) if a is None else (1, k[a[0]], a[1])
        bb = (0, 0, 0) if b is None else (1, k[b[0]], b[1])
        return self.lib.orc_outcome_cmp(aa[0], aa[1], C.c_uint(aa[2]), bb[0], bb[1], C.c_uint(bb[2]))

    # ---- connect4
    def c4_won(self, bb):
        return bool(self.lib.orc_c4_won(C.c_uint64(bb)))

    def c4_play(self, moves):
        n = len(moves)
        mv = np.asarray(moves, dtype=np.uint8)
        over = np.zeros(n, np.uint8); lb = np.zeros(n, np.uint8); la = np.zeros(n, np.uint8)
        my = C.c_uint64(); op = C.c_uint64(); player = C.c_int(); winner = C.c_int()
        rr = C.c_float(); rb = C.c_float(); rm = C.c_float()
        self.lib.orc_c4_play(_p(mv), n, _p(over), _p(lb), _p(la), C.byref(my), C.byref(op), C.byref(player),
                             C.byref(winner), C.byref(rr), C.byref(rb), C.byref(rm))
        return dict(over=over.astype(bool), legal_before=lb.astype(bool), legal_after=la.astype(bool), my_bb=my.value,
                    op_bb=op.value, player=player.value, winner=winner.value, reward_red=rr.value,
                    reward_black=rb.value, reward_to_move=rm.value)

    def c4_features(self, my_bb, op_bb):
        my = np.ascontiguousarray(my_bb, dtype=np.uint64); op = np.ascontiguousarray(op_bb, dtype=np.uint64)
        out = np.zeros((my.size, 63), np.float32)
        self.lib.orc_c4_features(_p(my), _p(op), int(my.size), _p(out))
        return out

    # ---- layers
    def linear(self, W, b, x, mode=0):
        W = np.ascontiguousarray(W, np.float32); b = np.ascontiguousarray(b, np.float32)
        x = np.ascontiguousarray(x, np.float32).reshape(-1, W.shape[1])
        y = np.zeros((x.shape[0], W.shape[0]), np.float32)
        self.lib.orc_linear_forward(W.shape[1], W.shape[0], _p(W), _p(b), _p(x), x.shape[0], _p(y), mode)
        return y

    def conv2d(self, W, b, x, row_pad, col_pad, stride, mode=0):
        W = np.ascontiguousarray(W, np.float32); b = np.ascontiguousarray(b, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim == 3:
            x = x[None]
        cout, cin, k, _ = W.shape
        n, _, h_in, w_in = x.shape
        h_out = (h_in + 2 * row_pad - k) // stride + 1
        w_out = (w_in + 2 * col_pad - k) // stride + 1
        y = np.zeros((n, cout, h_out, w_out), np.float32)
        rc = self.lib.orc_conv2d_forward(cin, cout, k, row_pad, col_pad, stride, h_in, w_in, h_out, w_out, _p(W), _p(b),
                                         _p(x), n, _p(y), mode)
        assert rc == 0
        return y

    def relu(self, x):
        y = np.array(x, np.float32).ravel().copy()
        self.lib.orc_relu(_p(y), int(y.size))
        return y

    def det_expf(self, x):
        x = np.ascontiguousarray(x, np.float32).ravel()
        y = np.zeros_like(x)
        self.lib.orc_det_expf(_p(x), _p(y), int(x.size))
        return y

    def det_logf(self, x):
        x = np.ascontiguousarray(x, np.float32).ravel()
        y = np.zeros_like(x)
        self.lib.orc_det_logf(_p(x), _p(y), int(x.size))
        return y

    def std_normal_from_bits(self, k):
        """oracle/noise.hpp fpu_std_normal: the standard normal the Fpu::Func draw makes of 23 random bits"""
        k = np.ascontiguousarray(k, np.uint32)
        out = np.zeros(k.size, np.float32)
        self.lib.orc_std_normal_from_bits(_p(k), int(k.size), _p(out))
        return out

    def noise_fpu_normals(self, tree_seed, n_scans, mean=0.0, std=1.0):
        """Fpu::Func draws [scan][child slot] of one tree (oracle/noise.hpp noise_fpu_normal)"""
        out = np.zeros((int(n_scans), 9), np.float32)
        self.lib.orc_noise_fpu_normals(C.c_uint64(int(tree_seed)), int(n_scans), C.c_float(mean), C.c_float(std), _p(out))
        return out

    def tanh(self, x):
        y = np.ascontiguousarray(x, np.float32).ravel().copy()
        self.lib.orc_tanh(_p(y), int(y.size))
        return y.reshape(np.shape(x))

    def softmax_slimnn(self, x):
        x = np.ascontiguousarray(x, np.float32).ravel()
        y = np.zeros_like(x)
        self.lib.orc_softmax_slimnn(_p(x), _p(y), int(x.size))
        return y

    def softmax_stable(self, x):
        x = np.ascontiguousarray(x, np.float32).ravel()
        y = np.zeros_like(x)
        self.lib.orc_softmax_stable(_p(x), _p(y), int(x.size))
        return y

    # ---- Connect4Net
    def c4net_eval(self, blob, my_bb, op_bb, mode=0):
        blob = np.ascontiguousarray(blob, np.float32)
        assert blob.size == self.lib.orc_c4net_num_params()
        my = np.ascontiguousarray(my_bb, dtype=np.uint64); op = np.ascontiguousarray(op_bb, dtype=np.uint64)
        n = int(my.size)
        logits = np.zeros((n, 9), np.float32); value = np.zeros((n, 3), np.float32)
        self.lib.orc_c4net_eval(_p(blob), _p(my), _p(op), n, _p(logits), _p(value), mode)
        return logits, value

    # ---- Connect4ConvNet (oracle/nn.hpp)
    def mfma_f16_k32(self, a_bits, b_bits, c):
        """One v_mfma_f32_16x16x32_f16 output per row: a_bits, b_bits [n][32] uint16 (f16 bit patterns, the instruction's k order), c [n]."""
        a = np.ascontiguousarray(a_bits, np.uint16); b = np.ascontiguousarray(b_bits, np.uint16); c = np.ascontiguousarray(c, np.float32)
        out = np.zeros(c.size, np.float32)
        self.lib.orc_mfma_f16_k32(_p(a), _p(b), _p(c), int(c.size), _p(out))
        return out

    def f16_round_trip(self, x):
        x = np.ascontiguousarray(x, np.float32)
        bits = np.zeros(x.size, np.uint16); back = np.zeros(x.size, np.float32)
        self.lib.orc_f16_round_trip(_p(x), int(x.size), _p(bits), _p(back))
        return bits, back

    def f16x2_plan(self, blob):
        blob = np.ascontiguousarray(blob, np.float32)
        e = np.zeros(15, np.int32); bnd = np.zeros(5, np.float64)
        ok = self.lib.orc_f16x2_plan(_p(blob), _p(e), _p(bnd))
        return dict(ok=bool(ok), activation_exp=e[:5].tolist(), weight_exp=e[5:10].tolist(), rescale_exp=e[10:14].tolist(), out_exp=int(e[14]),
                    bound=bnd.tolist())

    def c4conv_eval(self, blob, my_bb, op_bb, mode=0, raw=False):
        blob = np.ascontiguousarray(blob, np.float32)
        assert blob.size == self.lib.orc_c4conv_num_params()
        my = np.ascontiguousarray(my_bb, dtype=np.uint64); op = np.ascontiguousarray(op_bb, dtype=np.uint64)
        n = int(my.size)
        logits = np.zeros((n, 9), np.float32); value = np.zeros((n, 3), np.float32); raw12 = np.zeros((n, 12), np.float32)
        self.lib.orc_c4conv_eval(_p(blob), _p(my), _p(op), n, _p(logits), _p(value), _p(raw12), mode)
        return (logits, value, raw12) if raw else (logits, value)

    # ---- rng
    def stdrng_u32(self, seed, n, rounds=12):
        out = np.zeros(n, np.uint32)
        self.lib.orc_stdrng_seed_from_u64_u32s(C.c_uint64(seed), rounds, n, _p(out))
        return out

    def stdrng_from_seed_u64(self, seed32, n, rounds=12):
        out = np.zeros(n, np.uint64)
        self.lib.orc_stdrng_from_seed_u64s(bytes(seed32), rounds, n, _p(out))
        return out

    def chacha_block(self, key8, counter, stream, rounds):
        key = np.ascontiguousarray(key8, np.uint32); out = np.zeros(16, np.uint32)
        self.lib.orc_chacha_block(_p(key), C.c_uint64(counter), C.c_uint64(stream), rounds, _p(out))
        return out

    def gen_range_u8(self, seed, rng_range, n):
        out = np.zeros(n, np.uint8)
        self.lib.orc_stdrng_gen_range_u8(C.c_uint64(seed), C.c_uint8(rng_range), n, _p(out))
        return out

    def weighted_index(self, seed, w, n):
        w = np.ascontiguousarray(w, np.float32); out = np.zeros(n, np.int32)
        self.lib.orc_stdrng_weighted_index(C.c_uint64(seed), _p(w), int(w.size), n, _p(out))
        return out

    # ---- tictactoe KATs
    def ttt_kat(self, which, seed=0, rounds=12, max_explores=100000):
        cs = (C.c_int * 9)(); ck = (C.c_int * 9)(); ct = (C.c_uint * 9)()
        ba = C.c_int(); nn = C.c_uint(); rs = C.c_int(); rk = C.c_int(); rt = C.c_uint()
        sp = (C.c_float * 9)(); tq = (C.c_float * 3)()
        self.lib.orc_ttt_kat(which, C.c_uint64(seed), rounds, max_explores, cs, ck, ct, C.byref(ba), C.byref(nn),
                             C.byref(rs), C.byref(rk), C.byref(rt), sp, tq)
        kinds = ["Lose", "Draw", "Win"]
        return dict(child=[(kinds[ck[i]], ct[i]) if cs[i] else None for i in range(9)], best_action_q=ba.value,
                    nodes_len=nn.value, root=(kinds[rk.value], rt.value) if rs.value else None,
                    search_policy=np.array(sp[:], np.float32), target_q=np.array(tq[:], np.float32))

    def ttt_root_priors(self, seed=0):
        pr = (C.c_float * 9)(); n = C.c_int()
        self.lib.orc_ttt_root_priors(C.c_uint64(seed), pr, C.byref(n))
        return np.array(pr[: n.value], np.float32)

    # ---- connect4 MCTS search
    def c4_mcts_search(self, cfg, blob, my_bb, op_bb, explores, action_selection=1, nn_mode=1, net="mlp"):
        blob = np.ascontiguousarray(blob, np.float32)
        my = np.ascontiguousarray(my_bb, dtype=np.uint64); op = np.ascontiguousarray(op_bb, dtype=np.uint64)
        n = int(my.size)
        r = dict(child_N=np.zeros((n, 9), np.float32), child_W=np.zeros((n, 9, 3), np.float32),
                 child_P=np.zeros((n, 9), np.float32), child_sol=np.zeros((n, 9, 3), np.int32),
                 root_stat=np.zeros((n, 4), np.float32), root_sol=np.zeros((n, 3), np.int32),
                 num_nodes=np.zeros(n, np.uint32), best_action=np.zeros(n, np.int32),
                 target_pi=np.zeros((n, 9), np.float32), target_q=np.zeros((n, 3), np.float32))
        fn = self.lib.orc_c4conv_mcts_search if net == "conv" else self.lib.orc_c4_mcts_search
        fn(C.byref(cfg), _p(blob), nn_mode, _p(my), _p(op), n, explores, action_selection,
           _p(r["child_N"]), _p(r["child_W"]), _p(r["child_P"]), _p(r["child_sol"]),
           _p(r["root_stat"]), _p(r["root_sol"]), _p(r["num_nodes"]), _p(r["best_action"]),
           _p(r["target_pi"]), _p(r["target_q"]))
        return r

    def c4_mcts_search_rollout(self, cfg, seed, my_bb, op_bb, explores, action_selection=1):
        """MCTS over RolloutPolicy (rollout.rs:8-31); root i plays its rollouts from StdRng::seed_from_u64(seed + i)."""
        my = np.ascontiguousarray(my_bb, dtype=np.uint64); op = np.ascontiguousarray(op_bb, dtype=np.uint64)
        n = int(my.size)
        r = dict(child_N=np.zeros((n, 9), np.float32), child_W=np.zeros((n, 9, 3), np.float32),
                 child_P=np.zeros((n, 9), np.float32), child_sol=np.zeros((n, 9, 3), np.int32),
                 root_stat=np.zeros((n, 4), np.float32), root_sol=np.zeros((n, 3), np.int32),
                 num_nodes=np.zeros(n, np.uint32), best_action=np.zeros(n, np.int32),
                 target_pi=np.zeros((n, 9), np.float32), target_q=np.zeros((n, 3), np.float32))
        self.lib.orc_c4_mcts_search_rollout(C.byref(cfg), C.c_uint64(seed), _p(my), _p(op), n, explores, action_selection,
                                            _p(r["child_N"]), _p(r["child_W"]), _p(r["child_P"]), _p(r["child_sol"]),
                                            _p(r["root_stat"]), _p(r["root_sol"]), _p(r["num_nodes"]), _p(r["best_action"]),
                                            _p(r["target_pi"]), _p(r["target_q"]))
        return r

    def c4_frozen_search(self, cfg, seeds, rng_words, my_bb, op_bb, explores, action_selection=1, blob=None, nn_mode=1):
        """FrozenMCTS (evaluator.rs:230-534) over RolloutPolicy on StdRng(seeds[i]) from word rng_words[i] (blob given: over the
        network instead). Returns the records plus "rng_words" = stream position after each search."""
        my = np.ascontiguousarray(my_bb, dtype=np.uint64); op = np.ascontiguousarray(op_bb, dtype=np.uint64)
        n = int(my.size)
        sd = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (n,)))
        words = np.array(np.broadcast_to(np.asarray(rng_words, dtype=np.uint64), (n,)), dtype=np.uint64)
        ex = np.ascontiguousarray(np.broadcast_to(np.asarray(explores, dtype=np.int32), (n,)))
        r = dict(child_N=np.zeros((n, 9), np.float32), child_cum=np.zeros((n, 9), np.float32),
                 child_P=np.zeros((n, 9), np.float32), child_sol=np.zeros((n, 9, 3), np.int32),
                 root_stat=np.zeros((n, 2), np.float32), root_sol=np.zeros((n, 3), np.int32),
                 num_nodes=np.zeros(n, np.uint32), best_action=np.zeros(n, np.int32))
        kind = 1 if blob is None else 0
        b = np.ascontiguousarray(blob, np.float32) if blob is not None else np.zeros(1, np.float32)
        self.lib.orc_c4_frozen_search(C.byref(cfg), kind, _p(b), nn_mode, _p(sd), _p(words), _p(my), _p(op), n, _p(ex),
                                      action_selection, _p(r["child_N"]), _p(r["child_cum"]), _p(r["child_P"]),
                                      _p(r["child_sol"]), _p(r["root_stat"]), _p(r["root_sol"]), _p(r["num_nodes"]),
                                      _p(r["best_action"]))
        r["rng_words"] = words
        return r

    def c4_mcts_vs_mcts(self, rollout_cfg, player, p1_explores, p2_explores, seed, rollout_action=1):
        """evaluator.rs:200-228. Returns (reward for the first player, moves, stream words consumed after each ply)."""
        moves = np.zeros(63, np.uint8); words = np.zeros(63, np.uint64); n = C.c_int(0)
        self.lib.orc_c4_mcts_vs_mcts.restype = C.c_float
        r = self.lib.orc_c4_mcts_vs_mcts(C.byref(rollout_cfg), rollout_action, player, p1_explores, p2_explores,
                                         C.c_uint64(seed), _p(moves), C.byref(n), _p(words))
        return float(r), moves[:n.value].copy(), words[:n.value].copy()

    def c4_eval_against_rollout(self, policy_cfg, policy_explores, blob, rollout_cfg, player, opponent_explores, seed,
                                policy_action=1, rollout_action=1, nn_mode=1):
        """evaluator.rs:163-198. Returns (reward for the first player, moves, stream words consumed after each ply)."""
        moves = np.zeros(63, np.uint8); words = np.zeros(63, np.uint64); n = C.c_int(0)
        b = np.ascontiguousarray(blob, np.float32)
        self.lib.orc_c4_eval_against_rollout.restype = C.c_float
        r = self.lib.orc_c4_eval_against_rollout(C.byref(policy_cfg), policy_explores, policy_action, _p(b), nn_mode,
                                                 C.byref(rollout_cfg), rollout_action, player, opponent_explores,
                                                 C.c_uint64(seed), _p(moves), C.byref(n), _p(words))
        return float(r), moves[:n.value].copy(), words[:n.value].copy()

    def c4_eval_against_old(self, policy_cfg, policy_explores, blob1, blob2, policy_action=1, nn_mode=1):
        """evaluator.rs:129-160. Returns (reward for the first player, moves)."""
        moves = np.zeros(63, np.uint8); n = C.c_int(0)
        b1 = np.ascontiguousarray(blob1, np.float32); b2 = np.ascontiguousarray(blob2, np.float32)
        self.lib.orc_c4_eval_against_old.restype = C.c_float
        r = self.lib.orc_c4_eval_against_old(C.byref(policy_cfg), policy_explores, policy_action, _p(b1), _p(b2), nn_mode,
                                             _p(moves), C.byref(n))
        return float(r), moves[:n.value].copy()

    # ---- training step / dedup
    def train_gradients(self, blob, hp, X, tpi, tv):
        blob = np.ascontiguousarray(blob, np.float32)
        X = np.ascontiguousarray(X, np.float32); tpi = np.ascontiguousarray(tpi, np.float32); tv = np.ascontiguousarray(tv, np.float32)
        grad = np.zeros_like(blob); losses = np.zeros(2, np.float32)
        self.lib.orc_train_gradients(_p(blob), C.byref(hp), _p(X), _p(tpi), _p(tv), int(X.shape[0]), _p(grad), _p(losses))
        return grad, losses

    def train_steps(self, blob, hp, X, tpi, tv, lrs, m=None, v=None, step=0):
        """X[n_steps][B][63] ...; returns (blob', m', v', step', losses[n_steps][2])."""
        blob = np.array(blob, np.float32).copy()
        X = np.ascontiguousarray(X, np.float32); tpi = np.ascontiguousarray(tpi, np.float32); tv = np.ascontiguousarray(tv, np.float32)
        n_steps, B = X.shape[0], X.shape[1]
        lrs = np.ascontiguousarray(np.broadcast_to(np.asarray(lrs, np.float32), (n_steps,)))
        m = np.zeros_like(blob) if m is None else np.array(m, np.float32).copy()
        v = np.zeros_like(blob) if v is None else np.array(v, np.float32).copy()
        st = C.c_longlong(step)
        losses = np.zeros((n_steps, 2), np.float32)
        self.lib.orc_train_steps(_p(blob), C.byref(hp), _p(X), _p(tpi), _p(tv), B, n_steps, _p(lrs), _p(m), _p(v),
                                 C.byref(st), _p(losses))
        return blob, m, v, st.value, losses

    # ---- the learner step for Connect4ConvNet (oracle/train.hpp ConvTrainer)
    def convtrain_gradients(self, blob, hp, my_bb, op_bb, tpi, tv):
        blob = np.ascontiguousarray(blob, np.float32)
        my = np.ascontiguousarray(my_bb, np.uint64).ravel(); op = np.ascontiguousarray(op_bb, np.uint64).ravel()
        tpi = np.ascontiguousarray(tpi, np.float32); tv = np.ascontiguousarray(tv, np.float32)
        g = np.zeros(blob.size, np.float32); losses = np.zeros(2, np.float32)
        self.lib.orc_convtrain_gradients(_p(blob), C.byref(hp), _p(my), _p(op), _p(tpi), _p(tv), int(my.size), _p(g), _p(losses))
        return g, losses

    def convtrain_steps(self, blob, hp, my_bb, op_bb, tpi, tv, lrs, m=None, v=None, step=0):
        """my_bb / op_bb [n_steps][B], tpi [n_steps][B][9], tv [n_steps][B][3]; returns (blob', m', v', step', losses[n_steps][2])."""
        blob = np.array(blob, np.float32).copy()
        my = np.ascontiguousarray(my_bb, np.uint64); op = np.ascontiguousarray(op_bb, np.uint64)
        tpi = np.ascontiguousarray(tpi, np.float32); tv = np.ascontiguousarray(tv, np.float32)
        n_steps, B = my.shape[0], my.shape[1]
        lrs = np.ascontiguousarray(np.broadcast_to(np.asarray(lrs, np.float32), (n_steps,)))
        m = np.zeros_like(blob) if m is None else np.array(m, np.float32).copy()
        v = np.zeros_like(blob) if v is None else np.array(v, np.float32).copy()
        st = C.c_longlong(step)
        losses = np.zeros((n_steps, 2), np.float32)
        self.lib.orc_convtrain_steps(_p(blob), C.byref(hp), _p(my), _p(op), _p(tpi), _p(tv), B, n_steps, _p(lrs), _p(m), _p(v),
                                     C.byref(st), _p(losses))
        return blob, m, v, st.value, losses

    def train_adam(self, blob, hp, grad, lr, m, v, step):
        blob = np.array(blob, np.float32).copy(); m = np.array(m, np.float32).copy(); v = np.array(v, np.float32).copy()
        grad = np.ascontiguousarray(grad, np.float32)
        st = C.c_longlong(step)
        self.lib.orc_train_adam(_p(blob), C.byref(hp), _p(grad), C.c_float(lr), _p(m), _p(v), C.byref(st))
        return blob, m, v, st.value

    def dedup(self, my_bb, op_bb, pis, vs):
        my = np.ascontiguousarray(my_bb, np.uint64); op = np.ascontiguousarray(op_bb, np.uint64)
        pis = np.ascontiguousarray(pis, np.float32); vs = np.ascontiguousarray(vs, np.float32)
        n = int(my.size)
        o = dict(my_bb=np.zeros(n, np.uint64), op_bb=np.zeros(n, np.uint64), pis=np.zeros((n, 9), np.float32),
                 vs=np.zeros((n, 3), np.float32), num=np.zeros(n, np.uint32))
        self.lib.orc_dedup.restype = C.c_size_t
        m = self.lib.orc_dedup(_p(my), _p(op), _p(pis), _p(vs), C.c_size_t(n), _p(o["my_bb"]), _p(o["op_bb"]),
                               _p(o["pis"]), _p(o["vs"]), _p(o["num"]))
        return {k: a[:m] for k, a in o.items()}

    # ---- self-play
    def c4_gather_experience(self, cfg, blob, seed, n_games, num_workers, use_cache=True, nn_mode=1):
        """gather_experience / run_n_games as the reference runs them (alpha_zero.rs:120-209): num_workers + 1 workers, one StdRng per
        worker (seed * (num_workers + 1) + i_worker) through all of its games. Records in worker order."""
        blob = np.ascontiguousarray(blob, np.float32)
        r = dict(plies=np.zeros(n_games, np.int32), final_kind=np.zeros(n_games, np.uint8), counters=np.zeros(12, np.uint64),
                 states_bb=np.zeros((n_games, 63, 2), np.uint64), pis=np.zeros((n_games, 63, 9), np.float32),
                 vs=np.zeros((n_games, 63, 3), np.float32), actions=np.zeros((n_games, 63), np.uint8),
                 root_nodes=np.zeros((n_games, 63), np.uint32))
        self.lib.orc_c4_gather_experience(C.byref(cfg), _p(blob), nn_mode, C.c_uint64(seed), n_games, int(num_workers) + 1, int(use_cache),
                                          _p(r["plies"]), _p(r["states_bb"]), _p(r["pis"]), _p(r["vs"]), _p(r["actions"]),
                                          _p(r["root_nodes"]), _p(r["final_kind"]), _p(r["counters"]))
        return r

    def c4_selfplay(self, cfg, blob, base_seed, n_games, first_game=0, threads=1, use_cache=False, nn_mode=1,
                    outputs=True, net="mlp"):
        blob = np.ascontiguousarray(blob, np.float32)
        r = dict(plies=np.zeros(n_games, np.int32), final_kind=np.zeros(n_games, np.uint8),
                 counters=np.zeros(12, np.uint64))
        if outputs:
            r.update(states_bb=np.zeros((n_games, 63, 2), np.uint64), pis=np.zeros((n_games, 63, 9), np.float32),
                     vs=np.zeros((n_games, 63, 3), np.float32), actions=np.zeros((n_games, 63), np.uint8),
                     root_nodes=np.zeros((n_games, 63), np.uint32))
        g = lambda k: _p(r[k]) if k in r else None
        fn = self.lib.orc_c4conv_selfplay if net == "conv" else self.lib.orc_c4_selfplay
        secs = fn(C.byref(cfg), _p(blob), nn_mode, C.c_uint64(base_seed), C.c_uint64(first_game),
                  n_games, threads, int(use_cache), _p(r["plies"]), g("states_bb"), g("pis"),
                  g("vs"), g("actions"), g("root_nodes"), _p(r["final_kind"]), _p(r["counters"]))
        r["seconds"] = secs
        names = ["explores", "select_levels", "children_scanned", "expansions", "new_nodes", "policy_evals",
                 "backprop_levels", "solver_children", "solved_hits", "cache_hits", "cache_misses", "max_depth"]
        r["counters"] = {k: int(v) for k, v in zip(names, r["counters"])}
        return r


def build():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def load():
    if not os.path.exists(LIB):
        build()
    return Oracle(C.CDLL(LIB))
