"""ctypes binding of the C ABI (include/synthesis_amd.h) + the Engine convenience class.

The library is loaded from synthesis_amd/libsynthesis_amd.so (built in-tree by `make -C synthesis_amd/csrc` or
__graft_entry__.build()). Nothing here computes: every method forwards to a HIP kernel launch inside the library.
"""
import ctypes as C
import os

import numpy as np

from .config import CEngineConfig, CMctsConfig, CRolloutConfig, MCTSConfig, RolloutConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
NUM_PARAMS = 30492  # Connect4Net: 63->128->96->64->48->12 (study-connect4/src/policies.rs:20-24)
CONV_NUM_PARAMS = 12412  # Connect4ConvNet: Conv2d<2,16,3,pad 1> + ReLU + Linear<1008,12> (include/synthesis_amd.h)

# every symbol include/synthesis_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "syn_default_rollout_config", "syn_engine_create", "syn_engine_destroy", "syn_last_error", "syn_load_weights",
    "syn_load_weights_conv", "syn_set_network_arithmetic", "syn_get_network_arithmetic", "syn_f16x2_plan_of_blob",
    "syn_policy_eval_batch", "syn_policy_eval_batch_device", "syn_eval_ctx_create", "syn_eval_ctx_submit", "syn_eval_ctx_wait",
    "syn_eval_ctx_eval", "syn_eval_ctx_last_error", "syn_eval_ctx_destroy", "syn_features_batch", "syn_linear_forward",
    "syn_conv2d_forward", "syn_activation_forward", "syn_mcts_search", "syn_mcts_search_rollout", "syn_mcts_search_lockstep", "syn_selfplay_run_lockstep", "syn_frozen_search_rollout", "syn_selfplay_run", "syn_progress", "syn_cancel", "syn_trainer_set_precision", "syn_last_timing", "syn_last_launch_shape", "syn_last_cache_stats", "syn_debug_stdrng_u32",
    "syn_debug_math", "syn_debug_fast_div", "syn_debug_small_int_math", "syn_debug_calibrate", "syn_trainer_init", "syn_trainer_init_conv", "syn_train_step", "syn_train_gradients_device",
    "syn_train_apply_device", "syn_train_gradients_enqueue", "syn_train_apply_enqueue", "syn_trainer_get_state", "syn_trainer_publish_weights", "syn_replay_deduplicate", "syn_train_set_data", "syn_train_epoch",
]


class SynthesisAmdError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"synthesis_amd error {code}: {message}")
        self.code = code


class CSearchResult(C.Structure):  # struct syn_search_result
    _fields_ = [
        ("child_N", C.c_float * 9), ("child_W", (C.c_float * 3) * 9), ("child_P", C.c_float * 9),
        ("child_sol", (C.c_int32 * 3) * 9),
        ("root_N", C.c_float), ("root_W", C.c_float * 3), ("root_sol", C.c_int32 * 3),
        ("num_nodes", C.c_uint32), ("best_action", C.c_int32),
        ("target_pi", C.c_float * 9), ("target_q", C.c_float * 3),
    ]


class CF16x2Plan(C.Structure):  # struct syn_f16x2_plan
    _fields_ = [("valid", C.c_int), ("activation_exp", C.c_int * 5), ("weight_exp", C.c_int * 5), ("out_exp", C.c_int),
                ("bound", C.c_double * 5)]


class CTrainConfig(C.Structure):  # struct syn_train_config
    _fields_ = [("weight_decay", C.c_float), ("policy_weight", C.c_float), ("value_weight", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]


class CCounters(C.Structure):  # struct syn_counters
    _fields_ = [(n, C.c_uint64) for n in (
        "explores", "select_levels", "children_scanned", "expansions", "new_nodes", "policy_evals", "backprop_levels",
        "solver_children", "solved_hits", "moves", "games", "max_depth")]


def library_path():
    """The in-tree build; SYNTHESIS_AMD_LIB points developer tools at another build of the same ABI (A/B timing)."""
    return os.environ.get("SYNTHESIS_AMD_LIB") or os.path.join(_HERE, "libsynthesis_amd.so")


_lib = None


def _share_hip_runtime_with_torch():
    """A process must hold ONE HIP runtime. PyTorch-ROCm wheels bundle their own libamdhip64.so.7; if this library pulled
    in /opt/rocm's copy first, a later `import torch` (device buffers, torch.distributed/RCCL) finds no GPU. When a torch
    with a bundled runtime is installed, bind to that copy (same soname, so our DT_NEEDED resolves to it); otherwise the
    RUNPATH copy under /opt/rocm is used. torch itself is not imported here."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """Loads libsynthesis_amd.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise SynthesisAmdError(-100, f"{path} not found: build it with `make -C synthesis_amd/csrc` "
                                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    lib.syn_last_error.restype = C.c_char_p
    lib.syn_last_error.argtypes = [C.c_void_p]
    lib.syn_engine_create.argtypes = [C.POINTER(CEngineConfig), C.c_int, C.POINTER(C.c_void_p)]
    lib.syn_engine_destroy.argtypes = [C.c_void_p]
    lib.syn_load_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.syn_load_weights_conv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.syn_set_network_arithmetic.argtypes = [C.c_void_p, C.c_int]
    lib.syn_get_network_arithmetic.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(CF16x2Plan)]
    lib.syn_f16x2_plan_of_blob.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(CF16x2Plan)]
    lib.syn_policy_eval_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.syn_policy_eval_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_int]
    lib.syn_eval_ctx_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.syn_eval_ctx_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.syn_eval_ctx_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.syn_eval_ctx_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.syn_eval_ctx_last_error.restype = C.c_char_p
    lib.syn_eval_ctx_last_error.argtypes = [C.c_void_p]
    lib.syn_eval_ctx_destroy.argtypes = [C.c_void_p]
    lib.syn_features_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.syn_linear_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_int]
    lib.syn_conv2d_forward.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                                      C.c_void_p, C.c_int]
    lib.syn_activation_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.syn_mcts_search.argtypes = [C.c_void_p, C.POINTER(CMctsConfig), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]
    lib.syn_mcts_search_lockstep.argtypes = [C.c_void_p, C.POINTER(CMctsConfig), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.syn_mcts_search_rollout.argtypes = [C.c_void_p, C.POINTER(CMctsConfig), C.c_uint64, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p]
    lib.syn_frozen_search_rollout.argtypes = [C.c_void_p, C.POINTER(CMctsConfig), C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.syn_selfplay_run.argtypes = [C.c_void_p, C.POINTER(CRolloutConfig), C.c_uint64, C.c_uint64, C.c_int] + \
                                    [C.c_void_p] * 8
    lib.syn_selfplay_run_lockstep.argtypes = [C.c_void_p, C.POINTER(CRolloutConfig), C.c_uint64, C.c_uint64, C.c_int, C.c_int] + \
                                             [C.c_void_p] * 8
    lib.syn_progress.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.syn_cancel.argtypes = [C.c_void_p]
    lib.syn_trainer_set_precision.argtypes = [C.c_void_p, C.c_int]
    lib.syn_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.syn_last_launch_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.syn_last_cache_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.syn_debug_stdrng_u32.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    lib.syn_debug_fast_div.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.syn_debug_small_int_math.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.syn_debug_math.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.syn_default_rollout_config.argtypes = [C.POINTER(CRolloutConfig)]
    lib.syn_debug_calibrate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.syn_trainer_init.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(CTrainConfig)]
    lib.syn_trainer_init_conv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(CTrainConfig)]
    lib.syn_train_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                   C.c_void_p]
    lib.syn_train_gradients_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p]
    lib.syn_train_apply_device.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float]
    lib.syn_train_gradients_enqueue.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_void_p]
    lib.syn_train_apply_enqueue.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]
    lib.syn_trainer_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_longlong),
                                          C.c_void_p]
    lib.syn_trainer_publish_weights.argtypes = [C.c_void_p]
    lib.syn_train_set_data.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.syn_train_epoch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_void_p]
    lib.syn_replay_deduplicate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_size_t)]
    _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f16x2_plan_of_blob(blob):
    """The f16x2 plan (power-of-two scales) syn_set_network_arithmetic would choose for a Connect4Net blob — host code only, no GPU.
    None when the blob has no plan (non-finite parameters)."""
    lib = load_library()
    blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
    pl = CF16x2Plan()
    rc = lib.syn_f16x2_plan_of_blob(_p(blob), blob.size, C.byref(pl))
    if rc != 0:
        raise SynthesisAmdError(rc, "syn_f16x2_plan_of_blob: bad arguments")
    if not pl.valid:
        return None
    return dict(activation_exp=list(pl.activation_exp), weight_exp=list(pl.weight_exp), out_exp=pl.out_exp, bound=list(pl.bound))


def shard_games(n_games, rank, world_size):
    """Static partition of game indices over ranks/GPUs (games share nothing: alpha_zero.rs:181-209). Rank r plays
    games r, r + world, r + 2*world, ... expressed as (first_game, count, stride=1) blocks: we use contiguous blocks so
    a rank's games are [first, first + count)."""
    base, rem = divmod(n_games, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


class EvalContext:
    """syn_eval_ctx: Policy::eval for a batch (policies/traits.rs:4-6) on its own stream — eval(), or submit() ... wait()."""

    def __init__(self, engine):
        self._engine = engine
        self._lib = engine._lib
        self._c = C.c_void_p()
        engine._check(self._lib.syn_eval_ctx_create(engine._h, C.byref(self._c)))
        self._n = None

    def _check(self, rc):
        if rc != 0:
            raise SynthesisAmdError(rc, (self._lib.syn_eval_ctx_last_error(self._c) or b"").decode())

    def submit(self, my_bb, op_bb):
        my = np.ascontiguousarray(my_bb, dtype=np.uint64).ravel()
        op = np.ascontiguousarray(op_bb, dtype=np.uint64).ravel()
        if my.size != op.size:
            raise ValueError("my_bb and op_bb must have the same length")
        self._check(self._lib.syn_eval_ctx_submit(self._c, _p(my), _p(op), int(my.size)))   # (a refused submit leaves a batch in flight as it was)
        self._n = int(my.size)

    def wait(self):
        """Results of the submitted batch. Raises when nothing was submitted (or the submission failed): there is nothing to wait for."""
        if self._n is None:
            raise SynthesisAmdError(-1, "EvalContext.wait(): no batch has been submitted")
        n, self._n = self._n, None
        logits = np.zeros((n, 9), np.float32)
        value = np.zeros((n, 3), np.float32)
        self._check(self._lib.syn_eval_ctx_wait(self._c, _p(logits), _p(value)))
        return logits, value

    def eval(self, my_bb, op_bb):
        self.submit(my_bb, op_bb)
        return self.wait()

    def close(self):
        if self._c is not None and self._c.value:
            self._lib.syn_eval_ctx_destroy(self._c)
            self._c = C.c_void_p()
            if self in getattr(self._engine, "_contexts", []):
                self._engine._contexts.remove(self)


class Engine:
    """One handle = one GPU + one stream (the reference's one policy per worker thread, alpha_zero.rs:192-198)."""

    def __init__(self, concurrent_games=4096, max_explores=800, device=0, policy_cache_log2=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        cfg = CEngineConfig(int(concurrent_games), int(max_explores), int(policy_cache_log2), 0)
        rc = self._lib.syn_engine_create(C.byref(cfg), int(device), C.byref(self._h))
        if rc != 0:
            msg = self._lib.syn_last_error(None)
            raise SynthesisAmdError(rc, msg.decode() if msg else "syn_engine_create failed")
        self.concurrent_games = int(concurrent_games)
        self.max_explores = int(max_explores)
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            for ctx in list(getattr(self, "_contexts", [])):
                ctx.close()
            self._lib.syn_engine_destroy(self._h)
            self._h = C.c_void_p()

    def eval_context(self):
        """One worker's policy object (alpha_zero.rs:192-198: every worker thread owns its policy): an evaluation context of this
        engine — own stream and staging, this engine's weights. Contexts may be used from different threads at the same time."""
        ctx = EvalContext(self)
        if not hasattr(self, "_contexts"):
            self._contexts = []
        self._contexts.append(ctx)
        return ctx

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.syn_last_error(self._h)
            raise SynthesisAmdError(rc, msg.decode() if msg else "")

    # ---- weights (vs.load, alpha_zero.rs:194)
    def load_weights(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
        self._check(self._lib.syn_load_weights(self._h, _p(blob), blob.size))

    def load_weights_conv(self, blob):
        """Connect4ConvNet (Conv2d<2,16,3,pad 1> + ReLU + Linear<1008,12>; include/synthesis_amd.h) becomes the engine's policy."""
        blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
        self._check(self._lib.syn_load_weights_conv(self._h, _p(blob), blob.size))

    # ---- the arithmetic Connect4Net is evaluated in (include/synthesis_amd.h: SYN_NET_ARITH_*)
    def set_network_arithmetic(self, arithmetic):
        """"f32" (default: v_mfma_f32_16x16x4_f32, the oracle's ACC_FMA) or "f16x2" (two-term f16 split on v_mfma_f32_16x16x32_f16, the
        oracle's ACC_F16X2); holds for policy_eval, evaluation contexts, mcts_search and selfplay until changed."""
        code = {"f32": 0, "f16x2": 1}.get(arithmetic, arithmetic)
        self._check(self._lib.syn_set_network_arithmetic(self._h, int(code)))

    def network_arithmetic(self):
        """(name, plan): plan = None or a dict of the f16x2 scales of the current Connect4Net."""
        a = C.c_int()
        pl = CF16x2Plan()
        self._check(self._lib.syn_get_network_arithmetic(self._h, C.byref(a), C.byref(pl)))
        plan = None
        if pl.valid:
            plan = dict(activation_exp=list(pl.activation_exp), weight_exp=list(pl.weight_exp), out_exp=pl.out_exp, bound=list(pl.bound))
        return ("f16x2" if a.value == 1 else "f32"), plan

    # ---- Policy::eval, batched (policies.rs:47-59)
    def policy_eval(self, my_bb, op_bb):
        my = np.ascontiguousarray(my_bb, dtype=np.uint64).ravel()
        op = np.ascontiguousarray(op_bb, dtype=np.uint64).ravel()
        if my.size != op.size:
            raise ValueError("my_bb and op_bb must have the same length")
        logits = np.zeros((my.size, 9), np.float32)
        value = np.zeros((my.size, 3), np.float32)
        self._check(self._lib.syn_policy_eval_batch(self._h, _p(my), _p(op), int(my.size), _p(logits), _p(value)))
        return logits, value

    def policy_eval_device(self, d_my, d_op, n, d_logits, d_value, sync=True):
        """Device-pointer form (ints = raw device addresses, e.g. torch tensor .data_ptr())."""
        self._check(self._lib.syn_policy_eval_batch_device(self._h, C.c_void_p(d_my), C.c_void_p(d_op), int(n),
                                                           C.c_void_p(d_logits), C.c_void_p(d_value), int(sync)))

    # ---- Game::features (connect4.rs:235-258)
    def features(self, my_bb, op_bb):
        my = np.ascontiguousarray(my_bb, dtype=np.uint64).ravel()
        op = np.ascontiguousarray(op_bb, dtype=np.uint64).ravel()
        out = np.zeros((my.size, 63), np.float32)
        self._check(self._lib.syn_features_batch(self._h, _p(my), _p(op), int(my.size), _p(out)))
        return out

    # ---- slimnn layers
    def linear(self, W, b, x, relu=False):
        W = np.ascontiguousarray(W, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        x = np.ascontiguousarray(x, np.float32).reshape(-1, W.shape[1])
        y = np.zeros((x.shape[0], W.shape[0]), np.float32)
        self._check(self._lib.syn_linear_forward(self._h, W.shape[1], W.shape[0], _p(W), _p(b), _p(x), x.shape[0],
                                                 _p(y), int(relu)))
        return y

    def activation(self, kind, x):
        """slimnn activation layer on x[batch][n]: kind 0 ReLU, 1 Tanh, 2 Softmax::apply_1d per row (activations.rs:31-63)."""
        x = np.ascontiguousarray(x, np.float32)
        x2 = x.reshape(1, -1) if x.ndim == 1 else x.reshape(x.shape[0], -1)
        y = np.zeros_like(x2)
        self._check(self._lib.syn_activation_forward(self._h, int(kind), _p(x2), int(x2.shape[0]), int(x2.shape[1]), _p(y)))
        return y.reshape(x.shape)

    def conv2d(self, W, b, x, row_pad=0, col_pad=0, stride=1, relu=False, out_hw=None):
        W = np.ascontiguousarray(W, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim == 3:
            x = x[None]
        cout, cin, k, _ = W.shape
        n, _, h_in, w_in = x.shape
        if out_hw is None:
            out_hw = ((h_in + 2 * row_pad - k) // stride + 1, (w_in + 2 * col_pad - k) // stride + 1)
        h_out, w_out = out_hw
        y = np.zeros((n, cout, max(h_out, 0), max(w_out, 0)), np.float32)
        self._check(self._lib.syn_conv2d_forward(self._h, cin, cout, k, row_pad, col_pad, stride, h_in, w_in, h_out,
                                                 w_out, _p(W), _p(b), _p(x), n, _p(y), int(relu)))
        return y

    # ---- MCTS::with_capacity + explore_n on n roots (mcts.rs:123-147)
    def mcts_search(self, cfg: MCTSConfig, my_bb, op_bb, explores, action_selection=1, rollout_seed=None):
        """rollout_seed=None: the network is the leaf policy. Otherwise MCTS over RolloutPolicy (policies/rollout.rs:8-31, the pairing of the reference's MCTS tests, mcts.rs:691-868): playout
        leaf evaluations, root i on StdRng::seed_from_u64(rollout_seed + i)."""
        my = np.ascontiguousarray(my_bb, dtype=np.uint64).ravel()
        op = np.ascontiguousarray(op_bb, dtype=np.uint64).ravel()
        n = int(my.size)
        res = (CSearchResult * max(n, 1))()
        c = cfg.to_c()
        if rollout_seed is None:
            rc = self._lib.syn_mcts_search(self._h, C.byref(c), _p(my), _p(op), n, int(explores),
                                           int(action_selection), C.cast(res, C.c_void_p))
        else:
            rc = self._lib.syn_mcts_search_rollout(self._h, C.byref(c), int(rollout_seed), _p(my), _p(op), n,
                                                   int(explores), int(action_selection), C.cast(res, C.c_void_p))
        if rc != -7:    # SYN_ERR_CANCELLED (Engine.cancel from another thread): searched roots are valid, the others all zero
            self._check(rc)
        out = self._search_records(res, n)
        if rc == -7:
            out["cancelled"] = True
        return out

    @staticmethod
    def _search_records(res, n):
        raw = np.frombuffer(res, dtype=np.uint8).reshape(max(n, 1), C.sizeof(CSearchResult))[:n]
        dt = np.dtype([("child_N", np.float32, (9,)), ("child_W", np.float32, (9, 3)), ("child_P", np.float32, (9,)),
                       ("child_sol", np.int32, (9, 3)), ("root_N", np.float32), ("root_W", np.float32, (3,)),
                       ("root_sol", np.int32, (3,)), ("num_nodes", np.uint32), ("best_action", np.int32),
                       ("target_pi", np.float32, (9,)), ("target_q", np.float32, (3,))])
        assert dt.itemsize == C.sizeof(CSearchResult)
        rec = raw.copy().view(dt).reshape(n)
        out = {k: rec[k].copy() for k in dt.names}
        out["root_stat"] = np.concatenate([out.pop("root_N")[:, None], out.pop("root_W")], axis=1)
        return out

    def mcts_search_lockstep(self, cfg: MCTSConfig, my_bb, op_bb, explores, action_selection=1, host_threads=0):
        """The same search with the trees on the host and only Policy::eval on the GPU (include/synthesis_amd_lockstep.hpp behind
        syn_mcts_search_lockstep): all roots advance in lock step, one syn_policy_eval_batch launch per round. Returns mcts_search's
        dict plus "stats" (rounds, positions_evaluated, seconds_total, seconds_policy)."""
        my = np.ascontiguousarray(my_bb, dtype=np.uint64).ravel()
        op = np.ascontiguousarray(op_bb, dtype=np.uint64).ravel()
        n = int(my.size)
        res = (CSearchResult * max(n, 1))()

        class CStats(C.Structure):
            _fields_ = [("rounds", C.c_uint64), ("positions_evaluated", C.c_uint64), ("seconds_total", C.c_double),
                        ("seconds_policy", C.c_double)]
        st = CStats()
        c = cfg.to_c()
        rc = self._lib.syn_mcts_search_lockstep(self._h, C.byref(c), _p(my), _p(op), n, int(explores), int(action_selection),
                                                int(host_threads), C.cast(res, C.c_void_p), C.cast(C.byref(st), C.c_void_p))
        self._check(rc)
        out = self._search_records(res, n)
        out["stats"] = {"rounds": int(st.rounds), "positions_evaluated": int(st.positions_evaluated),
                        "seconds_total": float(st.seconds_total), "seconds_policy": float(st.seconds_policy)}
        return out

    def selfplay_lockstep(self, cfg: RolloutConfig, base_seed, n_games, first_game=0, host_threads=0):
        """run_n_games with every game's trees on the host and only Policy::eval on the GPU (syn_selfplay_run_lockstep; BASELINE
        configs[1] as worded). Returns selfplay()'s arrays plus "stats"; the games equal selfplay()'s."""
        n = int(n_games)
        r = dict(plies=np.zeros(n, np.int32), states_bb=np.zeros((n, 63, 2), np.uint64), pis=np.zeros((n, 63, 9), np.float32),
                 vs=np.zeros((n, 63, 3), np.float32), actions=np.zeros((n, 63), np.uint8), root_nodes=np.zeros((n, 63), np.uint32),
                 final_kind=np.zeros(n, np.uint8))

        class CStats(C.Structure):
            _fields_ = [("rounds", C.c_uint64), ("positions_evaluated", C.c_uint64), ("seconds_total", C.c_double),
                        ("seconds_policy", C.c_double)]
        st = CStats()
        c = cfg.to_c()
        rc = self._lib.syn_selfplay_run_lockstep(
            self._h, C.byref(c), int(base_seed), int(first_game), n, int(host_threads), _p(r["plies"]),
            _p(r["states_bb"]), _p(r["pis"]), _p(r["vs"]), _p(r["actions"]), _p(r["root_nodes"]), _p(r["final_kind"]),
            C.cast(C.byref(st), C.c_void_p))
        self._check(rc)
        r["stats"] = {"rounds": int(st.rounds), "positions_evaluated": int(st.positions_evaluated),
                      "seconds_total": float(st.seconds_total), "seconds_policy": float(st.seconds_policy)}
        return r

    # ---- the evaluator's baseline: FrozenMCTS::exploit over RolloutPolicy on n roots (evaluator.rs:308-319)
    FROZEN_DTYPE = np.dtype([("child_N", np.float32, (9,)), ("child_cum", np.float32, (9,)), ("child_P", np.float32, (9,)),
                             ("child_sol", np.int32, (9, 3)), ("root_N", np.float32), ("root_cum", np.float32),
                             ("root_sol", np.int32, (3,)), ("num_nodes", np.uint32), ("best_action", np.int32)])

    def frozen_search(self, cfg: MCTSConfig, seeds, rng_words, my_bb, op_bb, explores, action_selection=1):
        """Root i plays out on StdRng::seed_from_u64(seeds[i]) from output word rng_words[i] on (one generator per match,
        evaluator.rs:171-172); returns the per-root records plus "rng_words", the position to hand to the match's next search.
        `explores` is a scalar or one value per root."""
        my = np.ascontiguousarray(my_bb, dtype=np.uint64).ravel()
        op = np.ascontiguousarray(op_bb, dtype=np.uint64).ravel()
        n = int(my.size)
        sd = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (n,)))
        words = np.array(np.broadcast_to(np.asarray(rng_words, dtype=np.uint64), (n,)), dtype=np.uint64)
        ex = np.ascontiguousarray(np.broadcast_to(np.asarray(explores, dtype=np.int32), (n,)))
        rec = np.zeros(max(n, 1), self.FROZEN_DTYPE)
        c = cfg.to_c()
        self._check(self._lib.syn_frozen_search_rollout(self._h, C.byref(c), _p(sd), _p(words), _p(my), _p(op), _p(ex), n,
                                                        int(action_selection), _p(rec)))
        out = {k: rec[k][:n].copy() for k in self.FROZEN_DTYPE.names}
        out["root_stat"] = np.stack([out.pop("root_N"), out.pop("root_cum")], axis=1)
        out["rng_words"] = words
        return out

    # ---- run_n_games (alpha_zero.rs:181-209)
    def selfplay(self, cfg: RolloutConfig, base_seed, n_games, first_game=0, outputs=True, counters=False):
        n = int(n_games)
        r = dict(plies=np.zeros(n, np.int32))
        if outputs:
            r.update(states_bb=np.zeros((n, 63, 2), np.uint64), pis=np.zeros((n, 63, 9), np.float32),
                     vs=np.zeros((n, 63, 3), np.float32), actions=np.zeros((n, 63), np.uint8),
                     root_nodes=np.zeros((n, 63), np.uint32), final_kind=np.zeros(n, np.uint8))
        ctr = CCounters() if counters else None
        c = cfg.to_c()
        rc = self._lib.syn_selfplay_run(
            self._h, C.byref(c), int(base_seed), int(first_game), n, _p(r["plies"]), _p(r.get("states_bb")),
            _p(r.get("pis")), _p(r.get("vs")), _p(r.get("actions")), _p(r.get("root_nodes")), _p(r.get("final_kind")),
            C.cast(C.byref(ctr), C.c_void_p) if ctr is not None else None)
        if rc == -7:    # SYN_ERR_CANCELLED (Engine.cancel from another thread): the games that finished are valid, plies == 0 otherwise
            r["cancelled"] = True
        else:
            self._check(rc)
        if ctr is not None:
            r["counters"] = {name: int(getattr(ctr, name)) for name, _ in CCounters._fields_}
        r["kernel_ms"] = self.last_kernel_ms()
        return r

    def progress(self):
        """(jobs started, games finished) of the call running on this engine — callable from another thread while it runs."""
        a, b = C.c_int(), C.c_int()
        self._check(self._lib.syn_progress(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def cancel(self):
        """No further game (or root) starts; the running selfplay() / mcts_search() returns once the started ones have finished,
        with result["cancelled"] = True if anything was left out (plies == 0 / num_nodes == 0 mark those). Raises
        SynthesisAmdError(SYN_ERR_INVALID_ARGUMENT) when no call is in flight on this engine."""
        self._check(self._lib.syn_cancel(self._h))

    def last_cache_stats(self):
        """(hits, misses) of the device PolicyWithCache during the last search / self-play call."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.syn_last_cache_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def last_launch_shape(self):
        """(shape, workgroups, threads per workgroup) of the last search / self-play launch; shape 4 = lane-per-tree."""
        s, g, t = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.syn_last_launch_shape(self._h, C.byref(s), C.byref(g), C.byref(t)))
        return s.value, g.value, t.value

    def last_kernel_ms(self):
        ms = C.c_float()
        nl = C.c_int()
        self._check(self._lib.syn_last_timing(self._h, C.byref(ms), C.byref(nl)))
        return float(ms.value)

    # ---- learner step (alpha_zero.rs:31-36,72-94) and replay de-duplication (data.rs:196-235)
    def trainer_init(self, blob, weight_decay=1e-6, policy_weight=1.0, value_weight=1.0, beta1=0.9, beta2=0.999,
                     eps=1e-8):
        blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
        cfg = CTrainConfig(weight_decay, policy_weight, value_weight, beta1, beta2, eps)
        self._check(self._lib.syn_trainer_init(self._h, _p(blob), blob.size, C.byref(cfg)))
        self._trainer_params = NUM_PARAMS

    def trainer_init_conv(self, blob, weight_decay=1e-6, policy_weight=1.0, value_weight=1.0, beta1=0.9, beta2=0.999, eps=1e-8):
        """The learner for Connect4ConvNet (load_weights_conv's network and blob order): train_step / train_epoch /
        trainer_state / trainer_publish_weights then work on that network (minibatches of at most 32 positions)."""
        blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
        cfg = CTrainConfig(weight_decay, policy_weight, value_weight, beta1, beta2, eps)
        self._check(self._lib.syn_trainer_init_conv(self._h, _p(blob), blob.size, C.byref(cfg)))
        self._trainer_params = CONV_NUM_PARAMS

    def trainer_set_precision(self, precision):
        """"f32" (default; bit-identical to the oracle) | "bf16" (Connect4ConvNet learner only: bf16 matrix cores, f32 accumulation,
        f32 master weights and Adam — BASELINE configs[4]'s "bf16 conv")."""
        self._check(self._lib.syn_trainer_set_precision(self._h, {"f32": 0, "bf16": 1}[precision]))

    def train_step(self, my_bb, op_bb, target_pi, target_v, lr):
        my = np.ascontiguousarray(my_bb, dtype=np.uint64).ravel()
        op = np.ascontiguousarray(op_bb, dtype=np.uint64).ravel()
        tpi = np.ascontiguousarray(target_pi, dtype=np.float32).reshape(my.size, 9)
        tv = np.ascontiguousarray(target_v, dtype=np.float32).reshape(my.size, 3)
        losses = np.zeros(2, np.float32)
        self._check(self._lib.syn_train_step(self._h, _p(my), _p(op), _p(tpi), _p(tv), int(my.size), float(lr),
                                             _p(losses)))
        return losses

    def train_gradients_device(self, d_my, d_op, d_tpi, d_tv, batch, d_grads):
        losses = np.zeros(2, np.float32)
        self._check(self._lib.syn_train_gradients_device(self._h, C.c_void_p(d_my), C.c_void_p(d_op),
                                                         C.c_void_p(d_tpi), C.c_void_p(d_tv), int(batch),
                                                         C.c_void_p(d_grads), _p(losses)))
        return losses

    def train_apply_device(self, d_grads, lr, grad_scale=1.0):
        self._check(self._lib.syn_train_apply_device(self._h, C.c_void_p(d_grads), float(lr), float(grad_scale)))

    def train_gradients_enqueue(self, stream, d_my, d_op, d_tpi, d_tv, batch, d_grads, d_losses=0):
        """syn_train_gradients_enqueue: the gradient kernel on `stream` (a raw hipStream_t, e.g. torch.cuda.current_stream().cuda_stream),
        loss sums to the device words at d_losses; nothing is synchronised."""
        self._check(self._lib.syn_train_gradients_enqueue(self._h, C.c_void_p(stream), C.c_void_p(d_my), C.c_void_p(d_op), C.c_void_p(d_tpi),
                                                          C.c_void_p(d_tv), int(batch), C.c_void_p(d_grads), C.c_void_p(d_losses)))

    def train_apply_enqueue(self, stream, d_grads, lr, grad_scale=1.0):
        self._check(self._lib.syn_train_apply_enqueue(self._h, C.c_void_p(stream), C.c_void_p(d_grads), float(lr), float(grad_scale)))

    def trainer_state(self):
        blob = np.zeros(getattr(self, "_trainer_params", NUM_PARAMS), np.float32)
        m = np.zeros_like(blob); v = np.zeros_like(blob); g = np.zeros_like(blob)
        step = C.c_longlong()
        self._check(self._lib.syn_trainer_get_state(self._h, _p(blob), _p(m), _p(v), C.byref(step), _p(g)))
        return dict(weights=blob, m=m, v=v, step=int(step.value), grads=g)

    def train_set_data(self, my_bb, op_bb, target_pi, target_v):
        """Uploads the de-duplicated buffer once; train_epoch then indexes it on the device."""
        my = np.ascontiguousarray(my_bb, dtype=np.uint64).ravel()
        op = np.ascontiguousarray(op_bb, dtype=np.uint64).ravel()
        tpi = np.ascontiguousarray(target_pi, dtype=np.float32).reshape(my.size, 9)
        tv = np.ascontiguousarray(target_v, dtype=np.float32).reshape(my.size, 3)
        self._check(self._lib.syn_train_set_data(self._h, _p(my), _p(op), _p(tpi), _p(tv), int(my.size)))

    def train_epoch(self, perm, batch_size, lr):
        """len(perm) // batch_size optimiser steps in one call (drop_last = true); returns the per-step losses [steps][2]."""
        perm = np.ascontiguousarray(perm, dtype=np.int32).ravel()
        steps = perm.size // int(batch_size)
        losses = np.zeros((steps, 2), np.float32)
        self._check(self._lib.syn_train_epoch(self._h, _p(perm), steps, int(batch_size), float(lr), _p(losses)))
        return losses

    def trainer_publish_weights(self):
        self._check(self._lib.syn_trainer_publish_weights(self._h))

    def replay_deduplicate(self, my_bb, op_bb, pis, vs):
        my = np.ascontiguousarray(my_bb, dtype=np.uint64).ravel()
        op = np.ascontiguousarray(op_bb, dtype=np.uint64).ravel()
        n = int(my.size)
        pis = np.ascontiguousarray(pis, dtype=np.float32).reshape(n, 9)
        vs = np.ascontiguousarray(vs, dtype=np.float32).reshape(n, 3)
        o = dict(my_bb=np.zeros(n, np.uint64), op_bb=np.zeros(n, np.uint64), pis=np.zeros((n, 9), np.float32),
                 vs=np.zeros((n, 3), np.float32), num=np.zeros(n, np.uint32))
        cnt = C.c_size_t()
        self._check(self._lib.syn_replay_deduplicate(self._h, _p(my), _p(op), _p(pis), _p(vs), n, _p(o["my_bb"]),
                                                     _p(o["op_bb"]), _p(o["pis"]), _p(o["vs"]), _p(o["num"]),
                                                     C.byref(cnt)))
        return {k: a[: cnt.value] for k, a in o.items()}

    # ---- parity probes
    def debug_stdrng_u32(self, seed, n):
        out = np.zeros(int(n), np.uint32)
        self._check(self._lib.syn_debug_stdrng_u32(self._h, int(seed), int(n), _p(out)))
        return out

    def debug_fast_div(self, a, b):
        a = np.ascontiguousarray(a, np.float32).ravel()
        b = np.ascontiguousarray(b, np.float32).ravel()
        f = np.zeros_like(a); d = np.zeros_like(a)
        self._check(self._lib.syn_debug_fast_div(self._h, _p(a), _p(b), int(a.size), _p(f), _p(d)))
        return f, d

    def debug_small_int_math(self, b_lo, b_hi):
        """(mismatches of div2_by_small_int, of div2_safe_range, of sqrt_normal_range) over every significand: see synthesis_amd.h"""
        out = np.zeros(3, np.uint64)
        self._check(self._lib.syn_debug_small_int_math(self._h, int(b_lo), int(b_hi), _p(out)))
        return tuple(int(x) for x in out)

    def debug_math(self, a, b):
        a = np.ascontiguousarray(a, np.float32).ravel()
        b = np.ascontiguousarray(b, np.float32).ravel()
        e = np.zeros_like(a); d = np.zeros_like(a); s = np.zeros_like(a)
        self._check(self._lib.syn_debug_math(self._h, _p(a), _p(b), int(a.size), _p(e), _p(d), _p(s)))
        return e, d, s
