"""Plain-data mirrors of synthesis/src/config.rs (same names, same fields, same meaning).

`Fpu.Func` (config.rs:25) is a Rust closure and cannot cross the C ABI; the closure the reference itself configures,
`Normal::new(1.0, 0.1).sample(&mut thread_rng())` (study-connect4/src/main.rs:43-47), is `Fpu.Func` with `fpu_value` =
mean and `fpu_std` = standard deviation, sampled on the device from a per-tree StdRng stream (reproducible, unlike
thread_rng). `PolicyNoise.Dirichlet` samples rand_distr's Dirichlet on the device the same way.
"""
import ctypes as C
import enum
from dataclasses import dataclass, field


class Exploration(enum.IntEnum):          # config.rs:9-13
    Uct = 0
    PolynomialUct = 1


class ActionSelection(enum.IntEnum):      # config.rs:15-19
    Q = 0
    NumVisits = 1


class Fpu(enum.IntEnum):                  # config.rs:21-26
    Const = 0
    ParentQ = 1
    Func = 2        # SYN_FPU_NORMAL: Func(|| Normal(fpu_value, fpu_std)), the closure study-connect4/src/main.rs:43-47 configures
    FuncPtr = 3     # SYN_FPU_FUNC: Func(fpu_fn), any fn() -> f32 (config.rs:25) — called by the host trees only


class PolicyNoise(enum.IntEnum):          # config.rs:39-44
    NoNoise = 0
    Equal = 1
    Dirichlet = 2


class ValueTarget(enum.IntEnum):          # config.rs:1-7
    Z = 0
    Q = 1
    QZaverage = 2
    QtoZ = 3


FPU_FN = C.CFUNCTYPE(C.c_float)           # float (*fpu_fn)(void): Fpu::Func's fn() -> f32 (config.rs:25)


class CMctsConfig(C.Structure):           # struct syn_mcts_config
    _fields_ = [
        ("exploration", C.c_int32), ("c", C.c_float),
        ("solve", C.c_int32), ("correct_values_on_solve", C.c_int32), ("select_solved_nodes", C.c_int32),
        ("auto_extend", C.c_int32),
        ("fpu", C.c_int32), ("fpu_value", C.c_float),
        ("root_policy_noise", C.c_int32), ("noise_alpha", C.c_float), ("noise_weight", C.c_float),
        ("fpu_std", C.c_float),
        ("fpu_fn", FPU_FN),
    ]


class CRolloutConfig(C.Structure):        # struct syn_rollout_config
    _fields_ = [
        ("num_explores", C.c_int32), ("random_actions_until", C.c_int32), ("sample_actions_until", C.c_int32),
        ("stop_games_when_solved", C.c_int32),
        ("value_target", C.c_int32), ("value_target_p", C.c_float), ("value_target_from", C.c_float),
        ("value_target_to", C.c_float),
        ("action", C.c_int32),
        ("mcts_cfg", CMctsConfig),
    ]


class CEngineConfig(C.Structure):         # struct syn_engine_config
    _fields_ = [("concurrent_games", C.c_int32), ("max_explores", C.c_int32), ("policy_cache_log2", C.c_int32),
                ("reserved1", C.c_int32)]


@dataclass
class MCTSConfig:                         # config.rs:28-37
    exploration: Exploration = Exploration.PolynomialUct
    c: float = 3.0
    solve: bool = True
    correct_values_on_solve: bool = True
    select_solved_nodes: bool = True
    auto_extend: bool = True
    fpu: Fpu = Fpu.Const
    fpu_value: float = 1.0
    root_policy_noise: PolicyNoise = PolicyNoise.NoNoise
    noise_alpha: float = 0.0
    noise_weight: float = 0.0
    fpu_std: float = 0.0              # Fpu.Func = Normal(fpu_value, fpu_std)
    fpu_fn: object = None             # Fpu.FuncPtr: a Python callable () -> float or a ctypes FPU_FN (kept alive by this object)

    def to_c(self) -> CMctsConfig:
        fn = self.fpu_fn
        if fn is not None and not isinstance(fn, FPU_FN):
            fn = FPU_FN(fn)
            self._fpu_fn_c = fn       # the callback object must outlive every call that holds its address
        return CMctsConfig(int(self.exploration), float(self.c), int(self.solve), int(self.correct_values_on_solve),
                           int(self.select_solved_nodes), int(self.auto_extend), int(self.fpu), float(self.fpu_value),
                           int(self.root_policy_noise), float(self.noise_alpha), float(self.noise_weight), float(self.fpu_std),
                           fn if fn is not None else FPU_FN())


@dataclass
class RolloutConfig:                      # config.rs:46-56 (num_workers -> Engine(concurrent_games=...))
    num_explores: int = 800
    random_actions_until: int = 1
    sample_actions_until: int = 30
    stop_games_when_solved: bool = False
    value_target: ValueTarget = ValueTarget.Q
    value_target_p: float = 0.0       # QZaverage { p }
    value_target_from: float = 0.0    # QtoZ { from, to }
    value_target_to: float = 0.0
    action: ActionSelection = ActionSelection.NumVisits
    mcts_cfg: MCTSConfig = field(default_factory=MCTSConfig)

    def to_c(self) -> CRolloutConfig:
        return CRolloutConfig(int(self.num_explores), int(self.random_actions_until), int(self.sample_actions_until),
                              int(self.stop_games_when_solved), int(self.value_target), float(self.value_target_p),
                              float(self.value_target_from), float(self.value_target_to), int(self.action),
                              self.mcts_cfg.to_c())


def parity_mcts_config(**kw) -> MCTSConfig:
    """policy_mcts_cfg of study-connect4/src/main.rs:58-66: the deterministic variant (Fpu::Const(1.0)) of the
    reference's self-play MCTS configuration (whose Fpu::Func samples N(1, 0.1) from thread_rng, main.rs:43-47)."""
    return MCTSConfig(**kw)


def reference_selfplay_mcts_config(**kw) -> MCTSConfig:
    """mcts_cfg of study-connect4/src/main.rs:37-49 as written: Fpu::Func(|| Normal(1.0, 0.1)) and no root noise."""
    d = dict(fpu=Fpu.Func, fpu_value=1.0, fpu_std=0.1)
    d.update(kw)
    return MCTSConfig(**d)


def parity_rollout_config(num_explores: int = 800, **kw) -> RolloutConfig:
    """rollout_cfg of study-connect4/src/main.rs:28-36 with parity_mcts_config; explores = 800 per BASELINE.json
    (the reference default is 1600, main.rs:30)."""
    return RolloutConfig(num_explores=num_explores, **kw)
