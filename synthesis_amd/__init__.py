"""synthesis_amd — MI355X-native batched self-play engine (9x7 Connect4 AlphaZero hot path of coreylowman/synthesis).

Host-side mirror of the reference's plug-in surface for this path:

  reference (Rust)                                   here
  ------------------------------------------------   -----------------------------------------------
  synthesis::config::{MCTSConfig, RolloutConfig,..}  synthesis_amd.config.{MCTSConfig, RolloutConfig, ...}
  trait Policy<G,N>::eval  (policies/traits.rs:4-6)   Engine.policy_eval(my_bb, op_bb)   (batched)
  trait Game<N>::features  (game.rs:86)               Engine.features(my_bb, op_bb)
  MCTS::with_capacity + explore_n (mcts.rs:123-147)   Engine.mcts_search(cfg, my_bb, op_bb, explores)
  run_n_games              (alpha_zero.rs:181-209)    Engine.selfplay(cfg, seed, n_games)
  slimnn::{Linear, Conv2d} (slimnn/src)               Engine.linear(...), Engine.conv2d(...)

Everything computes on the GPU through the C ABI in include/synthesis_amd.h (libsynthesis_amd.so, hand-written HIP for
gfx950). There is no CPU fallback: importing works anywhere, creating an Engine without the library or without an
MI355X raises.
"""
from .config import (ActionSelection, Exploration, Fpu, MCTSConfig, PolicyNoise, RolloutConfig, ValueTarget,
                     parity_mcts_config, parity_rollout_config, reference_selfplay_mcts_config)
from .engine import Engine, SynthesisAmdError, library_path, load_library, shard_games

__all__ = [
    "ActionSelection", "Exploration", "Fpu", "MCTSConfig", "PolicyNoise", "RolloutConfig", "ValueTarget",
    "parity_mcts_config", "parity_rollout_config", "reference_selfplay_mcts_config", "Engine", "SynthesisAmdError", "library_path", "load_library",
    "shard_games",
]
