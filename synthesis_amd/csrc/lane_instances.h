// synthesis_amd — which instantiations of selfplay_kernel_lanes (lane_kernel.cuh) for Connect4Net (POLICY 0) and RolloutPolicy
// (POLICY 1) the library ships, as X-macro lists: engine.hip declares them `extern template`, engine_lanes_fast.hip (the compile-
// time-folded parity configuration family) and engine_lanes_gen.hip (the runtime-switched one) define them. Separate translation
// units only so that `make -j` builds them beside engine.hip.
//   X(MODE, COUNT, FAST, NW, PROF, POLICY)      FAST: 0 runtime-switched, 1 parity family, 2 the reference's self-play configuration
//                                                (mcts.cuh CfgView; engine_lanes_ref.hip holds the third list)
#pragma once
#define SYN_LANES_FAST_LIST(X)                                                                                   \
    X(MODE_SELFPLAY, false, true, 4, false, 0) X(MODE_SELFPLAY, false, true, 8, false, 0)                        \
    X(MODE_SELFPLAY, false, true, 12, false, 0) X(MODE_SELFPLAY, false, true, 16, false, 0)                      \
    X(MODE_SELFPLAY, false, true, 4, true, 0) X(MODE_SELFPLAY, false, true, 8, true, 0)                          \
    X(MODE_SELFPLAY, false, true, 12, true, 0) X(MODE_SELFPLAY, false, true, 16, true, 0)                        \
    X(MODE_SELFPLAY, true, true, 4, false, 0) X(MODE_SELFPLAY, true, true, 8, false, 0)                          \
    X(MODE_SELFPLAY, true, true, 12, false, 0) X(MODE_SELFPLAY, true, true, 16, false, 0)                        \
    X(MODE_SEARCH, false, true, 4, false, 0) X(MODE_SEARCH, false, true, 8, false, 0)                            \
    X(MODE_SEARCH, false, true, 12, false, 0) X(MODE_SEARCH, false, true, 16, false, 0)
#define SYN_LANES_GEN_LIST(X)                                                                                    \
    X(MODE_SELFPLAY, false, false, 4, false, 0) X(MODE_SELFPLAY, false, false, 8, false, 0)                      \
    X(MODE_SELFPLAY, false, false, 12, false, 0) X(MODE_SELFPLAY, false, false, 16, false, 0)                    \
    X(MODE_SELFPLAY, false, false, 4, true, 0) X(MODE_SELFPLAY, false, false, 8, true, 0)                        \
    X(MODE_SELFPLAY, false, false, 12, true, 0) X(MODE_SELFPLAY, false, false, 16, true, 0)                      \
    X(MODE_SELFPLAY, true, false, 4, false, 0) X(MODE_SELFPLAY, true, false, 8, false, 0)                        \
    X(MODE_SELFPLAY, true, false, 12, false, 0) X(MODE_SELFPLAY, true, false, 16, false, 0)                      \
    X(MODE_SEARCH, false, false, 4, false, 0) X(MODE_SEARCH, false, false, 8, false, 0)                          \
    X(MODE_SEARCH, false, false, 12, false, 0) X(MODE_SEARCH, false, false, 16, false, 0)                        \
    X(MODE_SEARCH, false, false, 8, false, 1)
#define SYN_LANES_REF_LIST(X)                                                                                    \
    X(MODE_SELFPLAY, false, 2, 8, false, 0) X(MODE_SELFPLAY, false, 2, 16, false, 0)                             \
    X(MODE_SELFPLAY, true, 2, 8, false, 0) X(MODE_SELFPLAY, true, 2, 16, false, 0)                               \
    X(MODE_SEARCH, false, 2, 8, false, 0) X(MODE_SEARCH, false, 2, 16, false, 0)                                 \
    X(MODE_SELFPLAY, false, 2, 8, true, 0) X(MODE_SELFPLAY, false, 2, 16, true, 0)                               \
    X(MODE_SELFPLAY, false, 2, 12, false, 0) X(MODE_SELFPLAY, true, 2, 12, false, 0) X(MODE_SEARCH, false, 2, 12, false, 0) \
    X(MODE_SELFPLAY, false, 2, 12, true, 0)
// Connect4Net in the f16x2 arithmetic (POLICY 3, f16x2_tile.cuh): every configuration family at every wave count
// (engine_lanes_f16.hip)
#define SYN_LANES_F16_LIST(X)                                                                                    \
    X(MODE_SELFPLAY, false, true, 4, false, 3) X(MODE_SELFPLAY, false, true, 8, false, 3)                        \
    X(MODE_SELFPLAY, false, true, 12, false, 3) X(MODE_SELFPLAY, false, true, 16, false, 3)                      \
    X(MODE_SELFPLAY, true, true, 4, false, 3) X(MODE_SELFPLAY, true, true, 8, false, 3)                          \
    X(MODE_SELFPLAY, true, true, 12, false, 3) X(MODE_SELFPLAY, true, true, 16, false, 3)                        \
    X(MODE_SEARCH, false, true, 4, false, 3) X(MODE_SEARCH, false, true, 8, false, 3)                            \
    X(MODE_SEARCH, false, true, 12, false, 3) X(MODE_SEARCH, false, true, 16, false, 3)                          \
    X(MODE_SELFPLAY, false, true, 12, true, 3) X(MODE_SELFPLAY, false, true, 16, true, 3)                        \
    X(MODE_SELFPLAY, false, 2, 4, false, 3) X(MODE_SELFPLAY, false, 2, 8, false, 3)                              \
    X(MODE_SELFPLAY, false, 2, 12, false, 3) X(MODE_SELFPLAY, false, 2, 16, false, 3)                            \
    X(MODE_SELFPLAY, true, 2, 4, false, 3) X(MODE_SELFPLAY, true, 2, 8, false, 3)                                \
    X(MODE_SELFPLAY, true, 2, 12, false, 3) X(MODE_SELFPLAY, true, 2, 16, false, 3)                              \
    X(MODE_SEARCH, false, 2, 4, false, 3) X(MODE_SEARCH, false, 2, 8, false, 3)                                  \
    X(MODE_SEARCH, false, 2, 12, false, 3) X(MODE_SEARCH, false, 2, 16, false, 3)
#define SYN_LANES_F16_GEN_LIST(X)                                                                                \
    X(MODE_SELFPLAY, false, false, 4, false, 3) X(MODE_SELFPLAY, false, false, 8, false, 3)                      \
    X(MODE_SELFPLAY, true, false, 4, false, 3) X(MODE_SELFPLAY, true, false, 8, false, 3)                        \
    X(MODE_SEARCH, false, false, 4, false, 3) X(MODE_SEARCH, false, false, 8, false, 3)
// The pool kernel (pool_kernel.cuh: trees unbound from the lanes for the descent), Connect4Net f32 (POLICY 0: engine_pool.hip) and f16x2
// (POLICY 3: engine_pool_f16.hip), the two compile-time-folded families, 12 waves:  X(MODE, COUNT, FAST, NW, POLICY)
#define SYN_POOL_F32_LIST(X)                                                                                     \
    X(MODE_SELFPLAY, false, 1, 12, 0) X(MODE_SELFPLAY, true, 1, 12, 0) X(MODE_SEARCH, false, 1, 12, 0)           \
    X(MODE_SELFPLAY, false, 2, 12, 0) X(MODE_SELFPLAY, true, 2, 12, 0) X(MODE_SEARCH, false, 2, 12, 0)           \
    X(MODE_SELFPLAY, false, 1, 8, 0) X(MODE_SELFPLAY, true, 1, 8, 0) X(MODE_SEARCH, false, 1, 8, 0)              \
    X(MODE_SELFPLAY, false, 2, 8, 0) X(MODE_SELFPLAY, true, 2, 8, 0) X(MODE_SEARCH, false, 2, 8, 0)
#define SYN_POOL_F16_LIST(X)                                                                                     \
    X(MODE_SELFPLAY, false, 1, 12, 3) X(MODE_SELFPLAY, true, 1, 12, 3) X(MODE_SEARCH, false, 1, 12, 3)           \
    X(MODE_SELFPLAY, false, 2, 12, 3) X(MODE_SELFPLAY, true, 2, 12, 3) X(MODE_SEARCH, false, 2, 12, 3)           \
    X(MODE_SELFPLAY, false, 1, 8, 3) X(MODE_SELFPLAY, true, 1, 8, 3) X(MODE_SEARCH, false, 1, 8, 3)              \
    X(MODE_SELFPLAY, false, 2, 8, 3) X(MODE_SELFPLAY, true, 2, 8, 3) X(MODE_SEARCH, false, 2, 8, 3)
