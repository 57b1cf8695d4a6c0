// ORACLE — TEST INFRASTRUCTURE ONLY.
// AddressSanitizer / UndefinedBehaviorSanitizer run of the CPU restatement (GPU sanitizers are not available on the pool;
// SURVEY.md §5 asks for sanitizers on the CPU build): drives the same C API the tests use — the TicTacToe solver KATs,
// Connect4 searches (network and rollout policy), the evaluator's baseline tree and match loops, multi-threaded self-play with PolicyWithCache, training steps and
// de-duplication — and exits non-zero on any sanitizer report or failed sanity check.
//   make -C oracle sanitize   (g++ -fsanitize=address,undefined, runs the binary)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "oracle_capi.cpp"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main() {
    // deterministic pseudo-random Connect4Net blob
    std::vector<float> blob(30492);
    uint32_t s = 12345;
    for (auto& w : blob) { s = s * 1664525u + 1013904223u; w = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.25f; }

    {   // TicTacToe solver KATs (mcts.rs:691-831)
        int some[9], kind[9], best, rs, rk; unsigned turns[9], nodes, rt; float pol[9], q[3];
        orc_ttt_kat(1, 0, 12, 100000, some, kind, turns, &best, &nodes, &rs, &rk, &rt, pol, q);
        CHECK(nodes == 69 && rs == 1);
        orc_ttt_kat(0, 0, 12, 100000, some, kind, turns, &best, &nodes, &rs, &rk, &rt, pol, q);
        CHECK(best == 6);
        orc_ttt_kat(2, 0, 12, 100000, some, kind, turns, &best, &nodes, &rs, &rk, &rt, pol, q);
        CHECK(best == 1);
    }
    orc_mcts_config mc{1, 3.0f, 1, 1, 1, 1, 0, 1.0f, 0, 0.0f, 0.0f};
    {   // Connect4 searches from a few positions, network and rollout leaf policies
        const int n = 6;
        std::vector<uint64_t> my(n, 0), op(n, 0);
        my[1] = 1ull << 21; op[2] = (1ull << 28) | (1ull << 29); my[2] = 1ull << 35;
        std::vector<float> cN(n * 9), cW(n * 27), cP(n * 9), rst(n * 4), tpi(n * 9), tq(n * 3);
        std::vector<int> csol(n * 27), rsol(n * 3), best(n);
        std::vector<unsigned> nn(n);
        orc_c4_mcts_search(&mc, blob.data(), 1, my.data(), op.data(), n, 300, 1, cN.data(), cW.data(), cP.data(), csol.data(),
                           rst.data(), rsol.data(), nn.data(), best.data(), tpi.data(), tq.data());
        CHECK(rst[0] == 301.0f && nn[0] <= 1 + 9 * 301);
        orc_mcts_config uct{0, 2.0f, 1, 1, 1, 0, 0, 1e30f, 0, 0.0f, 0.0f};
        orc_c4_mcts_search_rollout(&uct, 7, my.data(), op.data(), n, 300, 1, cN.data(), cW.data(), cP.data(), csol.data(),
                                   rst.data(), rsol.data(), nn.data(), best.data(), tpi.data(), tq.data());
        CHECK(best[0] >= 0 && best[0] < 9);
        // the evaluator's baseline tree and its two match loops
        std::vector<uint64_t> seeds(n, 5), words(n, 3);
        std::vector<int> ex(n, 250);
        std::vector<float> fN(n * 9), fC(n * 9), fP(n * 9), frs(n * 2);
        std::vector<int> fsol(n * 27), frsol(n * 3);
        orc_c4_frozen_search(&uct, 1, blob.data(), 1, seeds.data(), words.data(), my.data(), op.data(), n, ex.data(), 1, fN.data(),
                             fC.data(), fP.data(), fsol.data(), frs.data(), frsol.data(), nn.data(), best.data());
        CHECK(frs[0] == 251.0f && words[0] > 3 && best[0] >= 0 && best[0] < 9);
        uint8_t moves[63];
        uint64_t after[63];
        int nm = 0;
        float r = orc_c4_mcts_vs_mcts(&uct, 1, 0, 120, 60, 9, moves, &nm, after);
        CHECK(nm >= 7 && nm <= 63 && (r == 1.0f || r == 0.0f || r == -1.0f));
        r = orc_c4_eval_against_rollout(&mc, 60, 1, blob.data(), 1, &uct, 1, 1, 80, 4, moves, &nm, after);
        CHECK(nm >= 7 && nm <= 63 && (r == 1.0f || r == 0.0f || r == -1.0f));
    }
    {   // self-play: 12 games on 4 threads with PolicyWithCache, all outputs
        orc_rollout_config rc{120, 1, 30, 0, 1, 0.0f, 0.0f, 0.0f, 1, mc};
        const int n = 12;
        std::vector<int> plies(n);
        std::vector<uint64_t> st((size_t)n * 63 * 2), ctr(12);
        std::vector<float> pis((size_t)n * 63 * 9), vs((size_t)n * 63 * 3);
        std::vector<uint8_t> act((size_t)n * 63), fin(n);
        std::vector<uint32_t> rn((size_t)n * 63);
        orc_c4_selfplay(&rc, blob.data(), 1, 3, 0, n, 4, 1, plies.data(), st.data(), pis.data(), vs.data(), act.data(), rn.data(),
                        fin.data(), ctr.data());
        for (int g = 0; g < n; g++) CHECK(plies[g] >= 7 && plies[g] <= 63);
        // de-duplicate the recorded positions and train two steps on the first 32 unique ones
        std::vector<uint64_t> my, op; std::vector<float> pi, v;
        for (int g = 0; g < n; g++)
            for (int k = 0; k < plies[g]; k++) {
                size_t p = (size_t)g * 63 + k;
                my.push_back(st[p * 2]); op.push_back(st[p * 2 + 1]);
                pi.insert(pi.end(), &pis[p * 9], &pis[p * 9] + 9);
                v.insert(v.end(), &vs[p * 3], &vs[p * 3] + 3);
            }
        const size_t m = my.size();
        std::vector<uint64_t> umy(m), uop(m); std::vector<float> upi(m * 9), uv(m * 3); std::vector<uint32_t> num(m);
        size_t u = orc_dedup(my.data(), op.data(), pi.data(), v.data(), m, umy.data(), uop.data(), upi.data(), uv.data(), num.data());
        CHECK(u >= 32 && u <= m);
        std::vector<float> X(2 * 32 * 63);
        orc_c4_features(umy.data(), uop.data(), 32, X.data());
        orc_c4_features(umy.data(), uop.data(), 32, X.data() + 32 * 63);
        std::vector<float> tp(2 * 32 * 9), tv(2 * 32 * 3);
        for (int r = 0; r < 2; r++) {
            std::copy(upi.begin(), upi.begin() + 32 * 9, tp.begin() + r * 32 * 9);
            std::copy(uv.begin(), uv.begin() + 32 * 3, tv.begin() + r * 32 * 3);
        }
        orc_train_hyper hp{1e-6f, 1.0f, 1.0f, 0.9f, 0.999f, 1e-8f};
        std::vector<float> w = blob, mm(blob.size(), 0.0f), vv(blob.size(), 0.0f), losses(4);
        float lrs[2] = {1e-3f, 1e-3f};
        long long step = 0;
        orc_train_steps(w.data(), &hp, X.data(), tp.data(), tv.data(), 32, 2, lrs, mm.data(), vv.data(), &step, losses.data());
        CHECK(step == 2 && losses[2] < losses[0]);  // the same batch twice: the policy loss must fall
    }
    std::printf(fails ? "sanitize_check: %d check(s) failed\n" : "sanitize_check ok\n", fails);
    return fails ? 1 : 0;
}
