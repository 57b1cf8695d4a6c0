// C++ caller of the drop-in boundary, standing in for the reference's Rust host (SURVEY.md §8b "Callers"): exercises
// include/synthesis_amd.hpp the way study-connect4 would — Policy::eval through the adaptor, run_n_games into a
// ReplayBuffer, deduplicate, one learner step — and prints everything as "key v v v ..." lines (floats as hex bit
// patterns) for tests/test_cpp_host.py to compare with the oracle. No GPU: prints "error <code> <message>", exits 3.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>

#include "synthesis_amd.hpp"

using namespace synthesis;

static void put(const char* key, const float* v, size_t n) {
    std::printf("%s", key);
    for (size_t i = 0; i < n; i++) {
        uint32_t u;
        std::memcpy(&u, &v[i], 4);
        std::printf(" %08x", u);
    }
    std::printf("\n");
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: host_harness <blob.f32>\n"); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<float> blob(30492);
    f.read(reinterpret_cast<char*>(blob.data()), (std::streamsize)(blob.size() * 4));
    if (!f) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    try {
        Engine engine(256, 64);
        engine.load_weights(blob);

        // Policy::eval on a scripted opening, position by position (batch of one), then as one batch
        HipPolicy policy(engine);
        Connect4 game;
        std::vector<Connect4> line{game};
        for (int col : {4, 4, 3, 5, 2, 4, 0, 8, 8}) {
            if (game.step(col)) break;
            line.push_back(game);
        }
        for (size_t i = 0; i < line.size(); i++) {
            auto out = policy.eval(line[i]);
            std::printf("pos %zu %llu %llu %d\n", i, (unsigned long long)line[i].my_bb(), (unsigned long long)line[i].op_bb(),
                        (int)line[i].player());
            put("logits", out.first.data(), 9);
            put("value", out.second.data(), 3);
            put("features", line[i].features().data(), 63);
        }
        std::vector<std::array<float, 9>> bl;
        std::vector<std::array<float, 3>> bv;
        policy.eval_batch(line, bl, bv);
        for (size_t i = 0; i < line.size(); i++) put("batch_logits", bl[i].data(), 9);
        {   // two workers, one HipPolicy each (alpha_zero.rs:192-198), batching at the same time: each on its own context
            int wrong[2] = {0, 0};
            auto work = [&](int w) {
                try {
                    HipPolicy mine(engine);
                    std::vector<std::array<float, 9>> l2;
                    std::vector<std::array<float, 3>> v2;
                    for (int it = 0; it < 300; it++) {
                        std::vector<Connect4> part(line.begin() + (it + w) % 3, line.end());
                        mine.eval_batch(part, l2, v2);
                        for (size_t i = 0; i < part.size(); i++)
                            if (std::memcmp(l2[i].data(), bl[i + (it + w) % 3].data(), 36) || std::memcmp(v2[i].data(), bv[i + (it + w) % 3].data(), 12)) wrong[w]++;
                    }
                } catch (const Error& e) { wrong[w] = -1000 - e.code; }
            };
            std::thread t0(work, 0), t1(work, 1);
            t0.join(); t1.join();
            std::printf("two_policies_batching %d %d\n", wrong[0], wrong[1]);
        }

        // run_n_games -> ReplayBuffer
        RolloutConfig cfg;
        cfg.num_explores = 64;
        syn_counters ctr{};
        ReplayBuffer buffer = run_n_games(engine, cfg, 12, 5, 0, &ctr);
        std::printf("buffer %zu %zu %zu %llu\n", buffer.total_games_played(), buffer.curr_games(), buffer.curr_steps(),
                    (unsigned long long)ctr.policy_evals);
        for (size_t i = 0; i < buffer.curr_steps(); i++) {
            std::printf("step %llu %llu\n", (unsigned long long)buffer.games[i].my_bb(), (unsigned long long)buffer.games[i].op_bb());
            put("pi", buffer.pis[i].data(), 9);
            put("v", buffer.vs[i].data(), 3);
        }
        ReplayBuffer more = run_n_games(engine, cfg, 4, 5, 12);
        buffer.extend(more);
        buffer.keep_last_n_games(10);
        std::printf("kept %zu %zu %zu\n", buffer.total_games_played(), buffer.curr_games(), buffer.curr_steps());
        FlatBatch flat = buffer.deduplicate(engine);
        std::printf("dedup %zu\n", flat.vs.size());

        // a search of the empty board
        auto res = mcts_search(engine, MCTSConfig(), {Connect4()}, 64);
        put("child_N", res[0].child_N, 9);
        std::printf("best %d nodes %u\n", res[0].best_action, res[0].num_nodes);

        // the evaluator's baseline ladder and one network-vs-baseline pairing (before the learner changes the network)
        auto ladder = mcts_vs_mcts(engine, rollout_mcts_cfg(), ActionSelection::NumVisits, Connect4::Red, 60, 30, {1, 2, 3, 4});
        put("vanilla_rewards", ladder.data(), (int)ladder.size());
        auto versus = eval_against_rollout_mcts(engine, MCTSConfig(), 48, ActionSelection::NumVisits, rollout_mcts_cfg(),
                                                ActionSelection::NumVisits, Connect4::Black, 40, {5, 6, 7});
        put("versus_rewards", versus.data(), (int)versus.size());

        {   // eval_against_old: this network against a copy with the first layer's first 200 weights negated, both colours
            std::vector<float> other = blob;
            for (int i = 0; i < 200; i++) other[i] = -other[i];
            const float r[2] = {eval_against_old(engine, MCTSConfig(), 40, ActionSelection::NumVisits, blob, other),
                                eval_against_old(engine, MCTSConfig(), 40, ActionSelection::NumVisits, other, blob)};
            put("old_rewards", r, 2);
            engine.load_weights(blob);
        }

        // one learner step on the first 32 unique states, then self-play continues on the trained network
        Learner learner(engine, blob, 1e-6f, 1.0f, 1.0f);
        std::vector<Connect4> bg(flat.games.begin(), flat.games.begin() + 32);
        std::vector<std::array<float, 9>> bp(flat.pis.begin(), flat.pis.begin() + 32);
        std::vector<std::array<float, 3>> bvv(flat.vs.begin(), flat.vs.begin() + 32);
        auto losses = learner.step(bg, bp, bvv, 1e-3f);
        put("losses", losses.data(), 2);
        learner.publish();
        auto out2 = policy.eval(Connect4());
        put("logits_after_step", out2.first.data(), 9);

        // error behaviour: the reference panics; here a typed exception with the ABI's status code
        try {
            engine.load_weights(std::vector<float>(7));
            std::printf("no_error\n");
        } catch (const Error& e) {
            std::printf("caught %d\n", e.code);
        }
        try {
            Connect4 g;
            for (int i = 0; i < 8; i++) g.step(0);
            std::printf("no_error\n");
        } catch (const Error& e) {
            std::printf("caught %d\n", e.code);
        }
    } catch (const Error& e) {
        std::printf("error %d %s\n", e.code, e.what());
        return 3;
    }
    return 0;
}
