// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the reference's 9-wide x 7-high Connect4:
//   study-connect4/src/connect4.rs:15-83   (layout, masks, won)
//   study-connect4/src/connect4.rs:108-258 (struct, Hash, FreeColumns, Game impl incl. step/features)
// Bit layout (connect4.rs:3-13): column-major, 7 bits per column, bit = row + 7*col, row 0 = bottom.
// Parity: pinned by the reference's tests connect4.rs:299-498 (replayed in tests/test_oracle_kats.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>

namespace oracle {

namespace c4masks {
// connect4.rs:37-75
constexpr uint64_t FAB_COL = 0x7Full;
constexpr uint64_t fab_row() {
    uint64_t r = 0;
    for (int c = 0; c < 9; c++) r |= 1ull << (7 * c);
    return r;
}
constexpr uint64_t col_mask(int c) { return FAB_COL << (7 * c); }
constexpr uint64_t row_mask(int r) { return fab_row() << r; }
constexpr uint64_t cols0to5() {
    return col_mask(0) | col_mask(1) | col_mask(2) | col_mask(3) | col_mask(4) | col_mask(5);
}
constexpr uint64_t D1_MASK = cols0to5() & (row_mask(3) | row_mask(4) | row_mask(5) | row_mask(6));
constexpr uint64_t D2_MASK = cols0to5() & (row_mask(0) | row_mask(1) | row_mask(2) | row_mask(3));
constexpr uint64_t H_MASK = cols0to5();
constexpr uint64_t V_MASK = row_mask(0) | row_mask(1) | row_mask(2) | row_mask(3);
}  // namespace c4masks

struct Connect4 {
    static constexpr int N = 9;          // MAX_NUM_ACTIONS (connect4.rs:173)
    static constexpr int WIDTH = 9;      // connect4.rs:15
    static constexpr int HEIGHT = 7;     // connect4.rs:16
    static constexpr int MAX_TURNS = 63; // connect4.rs:176
    static constexpr int NUM_FEATURES = 63;

    uint64_t my_bb = 0;
    uint64_t op_bb = 0;
    uint8_t height[WIDTH] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint8_t player = 0;  // 0 = Red, 1 = Black (connect4.rs:18-22); Red moves first (connect4.rs:186)

    // connect4.rs:77-83. The reference sums the four words with `+` and tests `> 0`; in release Rust that add wraps.
    static bool won(uint64_t bb) {
        uint64_t d1 = bb & (bb >> 6) & (bb >> 12) & (bb >> 18) & c4masks::D1_MASK;
        uint64_t d2 = bb & (bb >> 8) & (bb >> 16) & (bb >> 24) & c4masks::D2_MASK;
        uint64_t h = bb & (bb >> 7) & (bb >> 14) & (bb >> 21) & c4masks::H_MASK;
        uint64_t v = bb & (bb >> 1) & (bb >> 2) & (bb >> 3) & c4masks::V_MASK;
        return (uint64_t)(v + h + d1 + d2) > 0;
    }

    static Connect4 new_game() { return Connect4{}; }  // connect4.rs:181-188

    int player_id() const { return player; }          // connect4.rs:190-192
    static int next_player(int p) { return p ^ 1; }   // connect4.rs:24-35

    // connect4.rs:163-171: winner is only ever the side that just moved (its stones are now in op_bb).
    int winner() const { return won(op_bb) ? next_player(player) : -1; }

    // connect4.rs:195-197
    bool is_over() const {
        if (winner() >= 0) return true;
        for (int c = 0; c < WIDTH; c++)
            if (height[c] != HEIGHT) return false;
        return true;
    }

    // connect4.rs:199-212
    float reward(int player_id) const {
        int w = winner();
        if (w < 0) return 0.0f;
        return w == player_id ? 1.0f : -1.0f;
    }

    // connect4.rs:138-161, 214-219: free columns in ascending column order.
    int legal_actions(int out[N]) const {
        int n = 0;
        for (int c = 0; c < WIDTH; c++)
            if (height[c] < HEIGHT) out[n++] = c;
        return n;
    }

    // connect4.rs:221-233
    bool step(int col) {
        my_bb ^= 1ull << (height[col] + HEIGHT * col);
        height[col] += 1;
        uint64_t t = my_bb;
        my_bb = op_bb;
        op_bb = t;
        player = (uint8_t)next_player(player);
        return is_over();
    }

    // connect4.rs:235-258: one 7x9 plane, flat index row*9 + col.
    void features(float s[NUM_FEATURES]) const {
        for (int row = 0; row < HEIGHT; row++) {
            for (int col = 0; col < WIDTH; col++) {
                uint64_t index = 1ull << (row + HEIGHT * col);
                float v;
                if (my_bb & index) v = 1.0f;
                else if (op_bb & index) v = -1.0f;
                else v = -0.1f;
                s[row * WIDTH + col] = v;
            }
        }
        for (int col = 0; col < WIDTH; col++) {
            int h = height[col];
            if (h < HEIGHT) s[h * WIDTH + col] = 0.1f;
        }
    }

    // Rebuild a position from the two bitboards (heights are popcounts of the occupied columns; the side to
    // move is Red iff an even number of stones is on the board). Used by the C API to accept (my_bb, op_bb).
    static Connect4 from_bitboards(uint64_t my, uint64_t op) {
        Connect4 g;
        g.my_bb = my;
        g.op_bb = op;
        uint64_t occ = my | op;
        int stones = 0;
        for (int c = 0; c < WIDTH; c++) {
            g.height[c] = (uint8_t)__builtin_popcountll(occ & c4masks::col_mask(c));
            stones += g.height[c];
        }
        g.player = (uint8_t)(stones & 1);
        return g;
    }

    // connect4.rs:108 derives Eq over all fields; Hash (116-121) covers only the two bitboards.
    bool operator==(const Connect4& o) const {
        return my_bb == o.my_bb && op_bb == o.op_bb && std::memcmp(height, o.height, WIDTH) == 0 &&
               player == o.player;
    }
};

struct Connect4Hash {
    size_t operator()(const Connect4& g) const {
        uint64_t x = g.my_bb * 0x9E3779B97F4A7C15ull ^ (g.op_bb + 0x7F4A7C159E3779B9ull);
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        return (size_t)x;
    }
};

}  // namespace oracle
