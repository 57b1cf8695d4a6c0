// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the test-local TicTacToe game the reference uses to pin its MCTS:
//   synthesis/src/mcts.rs:499-686 (PlayerId, Action, ActionIterator, won, Game<9> impl)
// Needed so the reference's known-answer tests mcts.rs:691-868 can be replayed against oracle/mcts.hpp.
#pragma once
#include <cstdint>

namespace oracle {

struct TicTacToe {
    static constexpr int N = 9;
    static constexpr int MAX_TURNS = 9;

    // board[row][col]: 0 = empty, 1 = X, 2 = O (mcts.rs:540). X moves first (mcts.rs:613).
    uint8_t board[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    uint8_t player = 1;
    uint8_t turn = 0;

    static TicTacToe new_game() { return TicTacToe{}; }
    int player_id() const { return player; }
    static int next_player(int p) { return p == 1 ? 2 : 1; }  // mcts.rs:505-516 (prev == next)

    // mcts.rs:567-595
    bool won(int p) const {
        for (int r = 0; r < 3; r++)
            if (board[r][0] == p && board[r][1] == p && board[r][2] == p) return true;
        for (int c = 0; c < 3; c++)
            if (board[0][c] == p && board[1][c] == p && board[2][c] == p) return true;
        if (board[0][0] == p && board[1][1] == p && board[2][2] == p) return true;
        if (board[0][2] == p && board[1][1] == p && board[2][0] == p) return true;
        return false;
    }

    // mcts.rs:622-624
    bool is_over() const { return won(player) || won(next_player(player)) || turn == 9; }

    // mcts.rs:626-634
    float reward(int player_id) const {
        if (won(player_id)) return 1.0f;
        if (won(next_player(player_id))) return -1.0f;
        return 0.0f;
    }

    // mcts.rs:545-564, 636-641: empty cells in ascending index order, index = row*3 + col.
    int legal_actions(int out[N]) const {
        int n = 0;
        for (int i = 0; i < 9; i++)
            if (board[i / 3][i % 3] == 0) out[n++] = i;
        return n;
    }

    // mcts.rs:642-650
    bool step(int action) {
        board[action / 3][action % 3] = player;
        player = (uint8_t)next_player(player);
        turn += 1;
        return is_over();
    }

    bool operator==(const TicTacToe& o) const {
        for (int i = 0; i < 9; i++)
            if (board[i / 3][i % 3] != o.board[i / 3][i % 3]) return false;
        return player == o.player && turn == o.turn;
    }
};

}  // namespace oracle
