// Test of the C++ facade (include/xq/xq.hpp) against the CPU oracle — built and run by tests/test_facade_gpu.py.
// Links libxqhip.so (product) and libxqoracle.so (checker).  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../include/xq/xq.hpp"
#include "../../oracle/xq_oracle.h"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static void boardEquals(const xq::ChessBoard& b, const xqo_board& o) {
    CHECK(std::memcmp(b.squares(), o.sq, 90) == 0);
    CHECK(b.getMoveCount() == o.moveCount);
    CHECK((int)b.getCurrentPlayer() == o.currentPlayer);
    CHECK(b.getRedScore() == o.redScore && b.getBlackScore() == o.blackScore);
    CHECK(b.checkGameOver() == (xqo_check_game_over(&o) != 0));
    CHECK((int)b.getWinner() == xqo_get_winner(&o));
}

int main() {
    // ---- ChessBoard vs oracle: 300 random plies incl. invalid attempts ----
    {
        xq::ChessBoard b;
        xqo_board o; xqo_reset(&o);
        std::mt19937 rng(7);
        boardEquals(b, o);
        CHECK(b.getPieceAt(0, 4).type == xq::PieceType::General && b.getPieceAt(0, 4).color == xq::PieceColor::Red);
        CHECK(b.getPieceAt(-1, 3).type == xq::PieceType::Empty && b.getPieceAt(9, 0).color == xq::PieceColor::Black);
        for (int ply = 0; ply < 300 && !b.checkGameOver(); ++ply) {
            const int player = o.currentPlayer;
            uint16_t codes[XQO_MAX_MOVES];
            const int n = xqo_all_valid_actions(&o, player, codes);
            const auto acts = b.allValidActions(player ? xq::PieceColor::Black : xq::PieceColor::Red);
            CHECK((int)acts.size() == n);
            for (int k = 0; k < n && k < (int)acts.size(); ++k) CHECK(acts[k].from * 90 + acts[k].to == codes[k]);
            if (n == 0) break;
            if (ply % 11 == 3) {                      // an invalid attempt: no state change, Empty piece back
                const auto cap = b.movePiece(0, 0, 5, 5);
                CHECK(cap.type == xq::PieceType::Empty);
                CHECK(xqo_move_piece(&o, 0, 0, 5, 5) == 0);
                boardEquals(b, o);
            }
            const int c = codes[rng() % n];
            const int fr = (c / 90) / 9, fc = (c / 90) % 9, tr = (c % 90) / 9, tc = (c % 90) % 9;
            CHECK(b.isValidMove(fr, fc, tr, tc));
            int moves[32];
            const int m = xqo_get_valid_moves(&o, fr, fc, moves);
            const auto vm = b.getValidMoves(fr, fc);
            CHECK((int)vm.size() == m);
            for (int k = 0; k < m && k < (int)vm.size(); ++k) CHECK(vm[k].first * 9 + vm[k].second == moves[k]);
            const auto cap = b.movePiece(fr, fc, tr, tc);
            const int ocap = xqo_move_piece(&o, fr, fc, tr, tc);
            CHECK((cap.type == xq::PieceType::Empty ? 0 : (int)cap.type + (cap.color == xq::PieceColor::Black ? 7 : 0)) == ocap);
            CHECK(b.lastStep().reward == xqo_evaluate_board(&o, player, o.moveCount));
            boardEquals(b, o);
        }
        b.reset(); xqo_reset(&o);
        boardEquals(b, o);
    }
    // ---- DQN: error behaviour + forward/backprop vs oracle ----
    {
        bool threw = false;
        try { xq::DQN bad(std::vector<int>{1260}); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw);
        const std::vector<int> sizes{1260, 128, 8100};
        xq::DQN d(sizes, 0.001, 0.99, 42);
        std::vector<double> w, bb;
        d.getParameters(w, bb);
        CHECK(w.size() == 1198080 && bb.size() == 8228);
        double mx = 0; for (double v : w) mx = std::fmax(mx, std::fabs(v));
        CHECK(mx <= 0.05 && mx > 0.049);                       // U(-0.05, 0.05), dqn.cu:99
        for (double v : bb) CHECK(v == 0.0);
        threw = false;
        try { d.getQValues(std::vector<double>(10, 0.0)); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { d.selectAction(std::vector<double>(1260, 0.0), 0.1, {}); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { d.loadModel("/nonexistent/model.bin"); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
        xqo_board o; xqo_reset(&o);
        std::vector<double> x(1260), q(8100);
        xqo_state_repr(&o, x.data());
        const int L[3] = {1260, 128, 8100};
        xqo_nn_forward(L, 3, w.data(), bb.data(), x.data(), q.data());
        const auto got = d.getQValues(x);
        double err = 0; for (int i = 0; i < 8100; ++i) err = std::fmax(err, std::fabs(got[i] - q[i]));
        CHECK(err < 1e-4);
        std::vector<double> target = q; target[20] = -0.5;
        xqo_nn_backprop(L, 3, w.data(), bb.data(), x.data(), target.data(), 0.01, 0);
        d.backpropagate(x, target, 0.01);
        std::vector<double> w2, b2;
        d.getParameters(w2, b2);
        err = 0; for (size_t i = 0; i < w.size(); ++i) err = std::fmax(err, std::fabs(w2[i] - w[i]));
        CHECK(err < 2e-5);
    }
    // ---- ChessAI::trainStep x 12 vs the oracle's loop body with the same rand() sequence ----
    {
        xq::ChessBoard board;
        xq::ChessAI ai(&board);
        ai.initializeDQN();
        CHECK(ai.isDQNInitialized());
        const int L[3] = {1260, 128, 8100};
        std::vector<double> w, bb;
        ai.network()->getParameters(w, bb);
        int completed = 0;
        ai.gameCompleted = [&](int, int, int) { ++completed; };
        ai.beginEpisode();
        xqo_board o; xqo_reset(&o);
        int player = 0;
        std::vector<double> state(1260), next(1260), q(8100), tq(8100);
        xqo_state_repr(&o, state.data());
        std::srand(2024);
        std::vector<int> draws;                       // replay the exact rand() stream for the oracle side
        {   // pre-draw: the facade consumes rand() in the same order the reference does
            std::srand(2024);
            for (int i = 0; i < 64; ++i) draws.push_back(std::rand());
            std::srand(2024);
        }
        size_t di = 0;
        for (int ply = 0; ply < 12; ++ply) {
            CHECK(ai.trainStep());
            uint16_t codes[XQO_MAX_MOVES];
            const int n = xqo_all_valid_actions(&o, player, codes);
            xqo_nn_forward(L, 3, w.data(), bb.data(), state.data(), q.data());
            const int r1 = draws[di++];
            int r2 = 0;
            if ((double)r1 / RAND_MAX < 0.1) r2 = draws[di++];
            const int idx = xqo_select_action(q.data(), 8100, codes, n, r1, r2, RAND_MAX, 0.1);
            const int from = codes[idx] / 90, to = codes[idx] % 90;
            xqo_move_piece(&o, from / 9, from % 9, to / 9, to % 9);
            const double reward = xqo_evaluate_board(&o, player, o.moveCount);
            xqo_state_repr(&o, next.data());
            const int done = xqo_check_game_over(&o) || o.moveCount + 1 >= 200;
            xqo_td_target(L, 3, w.data(), bb.data(), state.data(), next.data(), to, reward, done, 0.99, tq.data());
            xqo_nn_backprop(L, 3, w.data(), bb.data(), state.data(), tq.data(), 0.001, 0);
            state = next; player ^= 1;
            boardEquals(board, o);
        }
        std::vector<double> w2, b2;
        ai.network()->getParameters(w2, b2);
        double err = 0; for (size_t i = 0; i < w.size(); ++i) err = std::fmax(err, std::fabs(w2[i] - w[i]));
        CHECK(err < 1e-4);
        // getAIMove returns a currently valid move of that colour
        const auto mv = ai.getAIMove(board.getCurrentPlayer());
        CHECK(board.isValidMove(mv.first.first, mv.first.second, mv.second.first, mv.second.second));
        // a whole episode + the batched train() path run and report through the callback
        const int plies = ai.trainEpisode();
        CHECK(plies > 0 && plies <= 200 && completed == 1);
        ai.setParallelGames(64);
        ai.train(40);
        CHECK(completed == 41);
        ai.saveModel("/tmp/xq_facade_model.bin");
        ai.loadModel("/tmp/xq_facade_model.bin");
    }
    // ---- ReplayBuffer / VecEnv ----
    {
        xq::ReplayBuffer rb(8, 3);
        xq::Transition t{};
        t.board[4] = 1; t.nextBoard[13] = 1; t.actionTo = 13; t.reward = -2.5f; t.done = true;
        rb.push(t);
        CHECK(rb.size() == 1 && rb.capacity() == 8);
        const auto g = rb.get(0);
        CHECK(g.actionTo == 13 && g.reward == -2.5f && g.done && g.board[4] == 1 && g.nextBoard[13] == 1);
        CHECK(rb.sample(5).size() == 5);
        xq::VecEnv env(3);
        std::vector<uint16_t> codes; std::vector<int32_t> counts;
        env.legalMoves(-1, codes, counts);
        CHECK(counts[0] == 44 && counts[2] == 44 && codes[0] == 0 * 90 + 9);
        const auto res = env.step({27 * 90 + 36, -1, 19 * 90 + 82});
        CHECK(res[0].valid && !res[1].valid && res[2].valid && res[2].captured == 11);   // cannon takes the horse on 82
    }
    std::printf(failures ? "facade_test: %d FAILURES\n" : "facade_test: all checks passed\n", failures);
    return failures ? 1 : 0;
}
