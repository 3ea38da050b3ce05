// Test of the C++ facade (include/xq/xq.hpp) against the CPU oracle — built and run by tests/test_facade_gpu.py.
// Links libxqhip.so (product) and libxqoracle.so (checker).  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <string>

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
        // train() with a replay ring (setReplay: the throughput schedule bench.py measures) + the periodic save of chessai.cpp:164-167:
        // episode 100 is crossed inside this call => model_after_100_games.bin in the reference's file format, loadable
        std::remove("model_after_100_games.bin");
        ai.setReplay(4096, 64);
        std::vector<double> wb, bb2, wa, ba;
        ai.network()->getParameters(wb, bb2);
        ai.train(70);
        CHECK(completed == 111);
        const xq::ChessAI::TrainStats st = ai.lastTrainStats();
        CHECK(st.updates > 0 && st.envSteps == st.updates * 64 && st.episodes >= 70 && st.seconds > 0);
        {
            FILE* f = std::fopen("model_after_100_games.bin", "rb");
            CHECK(f != nullptr);
            if (f) { std::fseek(f, 0, SEEK_END); CHECK(std::ftell(f) == 9650484L); std::fclose(f); }   // DQN::saveModel layout, {1260,128,8100}
        }
        ai.network()->getParameters(wa, ba);
        double moved = 0; for (size_t i = 0; i < wa.size(); ++i) moved = std::fmax(moved, std::fabs(wa[i] - wb[i]));
        CHECK(moved > 0);                                  // the ring loop trained the agent's own network
        xq::DQN probe(std::vector<int>{1260, 128, 8100}, 0.001, 0.99, 7);
        probe.loadModel("model_after_100_games.bin");
        std::vector<double> wp, bp;
        probe.getParameters(wp, bp);
        bool finite = true; for (double v : wp) finite = finite && std::isfinite(v);
        CHECK(finite);
        ai.setSaveInterval(0);
        ai.setReplay(0, 0);
    }
    // ---- getAIMove, the game_log.txt line, startSelfPlay — against the oracle's restatements ----
    {
        auto crand = [](void*) -> int { return std::rand(); };
        const int L[3] = {1260, 128, 8100};
        std::remove("game_log.txt");
        xq::ChessBoard board;
        xq::ChessAI ai(&board);
        ai.setDQN(std::make_unique<xq::DQN>(std::vector<int>{1260, 128, 8100}, 0.001, 0.99, 99));
        std::vector<double> w, bb;
        ai.network()->getParameters(w, bb);
        // (f2) getAIMove under srand(k), both colours, start position and a mid-game position (incl. explore draws)
        xqo_board o; xqo_reset(&o);
        std::mt19937 rng(5);
        for (int k = 0; k < 24; ++k) {
            if (k >= 4) {            // walk both boards a few random plies further
                uint16_t codes[XQO_MAX_MOVES];
                const int n = xqo_all_valid_actions(&o, o.currentPlayer, codes);
                if (n == 0 || xqo_check_game_over(&o)) break;
                const int c = codes[rng() % n];
                xqo_move_piece(&o, (c / 90) / 9, (c / 90) % 9, (c % 90) / 9, (c % 90) % 9);
                board.movePiece((c / 90) / 9, (c / 90) % 9, (c % 90) / 9, (c % 90) % 9);
            }
            for (int color = 0; color < 2; ++color) {
                int want[4];
                std::srand(1000 + k);
                xqo_get_ai_move(&o, color, L, 3, w.data(), bb.data(), crand, nullptr, RAND_MAX, want);
                std::srand(1000 + k);
                const auto mv = ai.getAIMove(color ? xq::PieceColor::Black : xq::PieceColor::Red);
                CHECK(mv.first.first == want[0] && mv.first.second == want[1] && mv.second.first == want[2] && mv.second.second == want[3]);
            }
        }
        {   // a colour with no piece at all: ten empty attempts, then ((-1,-1),(-1,-1)) (chessai.cpp:38-40,70-73)
            uint8_t sq[90] = {0};
            sq[4] = 1;                                   // a lone red general
            board.setState(sq, 0, xq::PieceColor::Black, 0, 0);
            const auto mv = ai.getAIMove(xq::PieceColor::Black);
            CHECK(mv.first.first == -1 && mv.first.second == -1 && mv.second.first == -1 && mv.second.second == -1);
            xqo_board e; memset(&e, 0, sizeof e); e.sq[4] = 1; e.currentPlayer = 1;
            int want[4];
            xqo_get_ai_move(&e, 1, L, 3, w.data(), bb.data(), crand, nullptr, RAND_MAX, want);
            CHECK(want[0] == -1 && want[3] == -1);
            CHECK(board.checkGameOver());                // black general missing
            const auto red = ai.getAIMove(xq::PieceColor::Red);
            CHECK(board.isValidMove(red.first.first, red.first.second, red.second.first, red.second.second));
        }
        // (f4) startSelfPlay(2): same rand() stream on both sides; no local ply cap, done = checkGameOver()
        std::vector<std::pair<int, std::pair<int, int>>> seen;
        ai.gameCompleted = [&](int g, int r, int b) { seen.push_back({g, {r, b}}); ai.onGameCompleted(g, r, b); };
        int finished = 0;
        ai.selfPlayFinished = [&]() { ++finished; };
        ai.numGames = 2;
        std::srand(31337);
        ai.startSelfPlay(2);
        std::srand(31337);
        xqo_episode_stats st[2];
        xqo_board fin;
        std::string want_log;
        for (int g = 0; g < 2; ++g) {
            xqo_selfplay_game(L, 3, w.data(), bb.data(), 0.001, 0.99, 0.1, crand, nullptr, RAND_MAX, 0, &fin, &st[g]);
            char line[256];
            xqo_game_log_line(g + 1, st[g].redScore, st[g].blackScore, 2, line, sizeof line);
            want_log += line;
        }
        CHECK(finished == 1 && seen.size() == 2);
        for (int g = 0; g < 2 && g < (int)seen.size(); ++g)
            CHECK(seen[g].first == g + 1 && seen[g].second.first == st[g].redScore && seen[g].second.second == st[g].blackScore);
        CHECK(std::memcmp(board.squares(), fin.sq, 90) == 0 && board.getMoveCount() == fin.moveCount);
        CHECK(board.checkGameOver() && (int)board.getWinner() == st[1].winner);
        std::vector<double> w2, b2;
        ai.network()->getParameters(w2, b2);
        double err = 0; for (size_t i = 0; i < w.size(); ++i) err = std::fmax(err, std::fabs(w2[i] - w[i]));
        CHECK(err < 1e-4);
        // (f3) the literal game_log.txt text (chessai.cpp:379-385), incl. the session trailer after the last game
        std::ifstream lf("game_log.txt");
        const std::string got((std::istreambuf_iterator<char>(lf)), std::istreambuf_iterator<char>());
        CHECK(got == want_log);
        if (got != want_log) std::printf("log got:\n%s\nwant:\n%s\n", got.c_str(), want_log.c_str());
        char line[256];
        xqo_game_log_line(7, 30, 30, 9, line, sizeof line);
        CHECK(std::string(line) == "Game 7 completed. Red Score: 30, Black Score: 30. It's a draw!\n");
        xqo_game_log_line(9, 10, 45, 9, line, sizeof line);
        CHECK(std::string(line) == "Game 9 completed. Red Score: 10, Black Score: 45. Black wins!\nAI self-play session completed. Total games: 9\n\n");
    }
    // ---- ChessBoard is a value type (chessboard.h:33-88) + the seven public validators (chessboard.h:50-56) ----
    {
        xq::ChessBoard a;
        a.movePiece(2, 1, 9, 1);                         // cannon takes the horse
        xq::ChessBoard b(a);                             // copy = an independent game in the same state
        CHECK(std::memcmp(a.squares(), b.squares(), 90) == 0 && b.getMoveCount() == 1 && b.getRedScore() == a.getRedScore());
        CHECK(b.getCurrentPlayer() == xq::PieceColor::Black);
        b.movePiece(9, 0, 9, 1);                         // chariot retakes on the copy only
        CHECK(a.getMoveCount() == 1 && b.getMoveCount() == 2 && a.getPieceAt(9, 1).type == xq::PieceType::Cannon);
        xq::ChessBoard c;
        c = b;
        CHECK(std::memcmp(c.squares(), b.squares(), 90) == 0 && c.getBlackScore() == b.getBlackScore() && c.getMoveCount() == 2);
        // validators: geometry + occupancy as upstream writes them, for whatever stands on `from`
        CHECK(a.isValidGeneralMove(0, 4, 1, 4) && !a.isValidGeneralMove(0, 4, 0, 2) && !a.isValidGeneralMove(3, 4, 4, 4));
        CHECK(a.isValidAdvisorMove(0, 3, 1, 4) && !a.isValidAdvisorMove(0, 3, 1, 2));
        CHECK(a.isValidAdvisorMove(-1, 2, 0, 3));        // `from` outside the board: only `to` is tested (chessboard.cpp:346-353)
        CHECK(a.isValidElephantMove(0, 2, 2, 4) && !a.isValidElephantMove(3, 2, 5, 4));
        CHECK(a.isValidHorseMove(0, 1, 2, 2) && !a.isValidHorseMove(0, 1, 1, 3));   // (1,3): leg (0,2) is occupied
        CHECK(a.isValidChariotMove(0, 0, 2, 0) && !a.isValidChariotMove(0, 0, 4, 0) && !a.isValidChariotMove(0, 0, 1, 1));
        CHECK(a.isValidCannonMove(2, 7, 9, 7) && !a.isValidCannonMove(2, 7, 7, 7) && a.isValidCannonMove(2, 7, 6, 7));
        CHECK(a.isValidSoldierMove(3, 0, 4, 0) && !a.isValidSoldierMove(3, 0, 3, 1) && a.isValidSoldierMove(6, 0, 5, 0));
        CHECK(a.isValidSoldierMove(4, 4, 3, 4));         // empty `from` takes the Black branch (chessboard.cpp:432-439)
        bool threw = false;
        try { a.isValidChariotMove(0, 0, 0, 0); } catch (const std::runtime_error&) { threw = true; }   // loop overflow upstream
        CHECK(threw);
        // repeated probes of a finished game fabricate nothing
        uint8_t sq[90] = {0};
        sq[4] = 1;
        a.setState(sq, 17, xq::PieceColor::Red, 0, 0);
        for (int i = 0; i < 5; ++i) CHECK(a.checkGameOver());
        xq_episode_record rec[8]; int n = -1; uint64_t total = 99;
        CHECK(xq_env_drain_episodes(a.handle(), rec, 8, &n, &total) == XQ_OK && n == 0 && total == 0);
        uint64_t cnt[6];
        CHECK(xq_env_counters(a.handle(), cnt) == XQ_OK && cnt[1] == 0 && cnt[2] == 0 && cnt[3] == 0);
    }
    // ---- data-parallel train() from C++: RCCL behind the C ABI, one-rank communicator through the file rendezvous ----
    {
        std::remove("xq_comm_id");
        xq::Comm comm(0, 1, "xq_comm_id");
        CHECK(comm.rank() == 0 && comm.world() == 1 && comm.collectivesIssued() == 0);
        const std::vector<int> sizes{1260, 32, 8100};
        std::vector<double> w[2], bb[2];
        for (int pass = 0; pass < 2; ++pass) {         // pass 0: no communicator, pass 1: all-reduce over the one-rank communicator
            xq::ChessBoard board;
            xq::ChessAI ai(&board);
            ai.setDQN(std::make_unique<xq::DQN>(sizes, 0.001, 0.99, 7));
            ai.setParallelGames(32);
            ai.setBatchSeed(4711);
            int completed = 0;
            ai.gameCompleted = [&](int, int, int) { ++completed; };
            if (pass == 1) ai.setCommunicator(&comm);
            ai.train(32);
            CHECK(completed == 32);
            ai.network()->getParameters(w[pass], bb[pass]);
        }
        CHECK(comm.collectivesIssued() > 0);            // one collective per update behind the fused launches (two buckets on two streams otherwise)
        CHECK(w[0] == w[1] && bb[0] == bb[1]);          // sum over one rank = identity: bit-identical training
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
