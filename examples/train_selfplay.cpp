// examples/train_selfplay.cpp — what the reference's Worker::process() (include/mainwindow.h:140-150) does, without Qt:
// construct ChessBoard + ChessAI, train N episodes, print every gameCompleted(game, red, black), save the model.
//
//   g++ -std=c++17 -O2 examples/train_selfplay.cpp -Iinclude -Lcn_chess_ai_amd -lxqhip -Wl,-rpath,$PWD/cn_chess_ai_amd -o train_selfplay
//   ./train_selfplay 2000 model.bin [parallel_games]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "xq/xq.hpp"

int main(int argc, char** argv) {
    const int episodes = argc > 1 ? std::atoi(argv[1]) : 1000;
    const char* file = argc > 2 ? argv[2] : "model.bin";
    const int parallel = argc > 3 ? std::atoi(argv[3]) : 8192;
    try {
        xq::ChessBoard board;                       // mainwindow.h:130-137: the caller owns the board
        xq::ChessAI ai(&board);
        ai.setParallelGames(parallel);              // 1 = the reference's sequential loop
        int red = 0, black = 0, games = 0;
        ai.gameCompleted = [&](int game, int redScore, int blackScore) {
            ++games; red += redScore; black += blackScore;
            if (game % 500 == 0) std::printf("Game %d completed. Red score: %d Black score: %d\n", game, redScore, blackScore);
        };
        ai.trainingFinished = [&] { ai.saveModel(file); };      // Worker::onTrainingFinished, mainwindow.h:159-167
        const auto t0 = std::chrono::steady_clock::now();
        ai.train(episodes);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%d episodes in %.2f s (%.0f episodes/s), mean captured material red %.1f black %.1f, model -> %s\n", games, s,
                    games / s, games ? (double)red / games : 0.0, games ? (double)black / games : 0.0, file);
        const auto mv = ai.getAIMove(xq::PieceColor::Red);       // the GUI's "AI move" path, chessai.cpp:29-83
        std::printf("AI move for Red from the final position of the caller's board: (%d,%d) -> (%d,%d)\n", mv.first.first,
                    mv.first.second, mv.second.first, mv.second.second);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
