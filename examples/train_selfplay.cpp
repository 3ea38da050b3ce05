// examples/train_selfplay.cpp — what the reference's Worker::process() (include/mainwindow.h:140-150) does, without Qt:
// construct ChessBoard + ChessAI, train N episodes, print every gameCompleted(game, red, black), save the model.
//
//   g++ -std=c++17 -O2 examples/train_selfplay.cpp -Iinclude -Lcn_chess_ai_amd -lxqhip -Wl,-rpath,$PWD/cn_chess_ai_amd -o train_selfplay
//   ./train_selfplay 2000 model.bin [parallel_games] [options]
//
// options (all beyond the reference, which has no flags at all — SURVEY section 5 "Config / flags"):
//   --replay CAP --minibatch MB   replay ring of CAP transitions, MB per update: train() then runs the throughput schedule that
//                                 bench.py measures (xq::ChessAI::setReplay); without them the loop is on-policy like the reference
//   --hidden 256,256              hidden layer widths (default 128: ChessAI::initializeDQN, chessai.cpp:395-404)
//   --save-every N                model_after_<N>_games.bin every N episodes (default 100 = chessai.cpp:165; 0 = never)
//   --derive                      layer-0 sums of s' derived from those of s (what bench.py runs; another summation order)
//   --prefill N                   N uniform-random plies in every game before training (spreads the games over all phases)
//   --seed S                      seed of the batched games (default: time, like the reference's srand(time))
//   --json                        one JSON line with the loop's counters (bench.py's `facade` leg reads it)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "xq/xq.hpp"

int main(int argc, char** argv) {
    int episodes = 1000, parallel = 8192, replay = 0, minibatch = 0, save_every = 100, prefill = 0, npos = 0;
    const char* file = "model.bin";
    std::vector<int> hidden;
    bool derive = false, json = false;
    unsigned long long seed = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--replay") replay = std::atoi(next());
        else if (a == "--minibatch") minibatch = std::atoi(next());
        else if (a == "--save-every") save_every = std::atoi(next());
        else if (a == "--prefill") prefill = std::atoi(next());
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 0);
        else if (a == "--derive") derive = true;
        else if (a == "--json") json = true;
        else if (a == "--hidden") {
            for (const char* p = next(); *p;) { hidden.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        } else if (a.rfind("--", 0) == 0) { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
        else if (npos == 0) { episodes = std::atoi(argv[i]); ++npos; }
        else if (npos == 1) { file = argv[i]; ++npos; }
        else if (npos == 2) { parallel = std::atoi(argv[i]); ++npos; }
    }
    try {
        xq::ChessBoard board;                       // mainwindow.h:130-137: the caller owns the board
        xq::ChessAI ai(&board);
        ai.setParallelGames(parallel);              // 1 = the reference's sequential loop
        if (!hidden.empty()) {
            std::vector<int> sizes{90 * 14};
            for (int h : hidden) sizes.push_back(h);
            sizes.push_back(90 * 90);
            ai.setDQN(std::make_unique<xq::DQN>(sizes));
        }
        if (replay > 0) ai.setReplay(replay, minibatch);
        ai.setSaveInterval(save_every);
        ai.setLayer0Derive(derive);
        ai.setPrefillRandomPlies(prefill);
        if (seed) ai.setBatchSeed(seed);
        int red = 0, black = 0, games = 0;
        ai.gameCompleted = [&](int game, int redScore, int blackScore) {
            ++games; red += redScore; black += blackScore;
            if (!json && game % 500 == 0) std::printf("Game %d completed. Red score: %d Black score: %d\n", game, redScore, blackScore);
        };
        ai.trainingFinished = [&] { ai.saveModel(file); };      // Worker::onTrainingFinished, mainwindow.h:159-167
        const auto t0 = std::chrono::steady_clock::now();
        ai.train(episodes);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const xq::ChessAI::TrainStats st = ai.lastTrainStats();
        if (json) {
            std::printf("{\"entry_point\": \"xq::ChessAI::train(%d)\", \"parallel_games\": %d, \"replay_capacity\": %d, \"minibatch\": %d, "
                        "\"episodes_reported\": %d, \"episodes_finished\": %llu, \"env_steps\": %llu, \"updates\": %llu, "
                        "\"loop_seconds\": %.6f, \"train_call_seconds\": %.6f, \"env_steps_per_s\": %.1f, \"updates_per_s\": %.1f, "
                        "\"screened_steps\": %llu, \"guard_fallbacks\": %llu, \"candidate_groups_per_sample\": %.3f, \"whole_groups_per_sample\": %.3f}\n",
                        episodes, parallel, replay, minibatch, games, (unsigned long long)st.episodes, (unsigned long long)st.envSteps,
                        (unsigned long long)st.updates, st.seconds, s, st.seconds > 0 ? st.envSteps / st.seconds : 0.0,
                        st.seconds > 0 ? st.updates / st.seconds : 0.0, (unsigned long long)st.screenedSteps, (unsigned long long)st.guardFallbacks, st.candidateGroupsPerSample, st.wholeGroupsPerSample);
            return 0;
        }
        std::printf("%d episodes in %.2f s (%.0f episodes/s), mean captured material red %.1f black %.1f, model -> %s\n", games, s,
                    games / s, games ? (double)red / games : 0.0, games ? (double)black / games : 0.0, file);
        if (st.seconds > 0)
            std::printf("batched loop: %llu plies + %llu updates in %.3f s = %.2f M env steps/s + %.0f updates/s\n",
                        (unsigned long long)st.envSteps, (unsigned long long)st.updates, st.seconds, st.envSteps / st.seconds / 1e6,
                        st.updates / st.seconds);
        const auto mv = ai.getAIMove(xq::PieceColor::Red);       // the GUI's "AI move" path, chessai.cpp:29-83
        std::printf("AI move for Red from the final position of the caller's board: (%d,%d) -> (%d,%d)\n", mv.first.first,
                    mv.first.second, mv.second.first, mv.second.second);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
