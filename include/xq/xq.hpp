// xq.hpp — C++ facade over include/xq_capi.h with the class surface of Qervas/cn_chess_ai's training path:
// ChessBoard (include/chessboard.h:35-56), Action (include/action.h), DQN (include/dqn.h:97-116),
// ChessAI (include/chessai.h:17-56) plus the build's ReplayBuffer / VecEnv (SURVEY §8b "New").
//
// Header-only, C++17, links against libxqhip.so only.  Same method names, argument meaning and error behaviour as the
// reference: std::invalid_argument / std::runtime_error where upstream throws them (dqn.cu:17-19,200,324-329;
// dqn.cpp:26-28,80,115,147), QVector/QPair/QString replaced by std::vector/std::pair/std::string, Qt signals by
// std::function callbacks.  Every rule evaluation and every network operation runs on the GPU through the C ABI;
// this header only marshals arguments (and keeps a host copy of the 90 board bytes for getPieceAt()).
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../xq_capi.h"

namespace xq {

// ---- error mapping ---------------------------------------------------------------------------------------------
inline void check(int rc) {
    if (rc == XQ_OK) return;
    const std::string msg = xq_last_error() ? xq_last_error() : "";
    if (rc == XQ_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

// ---- chessboard.h:8-31, action.h ---------------------------------------------------------------------------------
enum class PieceType { Empty, General, Advisor, Elephant, Horse, Chariot, Cannon, Soldier };
enum class PieceColor { Red, Black, None };
struct ChessPiece {
    PieceType type;
    PieceColor color;
    ChessPiece() : type(PieceType::Empty), color(PieceColor::None) {}
    ChessPiece(PieceType t, PieceColor c) : type(t), color(c) {}
};
enum class PieceScore { General = 1000, Advisor = 20, Elephant = 20, Horse = 40, Chariot = 90, Cannon = 45, Soldier = 10 };
inline int getPieceScore(PieceType t) {                        // chessboard.cpp:443-454
    switch (t) {
        case PieceType::General: return 1000; case PieceType::Advisor: return 20; case PieceType::Elephant: return 20;
        case PieceType::Horse: return 40; case PieceType::Chariot: return 90; case PieceType::Cannon: return 45;
        case PieceType::Soldier: return 10; default: return 0;
    }
}
struct Action {
    int from;  // source square 0..89
    int to;    // destination square 0..89
    bool operator==(const Action& o) const { return from == o.from && to == o.to; }
};
inline ChessPiece pieceFromCode(int code) {
    if (code <= 0 || code > 14) return ChessPiece();
    return ChessPiece(static_cast<PieceType>(code > 7 ? code - 7 : code), code > 7 ? PieceColor::Black : PieceColor::Red);
}

// ---- ChessBoard: one game resident on the GPU (an xq_env with n_games = 1) ------------------------------------
class ChessBoard {
public:
    ChessBoard() { check(xq_env_create(1, 0x5EED, 0, nullptr, &env_)); refresh(); }
    ~ChessBoard() { xq_env_destroy(env_); }
    // plain value type like upstream (chessboard.h:33-88): a copy is a second device-resident game in the same state
    ChessBoard(const ChessBoard& o) { check(xq_env_create(1, 0x5EED, 0, nullptr, &env_)); assign(o); }
    ChessBoard& operator=(const ChessBoard& o) { if (this != &o) assign(o); return *this; }

    void initializeBoard() { reset(); }
    ChessPiece getPieceAt(int row, int col) const {             // Empty outside the board, chessboard.cpp:31-36
        if (!isInsideBoard(row, col)) return ChessPiece();
        return pieceFromCode(sq_[row * 9 + col]);
    }
    // chessboard.cpp:38-64: invalid => Empty piece and NO state change; still moves after game over
    ChessPiece movePiece(int fromRow, int fromCol, int toRow, int toCol) {
        int32_t a = -1;
        if (isInsideBoard(fromRow, fromCol) && isInsideBoard(toRow, toCol))
            a = (fromRow * 9 + fromCol) * 90 + toRow * 9 + toCol;
        check(xq_env_step(env_, &a, 0, &last_));
        refresh();
        over_ = last_.terminated != 0; over_ok_ = true;
        return pieceFromCode(last_.captured);
    }
    bool isValidMove(int fromRow, int fromCol, int toRow, int toCol) const {   // chessboard.cpp:66-93
        if (!isInsideBoard(fromRow, fromCol) || !isInsideBoard(toRow, toCol)) return false;
        ensureMatrix();
        return valid_[(fromRow * 9 + fromCol) * 90 + toRow * 9 + toCol] != 0;
    }
    void reset() { check(xq_env_reset(env_)); refresh(); }
    int getRedScore() const { return meta_[2]; }
    int getBlackScore() const { return meta_[3]; }
    // chessboard.cpp:286-309, evaluated on device: the result of the last movePiece() is kept; after reset()/setState() one
    // probe (a rejected move: evaluates, changes nothing) refreshes it
    bool checkGameOver() const {
        if (!over_ok_) { over_ = probe().terminated != 0; over_ok_ = true; }
        return over_;
    }
    PieceColor getWinner() const {                                             // chessboard.cpp:312-320, on device
        uint8_t w = 2;
        check(xq_env_get_winner(env_, 0, 1, &w));
        return w == 0 ? PieceColor::Red : w == 1 ? PieceColor::Black : PieceColor::None;
    }
    PieceColor getCurrentPlayer() const { return meta_[1] == 0 ? PieceColor::Red : PieceColor::Black; }
    int getMoveCount() const { return meta_[0]; }
    // chessboard.cpp:112-147: ordered targets of the piece on (row, col); generated on device
    std::vector<std::pair<int, int>> getValidMoves(int row, int col) const {
        std::vector<std::pair<int, int>> out;
        if (!isInsideBoard(row, col) || sq_[row * 9 + col] == 0) return out;
        const int colour = sq_[row * 9 + col] > 7 ? 1 : 0;
        uint16_t codes[XQ_MAX_MOVES];
        int32_t n = 0;
        check(xq_env_legal_moves(env_, colour, codes, &n));
        for (int k = 0; k < n; ++k)
            if (codes[k] / 90 == row * 9 + col) out.emplace_back((codes[k] % 90) / 9, (codes[k] % 90) % 9);
        return out;
    }
    bool isInsideBoard(int row, int col) const { return row >= 0 && row < 10 && col >= 0 && col < 9; }
    // chessboard.h:50-56 — the per-piece validators as upstream writes them (geometry + occupancy; the piece standing on
    // `from` is not consulted, except for the soldier's colour).  In-board queries come from a cached device-computed table,
    // anything else is one device query (upstream reads squares outside the board as Empty).
    bool isValidGeneralMove(int fr, int fc, int tr, int tc) const { return rule(1, fr, fc, tr, tc); }
    bool isValidAdvisorMove(int fr, int fc, int tr, int tc) const { return rule(2, fr, fc, tr, tc); }
    bool isValidElephantMove(int fr, int fc, int tr, int tc) const { return rule(3, fr, fc, tr, tc); }
    bool isValidHorseMove(int fr, int fc, int tr, int tc) const { return rule(4, fr, fc, tr, tc); }
    bool isValidChariotMove(int fr, int fc, int tr, int tc) const { return rule(5, fr, fc, tr, tc); }
    bool isValidCannonMove(int fr, int fc, int tr, int tc) const { return rule(6, fr, fc, tr, tc); }
    bool isValidSoldierMove(int fr, int fc, int tr, int tc) const { return rule(7, fr, fc, tr, tc); }

    // ---- beyond the reference surface (used by ChessAI) ----
    std::vector<Action> allValidActions(PieceColor player) const {             // chessai.cpp:347-368, on device
        uint16_t codes[XQ_MAX_MOVES];
        int32_t n = 0;
        check(xq_env_legal_moves(env_, player == PieceColor::Black ? 1 : 0, codes, &n));
        std::vector<Action> v((size_t)n);
        for (int k = 0; k < n; ++k) v[k] = Action{codes[k] / 90, codes[k] % 90};
        return v;
    }
    const xq_step_result& lastStep() const { return last_; }
    const uint8_t* squares() const { return sq_; }
    void setState(const uint8_t* squares90, int moveCount, PieceColor player, int red, int black) {
        int32_t m[4] = {moveCount, player == PieceColor::Black ? 1 : 0, red, black};
        check(xq_env_set_state(env_, 0, 1, squares90, m));
        refresh();
    }
    xq_env* handle() const { return env_; }

private:
    void refresh() {
        check(xq_env_get_state(env_, 0, 1, sq_, meta_));
        matrix_ok_ = false; rules_ok_ = false; over_ok_ = false;
    }
    void assign(const ChessBoard& o) {
        check(xq_env_set_state(env_, 0, 1, o.sq_, o.meta_));
        refresh();
        last_ = o.last_;
    }
    bool rule(int type, int fr, int fc, int tr, int tc) const {
        if (isInsideBoard(fr, fc) && isInsideBoard(tr, tc) && !(type >= 5 && type <= 6 && fr == tr && fc == tc)) {
            if (!rules_ok_) { rules_.resize(7 * 8100); check(xq_env_rule_matrix(env_, 0, rules_.data())); rules_ok_ = true; }
            return rules_[(size_t)(type - 1) * 8100 + (fr * 9 + fc) * 90 + tr * 9 + tc] != 0;
        }
        int ok = 0;
        check(xq_env_rule_query(env_, 0, type, fr, fc, tr, tc, &ok));
        return ok != 0;
    }
    void ensureMatrix() const {
        if (!matrix_ok_) { check(xq_env_valid_matrix(env_, 0, valid_)); matrix_ok_ = true; }
    }
    xq_step_result probe() const {               // an invalid action leaves the state untouched but evaluates it
        int32_t a = -1;
        xq_step_result r;
        check(xq_env_step(env_, &a, 0, &r));
        return r;
    }
    xq_env* env_ = nullptr;
    uint8_t sq_[90];
    int32_t meta_[4];
    mutable uint8_t valid_[8100];
    mutable bool matrix_ok_ = false;
    mutable std::vector<uint8_t> rules_;
    mutable bool rules_ok_ = false;
    mutable bool over_ = false, over_ok_ = false;
    xq_step_result last_{};
};

// ---- DQN, dqn.h:97-116 ----------------------------------------------------------------------------------------------
class DQN {
public:
    DQN(const std::vector<int>& layerSizes, double learningRate = 0.001, double gamma = 0.99, uint64_t seed = 0)
        : layerSizes_(layerSizes), learningRate_(learningRate), gamma_(gamma) {
        if (seed == 0) seed = (uint64_t)std::time(nullptr);       // upstream: random_device / time seeded
        check(xq_dqn_create(layerSizes.data(), (int)layerSizes.size(), learningRate, gamma, seed, nullptr, &h_));
        std::srand((unsigned)std::time(nullptr));                 // dqn.cpp:19
    }
    virtual ~DQN() { xq_dqn_destroy(h_); }
    DQN(const DQN&) = delete;
    DQN& operator=(const DQN&) = delete;

    // dqn.cpp:24-56 — two rand() draws like upstream; q indexed by action.to only; first strict maximum wins
    Action selectAction(const std::vector<double>& state, double epsilon, const std::vector<Action> validActions) {
        if (validActions.empty()) throw std::runtime_error("No valid actions available.");
        const double randValue = static_cast<double>(std::rand()) / RAND_MAX;
        if (randValue < epsilon) return validActions[std::rand() % validActions.size()];
        const std::vector<double> q = getQValues(state);
        double maxQ = -std::numeric_limits<double>::infinity();
        Action best = validActions[0];
        for (const auto& a : validActions) {
            if ((size_t)a.to >= q.size()) continue;
            if (q[a.to] > maxQ) { maxQ = q[a.to]; best = a; }
        }
        return best;
    }
    void backpropagate(const std::vector<double>& state, const std::vector<double>& target, double learningRate) {
        if (state.size() != (size_t)layerSizes_.front())
            throw std::invalid_argument("Input size does not match network input layer size.");
        if (target.size() != (size_t)layerSizes_.back())
            throw std::invalid_argument("Target size does not match network output layer size.");
        check(xq_dqn_backpropagate(h_, state.data(), target.data(), 1, learningRate, 1.0, backpropMode_));
    }
    std::vector<double> getQValues(const std::vector<double>& state) { return forward(state, XQ_NET_ONLINE); }
    void updateTargetNetwork() { check(xq_dqn_update_target(h_)); }
    void saveModel(const std::string& filename) { check(xq_dqn_save_model(h_, filename.c_str())); }
    void loadModel(const std::string& filename) { check(xq_dqn_load_model(h_, filename.c_str())); }
    // dqn.cpp:157-172
    void train(const std::vector<double>& state, int action, double reward, const std::vector<double>& nextState, bool done) {
        std::vector<double> currentQ = forward(state, XQ_NET_ONLINE);
        if (done) currentQ[action] = reward;
        else {
            const std::vector<double> nextQ = forward(nextState, XQ_NET_TARGET);
            double m = nextQ[0];
            for (double v : nextQ) if (v > m) m = v;
            currentQ[action] = reward + gamma_ * m;
        }
        backpropagate(state, currentQ, learningRate_);
    }

    // ---- beyond the reference surface ----
    void setBackpropMode(int mode) { backpropMode_ = mode; }      // XQ_BACKPROP_REFERENCE (default) / _TEXTBOOK
    void setParameters(const std::vector<double>& w, const std::vector<double>& b, int net = XQ_NET_ONLINE) {
        size_t nw, nb;
        check(xq_dqn_num_params(h_, &nw, &nb));
        if (w.size() != nw || b.size() != nb) throw std::invalid_argument("parameter count mismatch");
        check(xq_dqn_set_params(h_, net, w.data(), b.data()));
    }
    void getParameters(std::vector<double>& w, std::vector<double>& b, int net = XQ_NET_ONLINE) {   // copyFromDevice
        size_t nw, nb;
        check(xq_dqn_num_params(h_, &nw, &nb));
        w.resize(nw); b.resize(nb);
        check(xq_dqn_get_params(h_, net, w.data(), b.data()));
    }
    const std::vector<int>& layerSizes() const { return layerSizes_; }
    double gamma() const { return gamma_; }
    xq_dqn* handle() const { return h_; }

private:
    std::vector<double> forward(const std::vector<double>& state, int net) {
        if (state.size() != (size_t)layerSizes_.front())
            throw std::invalid_argument("Input size does not match network input layer size.");
        std::vector<double> q((size_t)layerSizes_.back());
        check(xq_dqn_forward(h_, net, state.data(), 1, q.data()));
        return q;
    }
    xq_dqn* h_ = nullptr;
    std::vector<int> layerSizes_;
    double learningRate_, gamma_;
    int backpropMode_ = XQ_BACKPROP_REFERENCE;
};

// ---- ReplayBuffer / VecEnv (build-defined) -----------------------------------------------------------------------------
struct Transition { uint8_t board[90]; int actionTo; float reward; bool done; uint8_t nextBoard[90]; };
class ReplayBuffer {
public:
    explicit ReplayBuffer(int capacity, uint64_t seed = 0) { check(xq_replay_create(capacity, seed, nullptr, &h_)); }
    ~ReplayBuffer() { xq_replay_destroy(h_); }
    ReplayBuffer(const ReplayBuffer&) = delete;
    void push(const Transition& t) {
        const int32_t a = t.actionTo; const uint8_t d = t.done ? 1 : 0;
        check(xq_replay_push_host(h_, 1, t.board, &a, &t.reward, &d, t.nextBoard));
    }
    std::vector<int32_t> sample(int batch) {        // slots; xq_dqn_td_grads_replay consumes them on device
        std::vector<int32_t> s((size_t)batch);
        check(xq_replay_sample(h_, batch, s.data()));
        return s;
    }
    Transition get(int slot) {
        Transition t; int32_t a; uint8_t d;
        check(xq_replay_get(h_, slot, t.board, &a, &t.reward, &d, t.nextBoard));
        t.actionTo = a; t.done = d != 0;
        return t;
    }
    int size() const { int s = 0; check(xq_replay_size(h_, &s, nullptr, nullptr)); return s; }
    int capacity() const { int c = 0; check(xq_replay_size(h_, nullptr, &c, nullptr)); return c; }
    xq_replay* handle() const { return h_; }
private:
    xq_replay* h_ = nullptr;
};

class VecEnv {
public:
    explicit VecEnv(int nGames, uint64_t seed = 0x5EED, uint32_t firstGameId = 0) : n_(nGames) {
        check(xq_env_create(nGames, seed, firstGameId, nullptr, &h_));
    }
    ~VecEnv() { xq_env_destroy(h_); }
    VecEnv(const VecEnv&) = delete;
    int numGames() const { return n_; }
    void reset() { check(xq_env_reset(h_)); }
    // legal_moves: codes[g*128 + k], counts[g]; player -1 = side to move
    void legalMoves(int player, std::vector<uint16_t>& codes, std::vector<int32_t>& counts) {
        codes.resize((size_t)n_ * XQ_MAX_MOVES); counts.resize((size_t)n_);
        check(xq_env_legal_moves(h_, player, codes.data(), counts.data()));
    }
    std::vector<xq_step_result> step(const std::vector<int32_t>& actions, bool autoReset = true) {
        if ((int)actions.size() != n_) throw std::invalid_argument("one action per game");
        std::vector<xq_step_result> r((size_t)n_);
        check(xq_env_step(h_, actions.data(), autoReset ? 1 : 0, r.data()));
        return r;
    }
    xq_env* handle() const { return h_; }
private:
    xq_env* h_ = nullptr;
    int n_;
};

// ---- Comm: the RCCL communicator of the data-parallel loop (build-defined) ------------------------------------
// One process per GPU; rank r plays games [r*n, (r+1)*n) and every update sums the gradient buffer over all ranks.
class Comm {
public:
    // rendezvous through a file every rank can see (rank 0 writes the 128-byte id, the others poll)
    Comm(int rank, int world, const std::string& rendezvousFile, double timeoutSeconds = 120.0) : rank_(rank), world_(world) {
        check(xq_comm_create_from_file(rank, world, rendezvousFile.c_str(), timeoutSeconds, &h_));
    }
    // id drawn by uniqueId() on rank 0 and shipped by the caller's launcher
    Comm(int rank, int world, const std::vector<uint8_t>& id) : rank_(rank), world_(world) {
        if (id.size() != XQ_COMM_ID_BYTES) throw std::invalid_argument("communicator id must be 128 bytes");
        check(xq_comm_create(rank, world, id.data(), &h_));
    }
    static std::vector<uint8_t> uniqueId() { std::vector<uint8_t> id(XQ_COMM_ID_BYTES); check(xq_comm_unique_id(id.data())); return id; }
    ~Comm() { xq_comm_destroy(h_); }
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    int rank() const { return rank_; }
    int world() const { return world_; }
    uint64_t collectivesIssued() const { uint64_t n = 0; check(xq_comm_info(h_, nullptr, nullptr, &n, nullptr)); return n; }
    xq_comm* handle() const { return h_; }
private:
    xq_comm* h_ = nullptr;
    int rank_, world_;
};

// ---- ChessAI, chessai.h:17-56 ---------------------------------------------------------------------------------------
class ChessAI {
public:
    using Move = std::pair<std::pair<int, int>, std::pair<int, int>>;
    explicit ChessAI(ChessBoard* board) : board(board) {
        std::srand(static_cast<unsigned int>(std::time(nullptr)));           // chessai.cpp:13
        logFile.open("game_log.txt", std::ios::out | std::ios::app);         // chessai.cpp:17
    }
    ~ChessAI() = default;

    // signals (chessai.h:35-37) as callbacks
    std::function<void(int, int, int)> gameCompleted;
    std::function<void()> trainingFinished;
    std::function<void()> selfPlayFinished;

    void initializeDQN() {                                                   // chessai.cpp:395-404
        if (!dqn) dqn = std::make_unique<DQN>(std::vector<int>{90 * 14, 128, 90 * 90});
    }
    bool isDQNInitialized() const { return dqn != nullptr; }
    void saveModel(const std::string& filename) {
        if (dqn) dqn->saveModel(filename); else std::fprintf(stderr, "DQN is not initialized. Cannot save model.\n");
    }
    void loadModel(const std::string& filename) {
        if (dqn) dqn->loadModel(filename); else std::fprintf(stderr, "DQN is not initialized. Cannot load model.\n");
    }

    // chessai.cpp:29-83 — epsilon stays 0.1 at play time; 10 re-validated attempts, then a random valid action
    Move getAIMove(PieceColor color) {
        const std::vector<double> state = getStateRepresentation();
        for (int attempt = 0; attempt < 10; ++attempt) {
            const std::vector<Action> validActions = getAllValidActions(color);
            if (validActions.empty()) continue;
            const Action a = dqn->selectAction(state, 0.1, validActions);
            const int fromRow = a.from / 9, fromCol = a.from % 9, toRow = a.to / 9, toCol = a.to % 9;
            const auto validMoves = board->getValidMoves(fromRow, fromCol);
            if (board->getPieceAt(fromRow, fromCol).color == color && !validMoves.empty())
                for (const auto& m : validMoves)
                    if (m.first == toRow && m.second == toCol) return {{fromRow, fromCol}, {toRow, toCol}};
        }
        const std::vector<Action> validActions = getAllValidActions(color);
        if (validActions.empty()) return {{-1, -1}, {-1, -1}};
        const Action r = validActions[(size_t)std::rand() % validActions.size()];
        return {{r.from / 9, r.from % 9}, {r.to / 9, r.to % 9}};
    }

    // One ply of the loop body chessai.cpp:96-143.  Returns false when the episode is over (nothing was played).
    bool trainStep() { return ply(true); }

private:
    // trainLoop = true: ChessAI::train (local 200-ply cap, done includes moveCount+1 >= 200, :96/:119);
    // trainLoop = false: ChessAI::startSelfPlay (no local cap, done = checkGameOver(), :197/:227).
    bool ply(bool trainLoop) {
        const int maxMovesPerGame = 200;
        if (board->checkGameOver() || (trainLoop && moveCount_ >= maxMovesPerGame)) return false;   // :96 / :197
        const std::vector<Action> validActions = getAllValidActions(currentPlayer_);        // :98
        if (validActions.empty()) return false;                                             // :100-103
        const Action sel = dqn->selectAction(state_, 0.1, validActions);                    // :106
        board->movePiece(sel.from / 9, sel.from % 9, sel.to / 9, sel.to % 9);               // :113
        moveCount_ = board->getMoveCount();                                                 // :115
        const double reward = board->lastStep().reward;   // evaluateBoard(currentPlayer, moveCount) on device, :116
        const std::vector<double> nextState = getStateRepresentation();                     // :118
        const bool done = trainLoop ? board->lastStep().done != 0          // checkGameOver() || moveCount+1 >= 200, :119
                                    : board->lastStep().terminated != 0;   // checkGameOver(), :227
        std::vector<double> targetQ = dqn->getQValues(state_);                              // :122
        if (done) targetQ[sel.to] = reward;
        else {
            const std::vector<double> nextQ = dqn->getQValues(nextState);                   // ONLINE net, :126
            double m = nextQ[0];
            for (double v : nextQ) if (v > m) m = v;
            targetQ[sel.to] = reward + gamma * m;
        }
        dqn->backpropagate(state_, targetQ, learningRate);                                  // :131
        state_ = nextState;
        currentPlayer_ = currentPlayer_ == PieceColor::Red ? PieceColor::Black : PieceColor::Red;
        if (moveCount_ % 100 == 0) dqn->updateTargetNetwork();                              // :140
        return true;
    }

public:
    // One episode = one iteration of the `for episode` loop, chessai.cpp:90-167 (minus the periodic save).
    int trainEpisode() {
        initializeDQN();
        beginEpisode();
        int plies = 0;
        while (trainStep()) ++plies;
        ++episodes_;
        if (gameCompleted) gameCompleted(episodes_, board->getRedScore(), board->getBlackScore());   // :162
        return plies;
    }
    void beginEpisode() {
        board->reset();                                                                     // :90
        currentPlayer_ = PieceColor::Red;                                                   // :91
        state_ = getStateRepresentation();                                                  // :92
        moveCount_ = 0;
    }
    // chessai.cpp:85-170.  parallelGames == 1: the reference's sequential loop (trainEpisode() x numEpisodes).
    // parallelGames > 1 (default 8192): the batched device loop (xq_trainer) until numEpisodes episodes finished.
    void train(int numEpisodes) {
        initializeDQN();
        if (parallelGames_ <= 1) {
            for (int e = 0; e < numEpisodes; ++e) {
                trainEpisode();
                if (saveInterval_ > 0 && episodes_ % saveInterval_ == 0) {                                       // :165
                    saveModel("model_after_" + std::to_string(episodes_) + "_games.bin");
                    savedMark_ = episodes_ / saveInterval_;
                }
            }
        } else {
            trainBatched(numEpisodes);
        }
        if (trainingFinished) trainingFinished();                                           // :169
    }
    // chessai.cpp:191-266: same loop driven by board->getCurrentPlayer(), no local 200-ply cap (the board's own cap ends it)
    void startSelfPlay(int numGames) {
        initializeDQN();
        for (int i = 0; i < numGames; ++i) {
            board->reset();
            state_ = getStateRepresentation();
            moveCount_ = 0;
            while (true) {
                currentPlayer_ = board->getCurrentPlayer();                                 // :198
                if (!ply(false)) break;
            }
            if (gameCompleted) gameCompleted(i + 1, board->getRedScore(), board->getBlackScore());
            if ((i + 1) % 100 == 0) saveModel("model_after_" + std::to_string(i + 1) + "_games.bin");
        }
        if (selfPlayFinished) selfPlayFinished();
    }
    // chessai.cpp:370-393 — the game_log.txt line format
    void onGameCompleted(int gameNumber, int redScore, int blackScore) {
        if (!logFile.is_open()) { std::fprintf(stderr, "Log file is not open\n"); return; }
        const char* result = redScore > blackScore ? "Red wins!" : blackScore > redScore ? "Black wins!" : "It's a draw!";
        logFile << "Game " << gameNumber << " completed. Red Score: " << redScore << ", Black Score: " << blackScore << ". "
                << result << "\n";
        if (gameNumber == numGames) logFile << "AI self-play session completed. Total games: " << numGames << "\n\n";
        logFile.flush();
    }

    // ---- beyond the reference surface ----
    void setParallelGames(int n) { parallelGames_ = n; }
    // Replay ring of the batched train() — the ReplayBuffer of SURVEY section 8b ("New"; element = the arguments of DQN::train,
    // dqn.cpp:157): `capacity` transitions in HBM, `minibatch` of them per update.  With a ring, train() runs the throughput schedule
    // of INTEGRATION.md section 4 — the ply of iteration t on a stream of its own beside the TD step of iteration t, both on the
    // parameters of t, the minibatch drawn from the ring minus the slots that ply writes — which is what bench.py measures.
    // capacity 0 (default): on-policy like the reference, every update learns on the plies just played.
    void setReplay(int capacity, int minibatch) { replayCapacity_ = capacity; replayMinibatch_ = minibatch; }
    // "Save model every certain number of games" (chessai.cpp:164-167): model_after_<N>_games.bin every `episodes` finished episodes
    // (default 100, the reference's literal; 0 = never).  The batched loop finishes episodes in bursts and saves ONCE per drain, under
    // the name of the last multiple crossed.
    void setSaveInterval(int episodes) { saveInterval_ = episodes; }
    // layer-0 sums of s' derived from those of s in the batched TD step (xq_dqn_set_l0_derive: another summation order, ~1e-7)
    void setLayer0Derive(bool on) { l0Derive_ = on; }
    // uniform-random plies played in every game before the batched loop starts (spreads the games over all phases; 0 = from the opening)
    void setPrefillRandomPlies(int n) { prefillPlies_ = n; }
    // what the last batched train() did: plies played by this process, updates, episodes finished, wall seconds of the loop
    // (from the first iteration queued to the last one finished, model saves included)
    // screenedSteps / guardFallbacks: TD steps that found max_a' Q(s',a') by exact screening, and how often the screen's guard sent a run of
    // 512 steps to the full fp32 product instead (a net whose outputs lie within the bf16 bound of each other — typical of some fresh nets)
    struct TrainStats { uint64_t envSteps = 0, updates = 0, episodes = 0; double seconds = 0; uint64_t screenedSteps = 0, guardFallbacks = 0;
                        double candidateGroupsPerSample = 0, wholeGroupsPerSample = 0; };   // (sample, 32-output group) pairs the screen left for fp32
                                                                                           // re-evaluation per sample; those re-evaluated as whole groups
    TrainStats lastTrainStats() const { return stats_; }
    // data-parallel train(): this process is rank comm->rank() of comm->world(); its batched games take the id range
    // [rank * parallelGames, (rank + 1) * parallelGames) and every update all-reduces the gradients over RCCL (nullptr = off)
    void setCommunicator(Comm* comm) { comm_ = comm; }
    void setBatchSeed(uint64_t seed) { batchSeed_ = seed; }                  // 0 (default): time-seeded like upstream
    void setDQN(std::unique_ptr<DQN> d) { dqn = std::move(d); }
    DQN* network() { return dqn.get(); }
    std::vector<double> getStateRepresentation() {                           // chessai.cpp:268-289 (encoding only)
        std::vector<double> s(90 * 14, 0.0);
        const uint8_t* sq = board->squares();
        for (int i = 0; i < 90; ++i) if (sq[i]) s[(size_t)i * 14 + sq[i] - 1] = 1.0;
        return s;
    }
    std::vector<Action> getAllValidActions(PieceColor player) const { return board->allValidActions(player); }
    int numGames = 0;                                                        // chessai.h:55
    double learningRate = 0.001;                                             // chessai.h:48
    double gamma = 0.99;                                                     // chessai.h:49

private:
    void trainBatched(int numEpisodes) {
        xq_trainer_config cfg{};
        cfg.n_games = parallelGames_ < numEpisodes ? parallelGames_ : (numEpisodes > 0 ? numEpisodes : 1);
        const auto& ls = dqn->layerSizes();
        cfg.n_sizes = (int)ls.size();
        for (size_t i = 0; i < ls.size(); ++i) cfg.layer_sizes[i] = ls[i];
        cfg.learning_rate = learningRate; cfg.gamma = gamma; cfg.epsilon = 0.1;
        const bool ring = replayCapacity_ > 0;
        if (ring) {                              // setReplay(): the throughput schedule
            cfg.replay_capacity = replayCapacity_ > cfg.n_games ? replayCapacity_ : cfg.n_games;
            cfg.minibatch = replayMinibatch_ > 0 ? replayMinibatch_ : cfg.n_games;
            cfg.overlap_collect = 1;             // xq_trainer_step then queues learn_grads -> collect -> learn_apply
        } else {
            cfg.replay_capacity = 0;             // on-policy, like the reference: learn on the plies just played
            cfg.minibatch = cfg.n_games;
        }
        cfg.td_net = XQ_TD_ONLINE_NET;           // chessai.cpp:126 uses the online net
        cfg.backprop_mode = XQ_BACKPROP_REFERENCE;
        cfg.target_sync_interval = 100; cfg.mean_gradient = 1;
        cfg.seed = batchSeed_ ? batchSeed_ : (uint64_t)std::time(nullptr);   // replicas must share it: it seeds the initial weights
        cfg.first_game_id = comm_ ? (uint32_t)(comm_->rank() * cfg.n_games) : 0u;
        xq_trainer* t = nullptr;
        check(xq_trainer_create(&cfg, nullptr, &t));
        struct Guard { xq_trainer* t; ~Guard() { xq_trainer_destroy(t); } } guard{t};   // released on every path, also when check() throws
        xq_dqn* td = nullptr; xq_env* te = nullptr;
        check(xq_trainer_dqn(t, &td)); check(xq_trainer_env(t, &te));
        std::vector<double> w, b;
        dqn->getParameters(w, b);                // continue from this agent's weights
        check(xq_dqn_set_params(td, XQ_NET_ONLINE, w.data(), b.data()));
        check(xq_dqn_update_target(td));
        check(xq_dqn_set_qmax_mode(td, XQ_QMAX_SCREENED));    // same max_a' Q(s',a'), found by exact screening (large batches only)
        check(xq_dqn_set_l0_derive(td, l0Derive_ ? 1 : 0));
        if (comm_) check(xq_trainer_set_comm(t, comm_->handle()));
        if (prefillPlies_ > 0) check(xq_trainer_random_plies(t, prefillPlies_));
        std::vector<xq_episode_record> rec(4096);
        // Every rank must run the same number of iterations (each carries a collective): the loop ends when the episodes
        // finished on ALL ranks together reach numEpisodes per rank; a rank reports at most its own numEpisodes.
        const uint64_t quota = (uint64_t)numEpisodes * (uint64_t)(comm_ ? comm_->world() : 1);
        int finished = 0;
        uint64_t localDone = 0;
        stats_ = TrainStats();
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            // iterations per drain: a drain synchronises the device (a pipeline bubble), so far from the end the loop runs 64
            // iterations between two of them; close to it 8, so that few plies are played beyond the last episode asked for
            // (the episode ring holds max(4096, 4 n_games) records: 64 iterations finish ~0.3 n_games episodes)
            const long long left = (long long)numEpisodes - finished;
            check(xq_trainer_step(t, left > (long long)cfg.n_games ? 64 : 8));
            int n = 0; uint64_t total = 0;
            do {
                check(xq_env_drain_episodes(te, rec.data(), (int)rec.size(), &n, &total));
                for (int i = 0; i < n && finished < numEpisodes; ++i) {
                    ++finished; ++episodes_;
                    if (gameCompleted) gameCompleted(episodes_, rec[i].red_score, rec[i].black_score);
                }
            } while (n == (int)rec.size());
            if (saveInterval_ > 0 && episodes_ / saveInterval_ > savedMark_) {          // chessai.cpp:164-167
                savedMark_ = episodes_ / saveInterval_;
                check(xq_dqn_save_model(td, ("model_after_" + std::to_string(savedMark_ * saveInterval_) + "_games.bin").c_str()));
            }
            localDone = total;
            uint64_t global = localDone;
            if (comm_) check(xq_comm_sum_u64(comm_->handle(), &global));
            if (comm_ ? global >= quota : finished >= numEpisodes) break;
        }
        check(xq_dqn_get_params(td, XQ_NET_ONLINE, w.data(), b.data()));      // (synchronises: the loop's last iteration has finished)
        stats_.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t eps = 0;
        check(xq_trainer_counters(t, &stats_.envSteps, &stats_.updates, &eps));
        stats_.episodes = eps;
        {
            uint64_t q[4] = {0, 0, 0, 0}; int hold = 0;
            check(xq_dqn_qmax_stats(td, q));
            check(xq_dqn_qmax_guard(td, &stats_.guardFallbacks, &hold));
            stats_.screenedSteps = q[0];
            stats_.candidateGroupsPerSample = q[1] ? (double)q[2] / (double)q[1] : 0.0;
            stats_.wholeGroupsPerSample = q[1] ? (double)q[3] / (double)q[1] : 0.0;
        }
        dqn->setParameters(w, b);
        dqn->updateTargetNetwork();
    }

    ChessBoard* board;
    std::unique_ptr<DQN> dqn;
    std::ofstream logFile;
    std::vector<double> state_;
    PieceColor currentPlayer_ = PieceColor::Red;
    int moveCount_ = 0;
    int episodes_ = 0;
    int parallelGames_ = 8192;
    int replayCapacity_ = 0, replayMinibatch_ = 0;
    int saveInterval_ = 100, savedMark_ = 0;
    bool l0Derive_ = false;
    int prefillPlies_ = 0;
    TrainStats stats_;
    Comm* comm_ = nullptr;
    uint64_t batchSeed_ = 0;
};

}  // namespace xq
