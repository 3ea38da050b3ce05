// TEST INFRASTRUCTURE — not product code.
//
// Driver around the REAL reference rules engine.  It is linked against
// /root/reference/src/chessboard.cpp compiled UNMODIFIED (see oracle/Makefile,
// target `ref`); nothing from the reference is copied into this repository.
// The binary lands in oracle/_ref/ (git-ignored) and is used
//   (1) by oracle/gen_golden.py to produce tests/golden/*.npz, and
//   (2) by tests/test_oracle_vs_ref.py (only when the binary is present) to
//       cross-check oracle/xq_oracle.c live on fresh seeds.
//
// Only the public ChessBoard API (include/chessboard.h:35-56) is used.  The
// square scan that concatenates per-square lists restates
// ChessAI::getAllValidActions (src/chessai.cpp:347-368): rows 0..9, cols 0..8,
// pieces of `player`, per-square getValidMoves() order.  chessai.cpp itself is
// not buildable here without stand-ins for cuda_runtime.h / QRandomGenerator,
// so it is NOT part of this build.
#include "chessboard.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
};

inline uint8_t pieceCode(const ChessPiece& p) {
    if (p.type == PieceType::Empty) return 0;
    return (uint8_t)(static_cast<int>(p.type) + (p.color == PieceColor::Black ? 7 : 0));
}

int allActions(const ChessBoard& b, PieceColor player, uint16_t* out) {
    int n = 0;
    for (int row = 0; row < 10; ++row)
        for (int col = 0; col < 9; ++col) {
            ChessPiece piece = b.getPieceAt(row, col);
            if (piece.color == player) {
                const auto moves = b.getValidMoves(row, col);
                for (const auto& m : moves) {
                    if (n < 128) out[n] = (uint16_t)((row * 9 + col) * 90 + m.first * 9 + m.second);
                    ++n;
                }
            }
        }
    return n;
}

#pragma pack(push, 1)
struct Record {
    uint8_t board[90];
    int32_t moveCount;
    uint8_t player;      // ChessBoard::getCurrentPlayer(): 0 red, 1 black
    int32_t redScore, blackScore;
    uint8_t over;        // checkGameOver()
    uint8_t winner;      // getWinner(): 0 red, 1 black, 2 none
    uint16_t nRed, nBlack;
    uint16_t red[128], black[128];   // from*90+to, canonical order
    int8_t fr, fc, tr, tc;           // the move attempted after this snapshot
    uint8_t valid;                   // isValidMove(fr,fc,tr,tc) before the attempt
    uint8_t captured;                // piece code movePiece() returned
};
#pragma pack(pop)

void snapshot(const ChessBoard& b, Record& r) {
    std::memset(&r, 0, sizeof r);
    for (int s = 0; s < 90; ++s) r.board[s] = pieceCode(b.getPieceAt(s / 9, s % 9));
    r.moveCount = b.getMoveCount();
    r.player = (uint8_t)static_cast<int>(b.getCurrentPlayer());
    r.redScore = b.getRedScore();
    r.blackScore = b.getBlackScore();
    r.over = b.checkGameOver();
    r.winner = (uint8_t)static_cast<int>(b.getWinner());
    r.nRed = (uint16_t)allActions(b, PieceColor::Red, r.red);
    r.nBlack = (uint16_t)allActions(b, PieceColor::Black, r.black);
}

// trace: random self-play, one Record per attempted move.
int cmdTrace(uint64_t seed, int ngames, const char* path) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { std::perror(path); return 1; }
    SplitMix64 rng(seed);
    ChessBoard b;
    long nrec = 0;
    for (int g = 0; g < ngames; ++g) {
        b.reset();
        int extra = 2;   // attempts made after game over (movePiece still works then)
        while (true) {
            bool over = b.checkGameOver();
            if (over && extra-- <= 0) break;
            Record r;
            snapshot(b, r);
            PieceColor player = b.getCurrentPlayer();
            const uint16_t* list = (player == PieceColor::Red) ? r.red : r.black;
            int n = (player == PieceColor::Red) ? r.nRed : r.nBlack;
            uint32_t dice = rng.below(16);
            int fr, fc, tr, tc;
            if (dice == 0 || n == 0) {            // arbitrary (mostly invalid) attempt
                fr = (int)rng.below(12) - 1; fc = (int)rng.below(11) - 1;
                tr = (int)rng.below(12) - 1; tc = (int)rng.below(11) - 1;
            } else if (dice == 1) {               // the OTHER side moves out of turn (no turn check upstream)
                const uint16_t* ol = (player == PieceColor::Red) ? r.black : r.red;
                int on = (player == PieceColor::Red) ? r.nBlack : r.nRed;
                if (on == 0) { ol = list; on = n; }
                int c = ol[rng.below((uint32_t)on)];
                fr = (c / 90) / 9; fc = (c / 90) % 9; tr = (c % 90) / 9; tc = (c % 90) % 9;
            } else {
                int c = list[rng.below((uint32_t)n)];
                fr = (c / 90) / 9; fc = (c / 90) % 9; tr = (c % 90) / 9; tc = (c % 90) % 9;
            }
            r.fr = (int8_t)fr; r.fc = (int8_t)fc; r.tr = (int8_t)tr; r.tc = (int8_t)tc;
            r.valid = b.isValidMove(fr, fc, tr, tc);
            r.captured = pieceCode(b.movePiece(fr, fc, tr, tc));
            std::fwrite(&r, sizeof r, 1, f);
            ++nrec;
            if (n == 0 && !over) break;
        }
    }
    std::fclose(f);
    std::fprintf(stderr, "trace: %ld records of %zu bytes\n", nrec, sizeof(Record));
    return 0;
}

// validmat: for sampled random-play positions dump board + the full 90x90 isValidMove matrix.
int cmdValidMat(uint64_t seed, int npos, const char* path) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { std::perror(path); return 1; }
    SplitMix64 rng(seed);
    ChessBoard b;
    int written = 0;
    while (written < npos) {
        b.reset();
        while (!b.checkGameOver() && written < npos) {
            uint16_t list[128];
            int n = allActions(b, b.getCurrentPlayer(), list);
            if (n == 0) break;
            if (rng.below(8) == 0) {
                uint8_t board[90];
                uint8_t mat[8100];
                for (int s = 0; s < 90; ++s) board[s] = pieceCode(b.getPieceAt(s / 9, s % 9));
                for (int fsq = 0; fsq < 90; ++fsq)
                    for (int tsq = 0; tsq < 90; ++tsq)
                        mat[fsq * 90 + tsq] = b.isValidMove(fsq / 9, fsq % 9, tsq / 9, tsq % 9);
                std::fwrite(board, 1, 90, f);
                std::fwrite(mat, 1, 8100, f);
                ++written;
            }
            int c = list[rng.below((uint32_t)(n > 128 ? 128 : n))];
            b.movePiece((c / 90) / 9, (c / 90) % 9, (c % 90) / 9, (c % 90) % 9);
        }
    }
    std::fclose(f);
    return 0;
}

// tracebig: random play; one Record per position in which either side has MORE THAN 64 moves (the second half of the
// 128-entry move list), with one attempted move each.  Records do not chain.
int cmdTraceBig(uint64_t seed, int nrec, const char* path) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { std::perror(path); return 1; }
    SplitMix64 rng(seed);
    ChessBoard b;
    int written = 0;
    long plies = 0;
    while (written < nrec) {
        b.reset();
        while (!b.checkGameOver() && written < nrec) {
            Record r;
            snapshot(b, r);
            PieceColor player = b.getCurrentPlayer();
            const uint16_t* list = (player == PieceColor::Red) ? r.red : r.black;
            int n = (player == PieceColor::Red) ? r.nRed : r.nBlack;
            if (n == 0) break;
            int c = list[rng.below((uint32_t)(n > 128 ? 128 : n))];
            int fr = (c / 90) / 9, fc = (c / 90) % 9, tr = (c % 90) / 9, tc = (c % 90) % 9;
            r.fr = (int8_t)fr; r.fc = (int8_t)fc; r.tr = (int8_t)tr; r.tc = (int8_t)tc;
            r.valid = b.isValidMove(fr, fc, tr, tc);
            r.captured = pieceCode(b.movePiece(fr, fc, tr, tc));
            ++plies;
            if (r.nRed > 64 || r.nBlack > 64) { std::fwrite(&r, sizeof r, 1, f); ++written; }
        }
    }
    std::fclose(f);
    std::fprintf(stderr, "tracebig: %d records out of %ld plies\n", written, plies);
    return 0;
}

// rulemat: for sampled random-play positions dump the board, the seven PUBLIC per-piece validators
// (chessboard.h:50-56) over all in-board (from, to) pairs — 7 x 8100 bytes, type-major — and 64 queries with
// coordinates in [-2, 11] (squares outside the board read as Empty upstream): 5 x int8 (type, fr, fc, tr, tc) + result.
// from == to is skipped for chariot / cannon: the path loop `i != end` overflows there.
int cmdRuleMat(uint64_t seed, int npos, const char* path) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { std::perror(path); return 1; }
    SplitMix64 rng(seed);
    ChessBoard b;
    int written = 0;
    auto rule = [&](int type, int fr, int fc, int tr, int tc) -> bool {
        switch (type) {
            case 1: return b.isValidGeneralMove(fr, fc, tr, tc);
            case 2: return b.isValidAdvisorMove(fr, fc, tr, tc);
            case 3: return b.isValidElephantMove(fr, fc, tr, tc);
            case 4: return b.isValidHorseMove(fr, fc, tr, tc);
            case 5: return b.isValidChariotMove(fr, fc, tr, tc);
            case 6: return b.isValidCannonMove(fr, fc, tr, tc);
            default: return b.isValidSoldierMove(fr, fc, tr, tc);
        }
    };
    while (written < npos) {
        b.reset();
        while (!b.checkGameOver() && written < npos) {
            uint16_t list[128];
            int n = allActions(b, b.getCurrentPlayer(), list);
            if (n == 0) break;
            if (rng.below(16) == 0) {
                uint8_t board[90];
                static uint8_t mat[7 * 8100];
                for (int s = 0; s < 90; ++s) board[s] = pieceCode(b.getPieceAt(s / 9, s % 9));
                for (int type = 1; type <= 7; ++type)
                    for (int fsq = 0; fsq < 90; ++fsq)
                        for (int tsq = 0; tsq < 90; ++tsq)
                            mat[(type - 1) * 8100 + fsq * 90 + tsq] =
                                (fsq == tsq && (type == 5 || type == 6)) ? 0 : rule(type, fsq / 9, fsq % 9, tsq / 9, tsq % 9);
                std::fwrite(board, 1, 90, f);
                std::fwrite(mat, 1, sizeof mat, f);
                for (int q = 0; q < 64; ++q) {
                    int8_t t[6];
                    do {
                        t[0] = (int8_t)(1 + rng.below(7));
                        t[1] = (int8_t)((int)rng.below(14) - 2); t[2] = (int8_t)((int)rng.below(13) - 2);
                        if (rng.below(2)) { t[3] = (int8_t)(t[1] + (int)rng.below(5) - 2); t[4] = (int8_t)(t[2] + (int)rng.below(5) - 2); }
                        else { t[3] = (int8_t)((int)rng.below(14) - 2); t[4] = (int8_t)((int)rng.below(13) - 2); }
                    } while ((t[0] == 5 || t[0] == 6) && t[1] == t[3] && t[2] == t[4]);
                    t[5] = rule(t[0], t[1], t[2], t[3], t[4]);
                    std::fwrite(t, 1, 6, f);
                }
                ++written;
            }
            int c = list[rng.below((uint32_t)(n > 128 ? 128 : n))];
            b.movePiece((c / 90) / 9, (c / 90) % 9, (c % 90) / 9, (c % 90) % 9);
        }
    }
    std::fclose(f);
    return 0;
}

// bench: env-only random-policy stepping (move-gen for the side to move + movePiece + game-over test),
// the reference-CPU leg of BASELINE.md §3 C2.  Prints "steps seconds".
int cmdBench(uint64_t seed, double seconds) {
    SplitMix64 rng(seed);
    ChessBoard b;
    long steps = 0;
    auto t0 = std::chrono::steady_clock::now();
    double el = 0;
    while (el < seconds) {
        b.reset();
        while (!b.checkGameOver()) {
            uint16_t list[128];
            int n = allActions(b, b.getCurrentPlayer(), list);
            if (n == 0) break;
            int c = list[rng.below((uint32_t)(n > 128 ? 128 : n))];
            b.movePiece((c / 90) / 9, (c / 90) % 9, (c % 90) / 9, (c % 90) % 9);
            ++steps;
        }
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    std::printf("%ld %.6f\n", steps, el);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc >= 5 && !std::strcmp(argv[1], "trace"))
        return cmdTrace(std::strtoull(argv[2], nullptr, 0), std::atoi(argv[3]), argv[4]);
    if (argc >= 5 && !std::strcmp(argv[1], "validmat"))
        return cmdValidMat(std::strtoull(argv[2], nullptr, 0), std::atoi(argv[3]), argv[4]);
    if (argc >= 5 && !std::strcmp(argv[1], "tracebig"))
        return cmdTraceBig(std::strtoull(argv[2], nullptr, 0), std::atoi(argv[3]), argv[4]);
    if (argc >= 5 && !std::strcmp(argv[1], "rulemat"))
        return cmdRuleMat(std::strtoull(argv[2], nullptr, 0), std::atoi(argv[3]), argv[4]);
    if (argc >= 4 && !std::strcmp(argv[1], "bench"))
        return cmdBench(std::strtoull(argv[2], nullptr, 0), std::atof(argv[3]));
    if (argc >= 2 && !std::strcmp(argv[1], "recsize")) { std::printf("%zu\n", sizeof(Record)); return 0; }
    std::fprintf(stderr, "usage: xqref trace SEED NGAMES OUT | tracebig SEED NREC OUT | validmat SEED NPOS OUT | rulemat SEED NPOS OUT | bench SEED SECONDS | recsize\n");
    return 2;
}
