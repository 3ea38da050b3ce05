// xq_rules.cuh — Xiangqi rules engine for one wavefront = one board (gfx950, wave64).
//
// Semantics follow the reference rules engine exactly (Qervas/cn_chess_ai, src/chessboard.cpp — cited per function);
// the formulation is new: the 90-square board lives as bytes in a per-wave LDS slab, every lane owns the squares
// `lane` and `lane+64`, each lane generates the ordered move list of its own pieces into two packed registers, and a
// wave prefix-sum places them in the canonical order of ChessAI::getAllValidActions (chessai.cpp:347-368).
#pragma once

#include "xq_common.h"

namespace xq {

// per-wave LDS slab
struct __attribute__((aligned(16))) WaveSlab {
    float q[96];             // Q-values of outputs 0..89 for this game (selectAction reads q[action.to] only)
    uint16_t moves[kMaxMoves];
    uint8_t sq[96];          // piece code per square (90 used)
    int32_t misc[8];
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ int at(const uint8_t* sq, int r, int c) {   // getPieceAt: Empty outside (chessboard.cpp:31-36)
    return ((unsigned)r < 10u && (unsigned)c < 9u) ? (int)sq[r * 9 + c] : 0;
}
__device__ __forceinline__ bool inside(int r, int c) { return (unsigned)r < 10u && (unsigned)c < 9u; }   // :323-325
__device__ __forceinline__ bool in_any_palace(int r, int c) {          // :330-333
    return c >= 3 && c <= 5 && ((r >= 0 && r <= 2) || (r >= 7 && r <= 9));
}
__device__ __forceinline__ bool in_own_palace(int color, int r, int c) {   // chessboard.h:65-71
    return c >= 3 && c <= 5 && (color == C_RED ? (r >= 0 && r <= 2) : (r >= 7 && r <= 9));
}
__device__ __forceinline__ bool same_side(int a, int b) { return (a > 7) == (b > 7); }

__device__ __forceinline__ int count_between(const uint8_t* sq, int fr, int fc, int tr, int tc) {
    int n = 0;
    if (fr == tr) {
        const int step = tc > fc ? 1 : -1;
        for (int c = fc + step; c != tc; c += step) n += sq[fr * 9 + c] != 0;
    } else {
        const int step = tr > fr ? 1 : -1;
        for (int r = fr + step; r != tr; r += step) n += sq[r * 9 + fc] != 0;
    }
    return n;
}

// ChessBoard::isValidMove (chessboard.cpp:66-93) with the per-piece validators (:328-440).
// Pseudo-legal only: NO turn check, no check / flying-general rule.
__device__ inline bool is_valid_move(const uint8_t* sq, int fr, int fc, int tr, int tc) {
    if (!inside(fr, fc) || !inside(tr, tc)) return false;
    const int f = sq[fr * 9 + fc], t = sq[tr * 9 + tc];
    if (f == 0) return false;
    if (t != 0 && same_side(f, t)) return false;
    const int dr = tr - fr, dc = tc - fc;
    const int adr = dr < 0 ? -dr : dr, adc = dc < 0 ? -dc : dc;
    switch (code_type(f)) {
        case T_GENERAL:                                                     // :328-343
            return in_any_palace(fr, fc) && in_any_palace(tr, tc) && adr + adc == 1;
        case T_ADVISOR:                                                     // :346-353
            return in_any_palace(tr, tc) && adr == 1 && adc == 1;
        case T_ELEPHANT:                                                    // :355-367
            return adr == 2 && adc == 2 && ((fr < 5) == (tr < 5)) && at(sq, (fr + tr) / 2, (fc + tc) / 2) == 0;
        case T_HORSE:                                                       // :369-380 (truncating /2 picks the leg)
            return ((adr == 2 && adc == 1) || (adr == 1 && adc == 2)) && at(sq, fr + dr / 2, fc + dc / 2) == 0;
        case T_CHARIOT:                                                     // :382-397
            return (fr == tr || fc == tc) && count_between(sq, fr, fc, tr, tc) == 0;
        case T_CANNON: {                                                    // :399-421
            if (fr != tr && fc != tc) return false;
            const int n = count_between(sq, fr, fc, tr, tc);
            return t == 0 ? n == 0 : n == 1;
        }
        case T_SOLDIER:                                                     // :423-440
            if (f <= 7) return fr < 5 ? (dr == 1 && adc == 0) : ((dr == 1 && adc == 0) || (dr == 0 && adc == 1));
            return fr >= 5 ? (dr == -1 && adc == 0) : ((dr == -1 && adc == 0) || (dr == 0 && adc == 1));
        default: return false;
    }
}

// up to 17 targets of one piece, 7 bits each, in generation order
struct PieceMoves {
    unsigned long long lo = 0, hi = 0;
    int n = 0;
    __device__ __forceinline__ void emit(int to) {
        if (n < 9) lo |= (unsigned long long)to << (7 * n);
        else hi |= (unsigned long long)to << (7 * (n - 9));
        ++n;
    }
    __device__ __forceinline__ int get(int k) const {
        return (int)((k < 9 ? lo >> (7 * k) : hi >> (7 * (k - 9))) & 127ull);
    }
};

// ChessBoard::getValidMoves(row,col) (chessboard.cpp:112-147) — generator order of :149-283.
__device__ inline void gen_piece_moves(const uint8_t* sq, int s, PieceMoves& out) {
    const int p = sq[s];
    const int row = s / 9, col = s - row * 9;
    const int color = p > 7 ? C_BLACK : C_RED;
    switch (code_type(p)) {
        case T_GENERAL: {                                                   // :149-160  (1,0) (-1,0) (0,1) (0,-1)
            const int d[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int nr = row + d[k][0], nc = col + d[k][1];
                if (inside(nr, nc) && is_valid_move(sq, row, col, nr, nc)) out.emit(nr * 9 + nc);
            }
            break;
        }
        case T_ADVISOR: {                                                   // :162-177  own palace only
            const int d[4][2] = {{1, 1}, {1, -1}, {-1, 1}, {-1, -1}};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int nr = row + d[k][0], nc = col + d[k][1];
                if (in_own_palace(color, nr, nc) && is_valid_move(sq, row, col, nr, nc)) out.emit(nr * 9 + nc);
            }
            break;
        }
        case T_ELEPHANT: {                                                  // :179-196  own side, eye empty
            const int d[4][2] = {{2, 2}, {2, -2}, {-2, 2}, {-2, -2}};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int nr = row + d[k][0], nc = col + d[k][1];
                const bool own_side = color == C_RED ? (nr >= 0 && nr <= 4) : (nr >= 5 && nr <= 9);
                if (inside(nr, nc) && own_side && at(sq, row + d[k][0] / 2, col + d[k][1] / 2) == 0 &&
                    is_valid_move(sq, row, col, nr, nc))
                    out.emit(nr * 9 + nc);
            }
            break;
        }
        case T_HORSE: {                                                     // :248-263
            const int d[8][2] = {{1, 2}, {1, -2}, {-1, 2}, {-1, -2}, {2, 1}, {2, -1}, {-2, 1}, {-2, -1}};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int nr = row + d[k][0], nc = col + d[k][1];
                if (inside(nr, nc) && at(sq, row + d[k][0] / 2, col + d[k][1] / 2) == 0 &&
                    is_valid_move(sq, row, col, nr, nc))
                    out.emit(nr * 9 + nc);
            }
            break;
        }
        case T_CHARIOT: {                                                   // :198-218  right, left, +row, -row
            const int d[4][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int nr = row + d[k][0], nc = col + d[k][1];
                while (inside(nr, nc)) {
                    const int t = sq[nr * 9 + nc];
                    if (t != 0 && same_side(p, t)) break;      // isValidMove fails -> break
                    out.emit(nr * 9 + nc);
                    if (t != 0) break;                         // stop at the first piece
                    nr += d[k][0]; nc += d[k][1];
                }
            }
            break;
        }
        case T_CANNON: {                                                    // :220-246
            const int d[4][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int nr = row + d[k][0], nc = col + d[k][1];
                bool screen = false;
                while (inside(nr, nc)) {
                    const int t = sq[nr * 9 + nc];
                    if (!screen) {
                        if (t == 0) out.emit(nr * 9 + nc); else screen = true;
                    } else if (t != 0) {
                        // first piece behind the screen: capturable iff enemy; anything further has >= 2 screens
                        if (!same_side(p, t)) out.emit(nr * 9 + nc);
                        break;
                    }
                    nr += d[k][0]; nc += d[k][1];
                }
            }
            break;
        }
        case T_SOLDIER: {                                                   // :265-283  forward, col-1, col+1
            const int nr = row + (color == C_RED ? 1 : -1);
            if (inside(nr, col) && is_valid_move(sq, row, col, nr, col)) out.emit(nr * 9 + col);
            if ((color == C_RED && row > 4) || (color == C_BLACK && row < 5)) {
                if (inside(row, col - 1) && is_valid_move(sq, row, col, row, col - 1)) out.emit(row * 9 + col - 1);
                if (inside(row, col + 1) && is_valid_move(sq, row, col, row, col + 1)) out.emit(row * 9 + col + 1);
            }
            break;
        }
        default: break;
    }
}

// inclusive wave prefix sum over 64 lanes
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Orders LDS traffic between the lanes of ONE wave (each wave owns its slab; the LDS pipeline is in-order per wave):
// a compiler + counter fence, no s_barrier, so it is legal inside wave-divergent control flow.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Unpack 12 words (lanes 0..11) into the slab's byte board.  Caller synchronises afterwards.
__device__ __forceinline__ void unpack_to_slab(uint32_t word, uint8_t* sq) {
    const int lane = lane_id();
    if (lane < kBoardWords) {
        const uint32_t lo = word & 0xFFFFu, hi = word >> 16;
        // spread 4 nibbles to 4 bytes
        auto spread = [](uint32_t x) -> uint32_t {
            return (x & 0xFu) | ((x & 0xF0u) << 4) | ((x & 0xF00u) << 8) | ((x & 0xF000u) << 12);
        };
        reinterpret_cast<uint32_t*>(sq)[lane * 2] = spread(lo);
        reinterpret_cast<uint32_t*>(sq)[lane * 2 + 1] = spread(hi);
    }
}
__device__ __forceinline__ uint32_t pack_from_slab(const uint8_t* sq) {   // valid on lanes 0..11
    const int lane = lane_id();
    uint32_t v = 0;
    if (lane < kBoardWords) {
        const uint32_t a = reinterpret_cast<const uint32_t*>(sq)[lane * 2];
        const uint32_t b = reinterpret_cast<const uint32_t*>(sq)[lane * 2 + 1];
        auto squeeze = [](uint32_t x) -> uint32_t {
            return (x & 0xFu) | ((x >> 4) & 0xF0u) | ((x >> 8) & 0xF00u) | ((x >> 12) & 0xF000u);
        };
        v = squeeze(a) | (squeeze(b) << 16);
    }
    return v;
}

// ChessAI::getAllValidActions(player) (chessai.cpp:347-368): fills slab.moves in canonical order, returns the count.
// Every lane of the wave must call this; slab.sq must be visible (synchronised) before the call.  The caller
// synchronises again before reading slab.moves.
__device__ inline int gen_all_actions(WaveSlab& slab, int player) {
    const int lane = lane_id();
    const uint8_t* sq = slab.sq;
    const int p0 = sq[lane];
    const int p1 = lane < 26 ? sq[64 + lane] : 0;
    PieceMoves ma, mb;
    if (p0 != 0 && (p0 > 7) == (player == C_BLACK)) gen_piece_moves(sq, lane, ma);
    if (p1 != 0 && (p1 > 7) == (player == C_BLACK)) gen_piece_moves(sq, 64 + lane, mb);
    const int inc_a = wave_inclusive_scan(ma.n);
    const int tot_a = __shfl(inc_a, 63, 64);
    const int inc_b = wave_inclusive_scan(mb.n);
    const int tot_b = __shfl(inc_b, 63, 64);
    int off = inc_a - ma.n;
    for (int k = 0; k < ma.n; ++k) {
        if (off + k < kMaxMoves) slab.moves[off + k] = (uint16_t)(lane * 90 + ma.get(k));
    }
    off = tot_a + inc_b - mb.n;
    for (int k = 0; k < mb.n; ++k) {
        if (off + k < kMaxMoves) slab.moves[off + k] = (uint16_t)((64 + lane) * 90 + mb.get(k));
    }
    const int n = tot_a + tot_b;
    return n < kMaxMoves ? n : kMaxMoves;
}

__device__ __forceinline__ int piece_value(int code) {                   // PieceScore, chessboard.h:23-31
    const int t = code_type(code);
    // G1000 A20 E20 H40 R90 C45 S10
    return t == T_GENERAL ? 1000 : t == T_ADVISOR ? 20 : t == T_ELEPHANT ? 20 : t == T_HORSE ? 40
         : t == T_CHARIOT ? 90 : t == T_CANNON ? 45 : t == T_SOLDIER ? 10 : 0;
}

// ChessAI::evaluateBoard(color, moveCount) (chessai.cpp:311-345): material(own) - material(enemy), then
// `score -= moveCount * 0.1` evaluated in double and truncated toward zero.  The multiply and the subtract must
// stay two correctly-rounded fp64 operations (no FMA contraction) to reproduce the host result bit for bit.
__device__ inline int evaluate_board_wave(const uint8_t* sq, int color, int move_count) {
    const int lane = lane_id();
    const int p0 = sq[lane];
    const int p1 = lane < 26 ? sq[64 + lane] : 0;
    int v = 0;
    if (p0) v += ((p0 > 7) == (color == C_BLACK)) ? piece_value(p0) : -piece_value(p0);
    if (p1) v += ((p1 > 7) == (color == C_BLACK)) ? piece_value(p1) : -piece_value(p1);
    const int score = wave_sum(v);
    {
#pragma clang fp contract(off)
        const double pen = __dmul_rn((double)move_count, 0.1);
        const double x = __dsub_rn((double)score, pen);
        return (int)x;
    }
}

struct BoardStatus {
    bool red_general, black_general;
    int first_general_color;   // getWinner(): colour of the first general in index order, C_NONE if none
};
__device__ inline BoardStatus board_status_wave(const uint8_t* sq) {     // chessboard.cpp:286-320
    const int lane = lane_id();
    const int p0 = sq[lane];
    const int p1 = lane < 26 ? sq[64 + lane] : 0;
    const unsigned long long ra = __ballot(p0 == 1), rb = __ballot(p1 == 1);
    const unsigned long long ba = __ballot(p0 == 8), bb = __ballot(p1 == 8);
    BoardStatus st;
    st.red_general = (ra | rb) != 0;
    st.black_general = (ba | bb) != 0;
    const unsigned long long ga = ra | ba, gb = rb | bb;
    if (ga) st.first_general_color = ((ra >> (__ffsll((long long)ga) - 1)) & 1ull) ? C_RED : C_BLACK;
    else if (gb) st.first_general_color = ((rb >> (__ffsll((long long)gb) - 1)) & 1ull) ? C_RED : C_BLACK;
    else st.first_general_color = C_NONE;
    return st;
}

}  // namespace xq
