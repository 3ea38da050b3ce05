// xq_rules.hip.h — Xiangqi rules engine for one wavefront = one board (gfx950, wave64).
//
// Semantics follow the reference rules engine exactly (Qervas/cn_chess_ai, src/chessboard.cpp — cited per function);
// the formulation is new: the 90-square board lives as bytes in a per-wave LDS slab plus 90-bit occupancy bitboards in
// scalar registers (wave ballots); move generation gives every lane one (piece, direction) slot, resolves chariot and
// cannon rays with bit scans, and a wave prefix-sum places the moves in the canonical order of
// ChessAI::getAllValidActions (chessai.cpp:347-368).
#pragma once

#include "xq_common.h"

namespace xq {

// per-wave LDS slab
struct __attribute__((aligned(16))) WaveSlab {
    float q[96];             // Q-values of outputs 0..89 for this game (selectAction reads q[action.to] only)
    uint16_t moves[kMaxMoves];
    uint8_t sq[96];          // piece code per square (90 used)
    int32_t misc[8];         // scratch: rank -> square table of move generation (16 bytes) + spare
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
        for (int c = fc + step; c != tc; c += step) n += at(sq, fr, c) != 0;
    } else {
        const int step = tr > fr ? 1 : -1;
        for (int r = fr + step; r != tr; r += step) n += at(sq, r, fc) != 0;
    }
    return n;
}

// The seven public per-piece validators (chessboard.h:50-56, chessboard.cpp:328-440) as written: geometry + occupancy only,
// no look at WHICH piece stands on `from` (the soldier rule reads its colour: anything not Red takes the Black branch),
// coordinates outside the board read as Empty (getPieceAt, :31-36).  `type` = PieceType 1..7.
__device__ inline bool piece_rule(const uint8_t* sq, int type, int fr, int fc, int tr, int tc) {
    const int dr = tr - fr, dc = tc - fc;
    const int adr = dr < 0 ? -dr : dr, adc = dc < 0 ? -dc : dc;
    switch (type) {
        case T_GENERAL:                                                     // :328-343
            return in_any_palace(fr, fc) && in_any_palace(tr, tc) && adr + adc == 1;
        case T_ADVISOR:                                                     // :346-353
            return in_any_palace(tr, tc) && adr == 1 && adc == 1;
        case T_ELEPHANT:                                                    // :355-367
            return adr == 2 && adc == 2 && ((fr < 5 && tr < 5) || (fr >= 5 && tr >= 5)) && at(sq, (fr + tr) / 2, (fc + tc) / 2) == 0;
        case T_HORSE:                                                       // :369-380 (truncating /2 picks the leg)
            return ((adr == 2 && adc == 1) || (adr == 1 && adc == 2)) && at(sq, fr + dr / 2, fc + dc / 2) == 0;
        case T_CHARIOT:                                                     // :382-397
            return (fr == tr || fc == tc) && count_between(sq, fr, fc, tr, tc) == 0;
        case T_CANNON: {                                                    // :399-421
            if (fr != tr && fc != tc) return false;
            const int n = count_between(sq, fr, fc, tr, tc);
            return at(sq, tr, tc) == 0 ? n == 0 : n == 1;
        }
        case T_SOLDIER: {                                                   // :423-440
            const int f = at(sq, fr, fc);
            if (f >= 1 && f <= 7) return fr < 5 ? (dr == 1 && adc == 0) : ((dr == 1 && adc == 0) || (dr == 0 && adc == 1));
            return fr >= 5 ? (dr == -1 && adc == 0) : ((dr == -1 && adc == 0) || (dr == 0 && adc == 1));
        }
        default: return false;
    }
}

// ChessBoard::isValidMove (chessboard.cpp:66-93) over the per-piece validators above.
// Pseudo-legal only: NO turn check, no check / flying-general rule.
__device__ inline bool is_valid_move(const uint8_t* sq, int fr, int fc, int tr, int tc) {
    if (!inside(fr, fc) || !inside(tr, tc)) return false;
    const int f = sq[fr * 9 + fc], t = sq[tr * 9 + tc];
    if (f == 0) return false;
    if (t != 0 && same_side(f, t)) return false;
    return piece_rule(sq, code_type(f), fr, fc, tr, tc);
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

// 90-bit board masks as two 64-bit words (bits 0..63 / 64..89)
struct Mask90 { unsigned long long lo, hi; };
__device__ __forceinline__ unsigned bits_at(const Mask90& m, int pos, unsigned mask) {   // bits [pos, pos+len) of the mask
    unsigned long long v;
    if (pos >= 64) v = m.hi >> (pos - 64);
    else v = (m.lo >> pos) | (pos ? (m.hi << (64 - pos)) : 0ull);
    return (unsigned)v & mask;
}

// One ray of a chariot / cannon resolved with bit operations on the line's occupancy (`line`: bit i = square i of the
// row or column, `pos`: the piece's index on the line, `len`: squares on the line, forward = towards higher indices).
//   empties : empty squares before the first piece (chariot and cannon both emit them, chessboard.cpp:205-216/228-232)
//   first   : index of the first piece or -1      (chariot: capturable iff enemy, then stop)
//   second  : index of the next piece behind it or -1 (cannon: capturable iff enemy, chessboard.cpp:235-240)
struct RayHit { int empties, first, second; };
__device__ __forceinline__ RayHit ray_scan(unsigned line, int pos, int len, bool forward) {
    RayHit h;
    if (forward) {
        const unsigned x = line >> (pos + 1);
        if (x == 0) { h.empties = len - 1 - pos; h.first = -1; h.second = -1; return h; }
        const int e = __ffs((int)x) - 1;
        h.empties = e; h.first = pos + 1 + e;
        const unsigned y = x >> (e + 1);
        h.second = y ? h.first + 1 + (__ffs((int)y) - 1) : -1;
    } else {
        const unsigned x = line & ((1u << pos) - 1u);
        if (x == 0) { h.empties = pos; h.first = -1; h.second = -1; return h; }
        const int f = 31 - __clz((int)x);
        h.empties = pos - 1 - f; h.first = f;
        const unsigned y = x & ((1u << f) - 1u);
        h.second = y ? 31 - __clz((int)y) : -1;
    }
    return h;
}

// ChessAI::getAllValidActions(player) (chessai.cpp:347-368): fills slab.moves in canonical order (from ascending, then
// the generator's direction order, chessboard.cpp:149-283), returns the count.
//
// Lane mapping: the side's pieces are ranked by square (ballot + popcount); in pass P lane l owns direction slot l&7 of
// piece rank 8P + (l>>3).  A slot yields at most one target — except chariot/cannon slots, which yield one ray: a run of
// consecutive squares plus (cannon) one jump capture.  Rays are resolved with bit scans on the row occupancy and on a
// column-major copy of it (second ballot), never by walking squares.  A wave prefix sum over the slots (pass 0 then
// pass 1 = piece order, slot order = generator order) places every move at its canonical index.
// Every lane of the wave must call this; slab.sq must be visible (wave_sync) before the call, and the caller
// synchronises again before reading slab.moves.  slab.misc[0..15] is used as scratch.
__device__ inline int gen_all_actions(WaveSlab& slab, int player) {
    const int lane = lane_id();
    const uint8_t* sq = slab.sq;
    const int p0 = sq[lane];
    const int p1 = lane < 26 ? sq[64 + lane] : 0;
    const bool black = player == C_BLACK;
    Mask90 occ, own, occT;
    occ.lo = __ballot(p0 != 0); occ.hi = __ballot(p1 != 0);
    const bool o0 = p0 != 0 && (p0 > 7) == black, o1 = p1 != 0 && (p1 > 7) == black;
    own.lo = __ballot(o0); own.hi = __ballot(o1);
    {   // column-major occupancy: bit c*10 + r
        const int t0 = lane, t1 = 64 + lane;
        const int s0 = (t0 % 10) * 9 + t0 / 10;
        const int s1 = t1 < 90 ? (t1 % 10) * 9 + t1 / 10 : 0;
        occT.lo = __ballot(sq[s0] != 0);
        occT.hi = __ballot(t1 < 90 && sq[s1] != 0);
    }
    // rank -> square table of the side's pieces
    uint8_t* rank_sq = reinterpret_cast<uint8_t*>(slab.misc);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int n_lo = __popcll(own.lo);
    const int n_own = n_lo + __popcll(own.hi);
    if (o0) { const int k = __popcll(own.lo & below); if (k < 16) rank_sq[k] = (uint8_t)lane; }
    if (o1) { const int k = n_lo + __popcll(own.hi & below); if (k < 16) rank_sq[k] = (uint8_t)(64 + lane); }
    wave_sync();

    int total = 0;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass * 8 >= n_own) break;                      // wave-uniform
        const int k = pass * 8 + (lane >> 3), slot = lane & 7;
        int n_run = 0, t0 = 0, delta = 0, extra = -1, from = 0;
        if (k < n_own && k < 16) {
            from = rank_sq[k];
            const int p = sq[from];
            const int row = from / 9, col = from - row * 9;
            const int type = p > 7 ? p - 7 : p;
            if (type == T_CHARIOT || type == T_CANNON) {
                if (slot < 4) {                            // right, left, +row, -row (chessboard.cpp:199 / :221)
                    const bool horiz = slot < 2, fwd = (slot & 1) == 0;
                    const unsigned line = horiz ? bits_at(occ, row * 9, 0x1FFu) : bits_at(occT, col * 10, 0x3FFu);
                    const int pos = horiz ? col : row, len = horiz ? 9 : 10;
                    const RayHit h = ray_scan(line, pos, len, fwd);
                    delta = horiz ? (fwd ? 1 : -1) : (fwd ? 9 : -9);
                    t0 = from + delta;
                    n_run = h.empties;
                    if (type == T_CHARIOT) {
                        if (h.first >= 0) {
                            const int bs = horiz ? row * 9 + h.first : h.first * 9 + col;
                            if ((sq[bs] > 7) != black) n_run += 1;         // enemy blocker: captured, it is the next square
                        }
                    } else if (h.second >= 0) {
                        const int bs = horiz ? row * 9 + h.second : h.second * 9 + col;
                        if ((sq[bs] > 7) != black) extra = bs;             // first piece behind the screen, iff enemy
                    }
                }
            } else {
                int dr = 0, dc = 0;
                bool ok = false;
                int guard = -1;                             // square that must be empty (horse leg / elephant eye)
                switch (type) {
                    case T_GENERAL:                         // (1,0) (-1,0) (0,1) (0,-1), chessboard.cpp:150
                        if (slot < 4) { dr = slot == 0 ? 1 : slot == 1 ? -1 : 0; dc = slot == 2 ? 1 : slot == 3 ? -1 : 0; ok = true; }
                        break;
                    case T_ADVISOR:                         // (1,1) (1,-1) (-1,1) (-1,-1), :163
                        if (slot < 4) { dr = slot < 2 ? 1 : -1; dc = (slot & 1) ? -1 : 1; ok = true; }
                        break;
                    case T_ELEPHANT:                        // (2,2) (2,-2) (-2,2) (-2,-2), :180
                        if (slot < 4) { dr = slot < 2 ? 2 : -2; dc = (slot & 1) ? -2 : 2; ok = true; }
                        break;
                    case T_HORSE:                           // (1,2)(1,-2)(-1,2)(-1,-2)(2,1)(2,-1)(-2,1)(-2,-1), :249
                        dr = slot < 4 ? ((slot & 2) ? -1 : 1) : ((slot & 2) ? -2 : 2);
                        dc = slot < 4 ? ((slot & 1) ? -2 : 2) : ((slot & 1) ? -1 : 1);
                        ok = true;
                        break;
                    case T_SOLDIER: {                       // forward, col-1, col+1, :265-283
                        const int fw = black ? -1 : 1;
                        const bool crossed = black ? row < 5 : row > 4;
                        if (slot == 0) { dr = fw; ok = true; }
                        else if (slot < 3 && crossed) { dc = slot == 1 ? -1 : 1; ok = true; }
                        break;
                    }
                    default: break;
                }
                const int nr = row + dr, nc = col + dc;
                ok = ok && inside(nr, nc);
                if (ok) {
                    switch (type) {
                        case T_GENERAL: ok = in_any_palace(row, col) && in_any_palace(nr, nc); break;          // :328-343
                        case T_ADVISOR: ok = in_own_palace(black ? C_BLACK : C_RED, nr, nc); break;            // :170-172
                        case T_ELEPHANT:                                                                      // :189-192, :359
                            ok = (black ? nr >= 5 : nr <= 4) && ((row < 5) == (nr < 5));
                            guard = (row + dr / 2) * 9 + col + dc / 2;
                            break;
                        case T_HORSE: guard = (row + dr / 2) * 9 + col + dc / 2; break;                        // :254-258
                        default: break;
                    }
                }
                if (ok && guard >= 0) ok = sq[guard] == 0;
                if (ok) {
                    const int t = sq[nr * 9 + nc];
                    ok = t == 0 || (t > 7) != black;                                                           // :78-80
                }
                if (ok) { n_run = 1; t0 = nr * 9 + nc; }
            }
        }
        const int cnt = n_run + (extra >= 0 ? 1 : 0);
        const int inc = wave_inclusive_scan(cnt);
        int off = total + inc - cnt;
        const int base = from * 90;
        for (int i = 0; i < n_run; ++i, ++off)
            if (off < kMaxMoves) slab.moves[off] = (uint16_t)(base + t0 + i * delta);
        if (extra >= 0 && off < kMaxMoves) slab.moves[off] = (uint16_t)(base + extra);
        total += __shfl(inc, 63, 64);
    }
    return total < kMaxMoves ? total : kMaxMoves;
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
