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

// Cross-lane sums / maxima on the DPP network (row_shr 1,2,4,8 inside each row of 16 lanes, then row_bcast 15 / 31 across the
// rows): 6 VALU instructions per scan, against 6 x (ds_bpermute + select + add) for the shuffle form.  Lanes whose DPP source
// lies outside the row (or the row mask) receive the `old` operand: the identity of the operation.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_int(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_inclusive_scan(int v) {           // inclusive prefix sum over the 64 lanes
    v += dpp_int<0x111, 0xf>(0, v);      // row_shr:1
    v += dpp_int<0x112, 0xf>(0, v);      // row_shr:2
    v += dpp_int<0x114, 0xf>(0, v);      // row_shr:4
    v += dpp_int<0x118, 0xf>(0, v);      // row_shr:8
    v += dpp_int<0x142, 0xa>(0, v);      // row_bcast:15 into rows 1 and 3
    v += dpp_int<0x143, 0xc>(0, v);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {                      // total, wave-uniform (lane 63 of the scan)
    return __builtin_amdgcn_readlane(wave_inclusive_scan(v), 63);
}
// Orders LDS traffic between the lanes of ONE wave (each wave owns its slab; the LDS pipeline is in-order per wave):
// a compiler + counter fence, no s_barrier, so it is legal inside wave-divergent control flow.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float wave_max(float v) {                  // maximum over the 64 lanes, wave-uniform
    const int ninf = __float_as_int(-__builtin_inff());
    auto step = [&](int t) { v = fmaxf(v, __int_as_float(t)); };
    step(dpp_int<0x111, 0xf>(ninf, __float_as_int(v)));
    step(dpp_int<0x112, 0xf>(ninf, __float_as_int(v)));
    step(dpp_int<0x114, 0xf>(ninf, __float_as_int(v)));
    step(dpp_int<0x118, 0xf>(ninf, __float_as_int(v)));
    step(dpp_int<0x142, 0xa>(ninf, __float_as_int(v)));
    step(dpp_int<0x143, 0xc>(ninf, __float_as_int(v)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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
    // branch-free: both halves are formed and one is selected ((hi << 1) << (63 - pos) is hi << (64 - pos), and 0 for pos = 0)
    const unsigned long long a = (m.lo >> (pos & 63)) | ((m.hi << 1) << (63 - (pos & 63)));
    const unsigned long long b = m.hi >> (pos & 63);
    return (unsigned)(pos >= 64 ? b : a) & mask;
}

// One ray of a chariot / cannon resolved with bit operations on the line's occupancy (`line`: bit i = square i of the
// row or column, `pos`: the piece's index on the line, `len`: squares on the line, forward = towards higher indices).
//   empties : empty squares before the first piece (chariot and cannon both emit them, chessboard.cpp:205-216/228-232)
//   first   : index of the first piece or -1      (chariot: capturable iff enemy, then stop)
//   second  : index of the next piece behind it or -1 (cannon: capturable iff enemy, chessboard.cpp:235-240)
// A backward ray is the forward ray of the mirrored line (v_bfrev), so there is ONE scan and no divergent branch; the
// indices are mirrored back at the end.
struct RayHit { int empties, first, second; };
__device__ __forceinline__ RayHit ray_scan(unsigned line, int pos, int len, bool forward) {
    const unsigned l = forward ? line : (__brev(line) >> (32 - len));
    const int p = forward ? pos : len - 1 - pos;
    const unsigned x = l >> (p + 1);
    const int e = __ffs((int)x) - 1;                       // -1 when the ray is empty
    const unsigned y = x >> ((e + 1) & 31);                // e = -1: y = x = 0
    const int e2 = __ffs((int)y) - 1;
    int first = x ? p + 1 + e : -1;
    int second = (x && y) ? p + 2 + e + e2 : -1;
    RayHit h;
    h.empties = x ? e : len - 1 - p;
    h.first = (first >= 0 && !forward) ? len - 1 - first : first;
    h.second = (second >= 0 && !forward) ? len - 1 - second : second;
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
        const bool active = k < n_own;                     // n_own <= 16
        // Every lane evaluates its (piece, direction) slot with selects instead of a switch over the piece type: the seven
        // case bodies of a switch all execute anyway (a wave mixes piece types) and each costs an exec-mask save / restore /
        // branch on the scalar unit.  Inactive lanes compute on piece 0 and are masked at the end.
        const int from = rank_sq[active ? k : 0];
        const int p = sq[from];
        const int row = (from * 57) >> 9, col = from - row * 9;               // from / 9 for from < 96
        const int type = p > 7 ? p - 7 : p;
        const bool isG = type == T_GENERAL, isA = type == T_ADVISOR, isE = type == T_ELEPHANT, isH = type == T_HORSE;
        const bool isS = type == T_SOLDIER, isRay = type == T_CHARIOT || type == T_CANNON;
        const bool lt4 = slot < 4, lt2 = slot < 2;
        const int sg1 = (slot & 1) ? -1 : 1, sg2 = (slot & 2) ? -1 : 1;
        const int fw = black ? -1 : 1;
        const bool crossed = black ? row < 5 : row > 4;
        // direction tables of chessboard.cpp:150 / :163 / :180 / :249 / :265-283, in generator order
        const int mag = isE ? 2 : 1;
        int dr = lt2 ? mag : -mag, dc = sg1 * mag;                             // advisor / elephant: (m,m) (m,-m) (-m,m) (-m,-m)
        if (isG) { dr = lt2 ? sg1 : 0; dc = lt2 ? 0 : sg1; }                   // (1,0) (-1,0) (0,1) (0,-1)
        if (isH) { dr = lt4 ? sg2 : 2 * sg2; dc = lt4 ? 2 * sg1 : sg1; }       // (1,2)(1,-2)(-1,2)(-1,-2)(2,1)(2,-1)(-2,1)(-2,-1)
        if (isS) { dr = slot == 0 ? fw : 0; dc = slot == 1 ? -1 : (slot == 2 ? 1 : 0); }   // forward, col-1, col+1
        const bool slot_ok = isH || (isS ? (slot == 0 || (slot < 3 && crossed)) : ((isG || isA || isE) && lt4));
        const int nr = row + dr, nc = col + dc;
        const bool in = inside(nr, nc);
        const bool to_pal_col = nc >= 3 && nc <= 5;
        const bool to_any_pal = to_pal_col && (nr <= 2 || nr >= 7);
        const bool to_own_pal = to_pal_col && (black ? nr >= 7 : nr <= 2);
        const bool rule = isG ? (in_any_palace(row, col) && to_any_pal)                          // :328-343
                        : isA ? to_own_pal                                                       // :170-172
                        : isE ? ((black ? nr >= 5 : nr <= 4) && ((row < 5) == (nr < 5)))         // :189-192, :359
                        : true;
        bool ok = active && slot_ok && in && rule;
        // the square that must be empty (elephant eye, horse leg: truncating /2 as upstream) and the target; a slot that is
        // already out reads `from` for both, which holds an own piece and so fails either test
        const int gsq = (ok && (isE || isH)) ? (row + dr / 2) * 9 + col + dc / 2 : -1;
        const int tsq = ok ? nr * 9 + nc : from;
        const int gp = sq[gsq >= 0 ? gsq : from], tp = sq[tsq];
        ok = ok && (gsq < 0 || gp == 0) && (tp == 0 || (tp > 7) != black);                       // :78-80
        int n_run = ok ? 1 : 0, t0 = tsq, delta = 0, extra = -1;
        if (active && isRay && lt4) {                          // right, left, +row, -row (chessboard.cpp:199 / :221)
            const bool horiz = lt2, fwd = (slot & 1) == 0;
            const unsigned line = horiz ? bits_at(occ, row * 9, 0x1FFu) : bits_at(occT, col * 10, 0x3FFu);
            const int pos = horiz ? col : row, len = horiz ? 9 : 10;
            const RayHit h = ray_scan(line, pos, len, fwd);
            delta = horiz ? (fwd ? 1 : -1) : (fwd ? 9 : -9);
            t0 = from + delta;
            n_run = h.empties;
            const int hit = type == T_CHARIOT ? h.first : h.second;   // chariot: the blocker itself; cannon: the piece behind the screen
            if (hit >= 0) {
                const int bs = horiz ? row * 9 + hit : hit * 9 + col;
                if ((sq[bs] > 7) != black) {                   // enemy: capturable
                    if (type == T_CHARIOT) n_run += 1;         // it is the next square of the run
                    else extra = bs;
                }
            }
        }
        const int cnt = n_run + (extra >= 0 ? 1 : 0);
        const int inc = wave_inclusive_scan(cnt);
        const int off = total + inc - cnt;
        // The list holds kMaxMoves entries (a real position has at most ~85 moves; an arbitrary xq_env_set_state board may
        // have more): the run is clipped ONCE, so the store loop carries no bound test — with one, the compiler unrolls it
        // eight-fold into predicated stores and the exec-mask bookkeeping costs more than the stores.
        const int room = kMaxMoves - off;
        const int n_put = n_run < room ? n_run : (room > 0 ? room : 0);
        uint16_t* dst = slab.moves + off;
        int code = from * 90 + t0;
#pragma clang loop unroll(disable)
        for (int i = 0; i < n_put; ++i, code += delta) dst[i] = (uint16_t)code;
        if (extra >= 0 && n_run < room) dst[n_run] = (uint16_t)(from * 90 + extra);
        total += __builtin_amdgcn_readlane(inc, 63);
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
