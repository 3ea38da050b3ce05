/* TEST INFRASTRUCTURE — CPU oracle (see xq_oracle.h for scope and pinning status).
 *
 * Structure-faithful restatement: every function follows the cited reference lines, including the
 * generate-then-validate shape of the move generator and the as-written (bug-compatible) backprop.
 * All file:line citations are into /root/reference (Qervas/cn_chess_ai @ 2024-10-20).
 */
#include "xq_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Rules engine — src/chessboard.cpp
 * ---------------------------------------------------------------------------------------------- */
enum { T_EMPTY = 0, T_GENERAL, T_ADVISOR, T_ELEPHANT, T_HORSE, T_CHARIOT, T_CANNON, T_SOLDIER };

static int code_type(int c)  { return c == 0 ? T_EMPTY : (c > 7 ? c - 7 : c); }
static int code_color(int c) { return c == 0 ? XQO_NONE : (c > 7 ? XQO_BLACK : XQO_RED); }

static int inside(int r, int c) { return r >= 0 && r < 10 && c >= 0 && c < 9; }   /* chessboard.cpp:323-325 */
static int piece_at(const xqo_board* b, int r, int c) {                            /* :31-36 — Empty outside */
    return inside(r, c) ? b->sq[r * 9 + c] : 0;
}
static int in_red_palace(int r, int c)   { return r >= 0 && r <= 2 && c >= 3 && c <= 5; }  /* chessboard.h:65-67 */
static int in_black_palace(int r, int c) { return r >= 7 && r <= 9 && c >= 3 && c <= 5; }  /* chessboard.h:69-71 */
static int in_own_side(int color, int r) { return color == XQO_RED ? (r >= 0 && r <= 4) : (r >= 5 && r <= 9); } /* h:73-75 */

static const int PIECE_SCORE[8] = { 0, 1000, 20, 20, 40, 90, 45, 10 };  /* chessboard.h:23-31, cpp:443-454 */

void xqo_reset(xqo_board* b) {                                         /* :95-102 + :8-29 */
    static const uint8_t back[9] = { T_CHARIOT, T_HORSE, T_ELEPHANT, T_ADVISOR, T_GENERAL,
                                     T_ADVISOR, T_ELEPHANT, T_HORSE, T_CHARIOT };
    memset(b, 0, sizeof *b);
    for (int c = 0; c < 9; ++c) {
        b->sq[0 * 9 + c] = back[c];             /* red back rank, row 0 */
        b->sq[9 * 9 + c] = (uint8_t)(back[c] + 7);
    }
    b->sq[2 * 9 + 1] = b->sq[2 * 9 + 7] = T_CANNON;
    b->sq[7 * 9 + 1] = b->sq[7 * 9 + 7] = T_CANNON + 7;
    for (int c = 0; c < 9; c += 2) {
        b->sq[3 * 9 + c] = T_SOLDIER;
        b->sq[6 * 9 + c] = T_SOLDIER + 7;
    }
    b->moveCount = 0;
    b->currentPlayer = XQO_RED;
    b->redScore = b->blackScore = 0;
}

static int valid_general(int fr, int fc, int tr, int tc) {             /* :328-343 */
    int from_in = in_red_palace(fr, fc) || in_black_palace(fr, fc);
    int to_in = in_red_palace(tr, tc) || in_black_palace(tr, tc);
    if (!from_in || !to_in) return 0;
    return abs(tr - fr) + abs(tc - fc) == 1;
}
static int valid_advisor(int fr, int fc, int tr, int tc) {             /* :346-353 */
    int in_palace = in_red_palace(tr, tc) || in_black_palace(tr, tc);
    return in_palace && abs(tr - fr) == 1 && abs(tc - fc) == 1;
}
static int valid_elephant(const xqo_board* b, int fr, int fc, int tr, int tc) {   /* :355-367 */
    int rd = abs(tr - fr), cd = abs(tc - fc);
    int no_cross = (fr < 5 && tr < 5) || (fr >= 5 && tr >= 5);
    int mr = (fr + tr) / 2, mc = (fc + tc) / 2;
    int clear = piece_at(b, mr, mc) == 0;
    return rd == 2 && cd == 2 && no_cross && clear;
}
static int valid_horse(const xqo_board* b, int fr, int fc, int tr, int tc) {      /* :369-380 */
    int rd = abs(tr - fr), cd = abs(tc - fc);
    if ((rd == 2 && cd == 1) || (rd == 1 && cd == 2)) {
        int br = fr + (tr - fr) / 2, bc = fc + (tc - fc) / 2;   /* C truncating division, as upstream */
        return piece_at(b, br, bc) == 0;
    }
    return 0;
}
static int count_between(const xqo_board* b, int fr, int fc, int tr, int tc) {
    int same_row = (fr == tr);
    int step = same_row ? (tc > fc ? 1 : -1) : (tr > fr ? 1 : -1);
    int start = same_row ? fc : fr, end = same_row ? tc : tr, n = 0;
    for (int i = start + step; i != end; i += step)
        if (piece_at(b, same_row ? fr : i, same_row ? i : fc) != 0) ++n;
    return n;
}
static int valid_chariot(const xqo_board* b, int fr, int fc, int tr, int tc) {    /* :382-397 */
    if (fr != tr && fc != tc) return 0;
    /* from==to: the loop body never runs upstream (start+step != end fails only after wrap) — callers never pass it:
     * isValidMove rejects own-colour targets first.  Guard to keep this total. */
    if (fr == tr && fc == tc) return 1;
    return count_between(b, fr, fc, tr, tc) == 0;
}
static int valid_cannon(const xqo_board* b, int fr, int fc, int tr, int tc) {     /* :399-421 */
    if (fr != tr && fc != tc) return 0;
    if (fr == tr && fc == tc) return 0; /* unreachable through isValidMove (own-colour target) */
    int n = count_between(b, fr, fc, tr, tc);
    return piece_at(b, tr, tc) == 0 ? n == 0 : n == 1;
}
static int valid_soldier(const xqo_board* b, int fr, int fc, int tr, int tc) {    /* :423-440 */
    int rd = tr - fr, cd = abs(tc - fc);
    if (code_color(piece_at(b, fr, fc)) == XQO_RED) {
        if (fr < 5) return rd == 1 && cd == 0;
        return (rd == 1 && cd == 0) || (rd == 0 && cd == 1);
    }
    if (fr >= 5) return rd == -1 && cd == 0;
    return (rd == -1 && cd == 0) || (rd == 0 && cd == 1);
}

int xqo_is_valid_move(const xqo_board* b, int fr, int fc, int tr, int tc) {       /* :66-93 */
    if (!inside(fr, fc) || !inside(tr, tc)) return 0;
    int from = b->sq[fr * 9 + fc], to = b->sq[tr * 9 + tc];
    if (from == 0) return 0;
    if (code_color(from) == code_color(to) && to != 0) return 0;   /* no own-colour capture; NO turn check */
    switch (code_type(from)) {
        case T_GENERAL:  return valid_general(fr, fc, tr, tc);
        case T_ADVISOR:  return valid_advisor(fr, fc, tr, tc);
        case T_ELEPHANT: return valid_elephant(b, fr, fc, tr, tc);
        case T_HORSE:    return valid_horse(b, fr, fc, tr, tc);
        case T_CHARIOT:  return valid_chariot(b, fr, fc, tr, tc);
        case T_CANNON:   return valid_cannon(b, fr, fc, tr, tc);
        case T_SOLDIER:  return valid_soldier(b, fr, fc, tr, tc);
        default: return 0;
    }
}

/* the seven PUBLIC validators (chessboard.h:50-56) by PieceType 1..7, for any coordinates; -1 where upstream's path loop
 * overflows (from == to on a chariot / cannon, chessboard.cpp:390/:410) */
int xqo_piece_rule(const xqo_board* b, int type, int fr, int fc, int tr, int tc) {
    switch (type) {
        case T_GENERAL:  return valid_general(fr, fc, tr, tc);
        case T_ADVISOR:  return valid_advisor(fr, fc, tr, tc);
        case T_ELEPHANT: return valid_elephant(b, fr, fc, tr, tc);
        case T_HORSE:    return valid_horse(b, fr, fc, tr, tc);
        case T_CHARIOT:  return (fr == tr && fc == tc) ? -1 : valid_chariot(b, fr, fc, tr, tc);
        case T_CANNON:   return (fr == tr && fc == tc) ? -1 : valid_cannon(b, fr, fc, tr, tc);
        case T_SOLDIER:  return valid_soldier(b, fr, fc, tr, tc);
        default: return 0;
    }
}

#define EMIT(r, c) (out_sq[n++] = (r) * 9 + (c))

int xqo_get_valid_moves(const xqo_board* b, int row, int col, int* out_sq) {      /* :112-147 */
    int n = 0;
    int piece = piece_at(b, row, col);
    if (piece == 0) return 0;
    int color = code_color(piece);
    switch (code_type(piece)) {
    case T_GENERAL: {                                                              /* :149-160 */
        static const int d[4][2] = { {1, 0}, {-1, 0}, {0, 1}, {0, -1} };
        for (int k = 0; k < 4; ++k) {
            int nr = row + d[k][0], nc = col + d[k][1];
            if (inside(nr, nc) && xqo_is_valid_move(b, row, col, nr, nc)) EMIT(nr, nc);
        }
        break;
    }
    case T_ADVISOR: {                                                              /* :162-177 */
        static const int d[4][2] = { {1, 1}, {1, -1}, {-1, 1}, {-1, -1} };
        for (int k = 0; k < 4; ++k) {
            int nr = row + d[k][0], nc = col + d[k][1];
            if (inside(nr, nc) &&
                ((color == XQO_RED && in_red_palace(nr, nc)) || (color == XQO_BLACK && in_black_palace(nr, nc))) &&
                xqo_is_valid_move(b, row, col, nr, nc)) EMIT(nr, nc);
        }
        break;
    }
    case T_ELEPHANT: {                                                             /* :179-196 */
        static const int d[4][2] = { {2, 2}, {2, -2}, {-2, 2}, {-2, -2} };
        for (int k = 0; k < 4; ++k) {
            int nr = row + d[k][0], nc = col + d[k][1];
            int mr = row + d[k][0] / 2, mc = col + d[k][1] / 2;
            if (inside(nr, nc) && in_own_side(color, nr) && piece_at(b, mr, mc) == 0 &&
                xqo_is_valid_move(b, row, col, nr, nc)) EMIT(nr, nc);
        }
        break;
    }
    case T_CHARIOT: {                                                              /* :198-218 */
        static const int d[4][2] = { {0, 1}, {0, -1}, {1, 0}, {-1, 0} };
        for (int k = 0; k < 4; ++k) {
            int nr = row + d[k][0], nc = col + d[k][1];
            while (inside(nr, nc)) {
                if (xqo_is_valid_move(b, row, col, nr, nc)) {
                    EMIT(nr, nc);
                    if (piece_at(b, nr, nc) != 0) break;
                } else break;
                nr += d[k][0]; nc += d[k][1];
            }
        }
        break;
    }
    case T_CANNON: {                                                               /* :220-246 */
        static const int d[4][2] = { {0, 1}, {0, -1}, {1, 0}, {-1, 0} };
        for (int k = 0; k < 4; ++k) {
            int nr = row + d[k][0], nc = col + d[k][1];
            int screen = 0;
            while (inside(nr, nc)) {
                if (!screen) {
                    if (piece_at(b, nr, nc) == 0) EMIT(nr, nc); else screen = 1;
                } else if (piece_at(b, nr, nc) != 0 && xqo_is_valid_move(b, row, col, nr, nc)) {
                    EMIT(nr, nc);
                    break;
                }
                nr += d[k][0]; nc += d[k][1];
            }
        }
        break;
    }
    case T_HORSE: {                                                                /* :248-263 */
        static const int d[8][2] = { {1, 2}, {1, -2}, {-1, 2}, {-1, -2}, {2, 1}, {2, -1}, {-2, 1}, {-2, -1} };
        for (int k = 0; k < 8; ++k) {
            int nr = row + d[k][0], nc = col + d[k][1];
            int lr = row + d[k][0] / 2, lc = col + d[k][1] / 2;
            if (inside(nr, nc) && piece_at(b, lr, lc) == 0 && xqo_is_valid_move(b, row, col, nr, nc)) EMIT(nr, nc);
        }
        break;
    }
    case T_SOLDIER: {                                                              /* :265-283 */
        int fwd = color == XQO_RED ? 1 : -1;
        int nr = row + fwd;
        if (inside(nr, col) && xqo_is_valid_move(b, row, col, nr, col)) EMIT(nr, col);
        if ((color == XQO_RED && row > 4) || (color == XQO_BLACK && row < 5)) {
            int cs[2] = { col - 1, col + 1 };
            for (int k = 0; k < 2; ++k)
                if (inside(row, cs[k]) && xqo_is_valid_move(b, row, col, row, cs[k])) EMIT(row, cs[k]);
        }
        break;
    }
    default: break;
    }
    return n;
}
#undef EMIT

int xqo_move_piece(xqo_board* b, int fr, int fc, int tr, int tc) {                 /* :38-64 */
    if (!xqo_is_valid_move(b, fr, fc, tr, tc)) return 0;       /* invalid: Empty piece, NO state change */
    int captured = b->sq[tr * 9 + tc];
    b->sq[tr * 9 + tc] = b->sq[fr * 9 + fc];
    b->sq[fr * 9 + fc] = 0;
    if (captured != 0) {
        int score = PIECE_SCORE[code_type(captured)];
        if (code_color(captured) == XQO_RED) b->blackScore += score;   /* keyed on the VICTIM's colour */
        else b->redScore += score;
    }
    b->moveCount++;
    b->currentPlayer = b->currentPlayer == XQO_RED ? XQO_BLACK : XQO_RED;
    return captured;
}

int xqo_check_game_over(const xqo_board* b) {                                      /* :286-309 */
    int red = 0, black = 0;
    if (b->moveCount >= 200) return 1;                          /* maxMovePerGame, chessboard.h:63 */
    for (int i = 0; i < 90; ++i) {
        if (code_type(b->sq[i]) == T_GENERAL) {
            if (code_color(b->sq[i]) == XQO_RED) red = 1; else black = 1;
        }
        if (red && black) return 0;
    }
    return 1;
}

int xqo_get_winner(const xqo_board* b) {                                           /* :312-320 */
    for (int i = 0; i < 90; ++i)
        if (code_type(b->sq[i]) == T_GENERAL) return code_color(b->sq[i]);   /* FIRST general in index order */
    return XQO_NONE;
}

/* ------------------------------------------------------------------------------------------------
 * Agent side — src/chessai.cpp
 * ---------------------------------------------------------------------------------------------- */
int xqo_all_valid_actions(const xqo_board* b, int player, uint16_t* codes) {       /* :347-368 */
    int n = 0, moves[32];
    for (int row = 0; row < 10; ++row)
        for (int col = 0; col < 9; ++col) {
            int piece = piece_at(b, row, col);
            if (code_color(piece) == player) {
                int m = xqo_get_valid_moves(b, row, col, moves);
                for (int k = 0; k < m; ++k) {
                    if (n < XQO_MAX_MOVES) codes[n] = (uint16_t)((row * 9 + col) * 90 + moves[k]);
                    ++n;
                }
            }
        }
    return n;
}

int xqo_state_indices(const xqo_board* b, int* idx) {                              /* :268-289 */
    int n = 0;
    for (int s = 0; s < 90; ++s)
        if (b->sq[s] != 0) idx[n++] = s * 14 + (b->sq[s] - 1);   /* (type-1) + 7 if black == code-1 */
    return n;
}

void xqo_state_repr(const xqo_board* b, double* state) {
    for (int i = 0; i < 1260; ++i) state[i] = 0.0;
    for (int s = 0; s < 90; ++s)
        if (b->sq[s] != 0) state[s * 14 + (b->sq[s] - 1)] = 1.0;
}

int xqo_evaluate_board(const xqo_board* b, int color, int moveCount) {             /* :311-345 */
    int score = 0;
    for (int s = 0; s < 90; ++s) {
        int p = b->sq[s];
        int pc = code_color(p);
        if (pc == color) score += PIECE_SCORE[code_type(p)];
        else if (pc != XQO_NONE) score -= PIECE_SCORE[code_type(p)];
    }
    /* `score -= moveCount * 0.1;` on an int: int -> double, subtract, truncate toward zero (:343) */
    score = (int)((double)score - (double)moveCount * 0.1);
    return score;
}

/* ------------------------------------------------------------------------------------------------
 * DQN::selectAction — src/dqn.cpp:24-56
 * ---------------------------------------------------------------------------------------------- */
int xqo_select_action(const double* q, int nq, const uint16_t* codes, int n,
                      int rand1, int rand2, int rand_max, double eps) {
    if (n <= 0) return -1;                                   /* upstream: throws runtime_error (:26-28) */
    double randValue = (double)rand1 / (double)rand_max;     /* :30 */
    if (randValue < eps) return rand2 % n;                   /* :31-34 */
    double maxQ = -INFINITY;
    int best = 0;                                            /* bestAction = validActions[0] (:40) */
    for (int k = 0; k < n; ++k) {
        int to = codes[k] % 90;
        if (to >= nq) continue;                              /* :43-46 */
        double v = q[to];                                    /* indexed by action.to ONLY (:47) */
        if (v > maxQ) { maxQ = v; best = k; }                /* strict >, first max wins (:48-51) */
    }
    return best;
}

/* ------------------------------------------------------------------------------------------------
 * NeuralNetwork — src/dqn.cu
 * ---------------------------------------------------------------------------------------------- */
size_t xqo_nn_num_weights(const int* L, int ns) {
    size_t t = 0;
    for (int i = 0; i + 1 < ns; ++i) t += (size_t)L[i] * (size_t)L[i + 1];
    return t;
}
size_t xqo_nn_num_biases(const int* L, int ns) {
    size_t t = 0;
    for (int i = 0; i + 1 < ns; ++i) t += (size_t)L[i + 1];
    return t;
}
static void offsets(const int* L, int nl, size_t* wo, size_t* bo) {                /* dqn.cu:125-140 */
    size_t tw = 0, tb = 0;
    for (int l = 0; l < nl; ++l) {
        wo[l] = tw; bo[l] = tb;
        tw += (size_t)L[l] * (size_t)L[l + 1];
        tb += (size_t)L[l + 1];
    }
}

int xqo_nn_forward(const int* L, int ns, const double* w, const double* b, const double* in, double* out) {
    int nl = ns - 1;
    if (nl < 1 || nl > XQO_MAX_LAYERS) return -1;
    size_t wo[XQO_MAX_LAYERS] = {0}, bo[XQO_MAX_LAYERS] = {0};
    offsets(L, nl, wo, bo);
    int maxw = 0;
    for (int i = 0; i < ns; ++i) if (L[i] > maxw) maxw = L[i];
    double* cur = (double*)malloc(sizeof(double) * (size_t)maxw);
    double* nxt = (double*)malloc(sizeof(double) * (size_t)maxw);
    memcpy(cur, in, sizeof(double) * (size_t)L[0]);
    for (int l = 0; l < nl; ++l) {                           /* forwardKernel 6-arg, dqn.cu:184-195 */
        int I = L[l], O = L[l + 1];
        const double* W = w + wo[l];
        const double* B = b + bo[l];
        for (int j = 0; j < O; ++j) {
            double sum = 0.0;
            for (int i = 0; i < I; ++i) sum += cur[i] * W[(size_t)j * I + i];
            sum += B[j];                                     /* bias LAST */
            nxt[j] = tanh(sum);                              /* tanh on every layer incl. the output */
        }
        double* t = cur; cur = nxt; nxt = t;
    }
    memcpy(out, cur, sizeof(double) * (size_t)L[nl]);
    free(cur); free(nxt);
    return 0;
}

/* forward with the 7-arg kernel (dqn.cu:275-286): sum starts at the bias; keeps z and a of every layer */
static void forward_train(const int* L, int nl, const size_t* wo, const size_t* bo, const double* w, const double* b,
                          const double* in, double** a, double** z) {
    a[0] = (double*)in;
    for (int l = 0; l < nl; ++l) {
        int I = L[l], O = L[l + 1];
        const double* W = w + wo[l];
        const double* B = b + bo[l];
        for (int j = 0; j < O; ++j) {
            double sum = B[j];                               /* bias FIRST */
            for (int i = 0; i < I; ++i) sum += a[l][i] * W[(size_t)j * I + i];
            z[l][j] = sum;
            a[l + 1][j] = tanh(sum);
        }
    }
}

/* deltas for every layer; returns -1 where the as-written code reads out of bounds (undefined upstream) */
static int compute_deltas(const int* L, int nl, const size_t* wo, const double* w, size_t nw,
                          double** a, double** z, const double* target, int mode, double** d) {
    int out = nl - 1;
    for (int k = 0; k < L[nl]; ++k) {                        /* outputLayerDeltaKernel, dqn.cu:288-295 */
        double err = a[nl][k] - target[k];
        double der = 1 - tanh(z[out][k]) * tanh(z[out][k]);
        d[out][k] = err * der;
    }
    for (int l = out - 1; l >= 0; --l) {                     /* dqn.cu:406-427 */
        if (mode == 0) {
            /* as written: inputSize = L[l+1], outputSize = L[l]  (SHIFTED sizes) feeding hiddenLayerDeltaKernel (:297-308):
             *   delta_l[idx] = (sum_{i<L[l+1]} Wflat[wo[l+1] + i*L[l] + idx] * delta_{l+1}[i]) * (1 - tanh(z_l[idx])^2)
             * only idx < L[l+1] is consumed by the update; z_l has L[l+1] entries. */
            int inputSize = L[l + 1], outputSize = L[l];
            if (L[l + 2] < inputSize) return -1;             /* delta_{l+1} read past its end */
            if (outputSize < L[l + 1]) return -1;            /* delta_l buffer (outputSize doubles) shorter than its reader */
            if (wo[l + 1] + (size_t)(inputSize - 1) * outputSize + (size_t)(L[l + 1] - 1) >= nw) return -1;
            for (int idx = 0; idx < L[l + 1]; ++idx) {
                double sum = 0.0;
                for (int i = 0; i < inputSize; ++i)
                    sum += w[wo[l + 1] + (size_t)i * outputSize + idx] * d[l + 1][i];
                double der = 1 - tanh(z[l][idx]) * tanh(z[l][idx]);
                d[l][idx] = sum * der;
            }
        } else {
            /* textbook: delta_l[idx] = (sum_{k<L[l+2]} W_{l+1}[k][idx] * delta_{l+1}[k]) * (1 - tanh(z_l[idx])^2) */
            for (int idx = 0; idx < L[l + 1]; ++idx) {
                double sum = 0.0;
                for (int k = 0; k < L[l + 2]; ++k)
                    sum += w[wo[l + 1] + (size_t)k * L[l + 1] + idx] * d[l + 1][k];
                double der = 1 - tanh(z[l][idx]) * tanh(z[l][idx]);
                d[l][idx] = sum * der;
            }
        }
    }
    return 0;
}

typedef struct { double* a[XQO_MAX_LAYERS + 1]; double* z[XQO_MAX_LAYERS]; double* d[XQO_MAX_LAYERS]; double* pool; } scratch_t;

static void scratch_alloc(const int* L, int nl, scratch_t* s) {
    size_t tot = 0;
    for (int l = 0; l < nl; ++l) tot += 3 * (size_t)L[l + 1];
    s->pool = (double*)calloc(tot, sizeof(double));
    double* p = s->pool;
    for (int l = 0; l < nl; ++l) {
        s->a[l + 1] = p; p += L[l + 1];
        s->z[l] = p; p += L[l + 1];
        s->d[l] = p; p += L[l + 1];
    }
}

int xqo_nn_backprop(const int* L, int ns, double* w, double* b, const double* in, const double* target,
                    double lr, int mode) {                   /* dqn.cu:323-467 */
    int nl = ns - 1;
    if (nl < 1 || nl > XQO_MAX_LAYERS) return -1;
    size_t wo[XQO_MAX_LAYERS] = {0}, bo[XQO_MAX_LAYERS] = {0};
    offsets(L, nl, wo, bo);
    scratch_t s; scratch_alloc(L, nl, &s);
    forward_train(L, nl, wo, bo, w, b, in, s.a, s.z);
    int rc = compute_deltas(L, nl, wo, w, xqo_nn_num_weights(L, ns), s.a, s.z, target, mode, s.d);
    if (rc == 0) {
        /* ALL deltas use pre-update weights; then per layer (updateWeightsBiasesKernel, dqn.cu:310-319, :430-447).
         * a_l for l>=1 is read after cudaFree upstream; the intended (and in practice observed) value is the
         * hidden activation (verified max-abs-diff 0.0). */
        for (int l = 0; l < nl; ++l) {
            int I = L[l], O = L[l + 1];
            for (int j = 0; j < O; ++j) {
                b[bo[l] + j] -= lr * s.d[l][j];
                for (int i = 0; i < I; ++i)
                    w[wo[l] + (size_t)j * I + i] -= lr * s.d[l][j] * s.a[l][i];
            }
        }
    }
    free(s.pool);
    return rc;
}

int xqo_nn_accum_grad(const int* L, int ns, const double* w, const double* b, const double* in, const double* target,
                      int mode, double* gw, double* gb) {
    int nl = ns - 1;
    if (nl < 1 || nl > XQO_MAX_LAYERS) return -1;
    size_t wo[XQO_MAX_LAYERS] = {0}, bo[XQO_MAX_LAYERS] = {0};
    offsets(L, nl, wo, bo);
    scratch_t s; scratch_alloc(L, nl, &s);
    forward_train(L, nl, wo, bo, w, b, in, s.a, s.z);
    int rc = compute_deltas(L, nl, wo, w, xqo_nn_num_weights(L, ns), s.a, s.z, target, mode, s.d);
    if (rc == 0)
        for (int l = 0; l < nl; ++l) {
            int I = L[l], O = L[l + 1];
            for (int j = 0; j < O; ++j) {
                gb[bo[l] + j] += s.d[l][j];
                double dj = s.d[l][j];
                if (dj != 0.0)
                    for (int i = 0; i < I; ++i) gw[wo[l] + (size_t)j * I + i] += dj * s.a[l][i];
            }
        }
    free(s.pool);
    return rc;
}

int xqo_nn_forward_all(const int* L, int ns, const double* w, const double* b, const double* in, double* acts) {
    int nl = ns - 1;
    if (nl < 1 || nl > XQO_MAX_LAYERS) return -1;
    size_t wo[XQO_MAX_LAYERS] = {0}, bo[XQO_MAX_LAYERS] = {0};
    offsets(L, nl, wo, bo);
    scratch_t s; scratch_alloc(L, nl, &s);
    forward_train(L, nl, wo, bo, w, b, in, s.a, s.z);
    double* p = acts;
    for (int l = 0; l < nl; ++l) { memcpy(p, s.a[l + 1], sizeof(double) * (size_t)L[l + 1]); p += L[l + 1]; }
    free(s.pool);
    return 0;
}

int xqo_td_target(const int* L, int ns, const double* w, const double* b, const double* state,
                  const double* next_state, int action_to, double reward, int done, double gamma, double* targetQ) {
    int nout = L[ns - 1];
    if (xqo_nn_forward(L, ns, w, b, state, targetQ)) return -1;      /* chessai.cpp:122 (getQValues = forward) */
    if (done) {
        targetQ[action_to] = reward;                                 /* :123-124 */
    } else {
        double* nq = (double*)malloc(sizeof(double) * (size_t)nout);
        xqo_nn_forward(L, ns, w, b, next_state, nq);                 /* ONLINE net (:126) */
        double m = nq[0];
        for (int k = 1; k < nout; ++k) if (nq[k] > m) m = nq[k];     /* max over ALL outputs (:127) */
        targetQ[action_to] = reward + gamma * m;
        free(nq);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * ChessAI::train loop body — src/chessai.cpp:90-167 (one episode)
 * ---------------------------------------------------------------------------------------------- */
int xqo_rand(uint64_t* s) {
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (int)((*s >> 33) & 0x7fffffff);
}

int xqo_train_episode(const int* L, int ns, double* w, double* b, double lr, double gamma, double eps,
                      uint64_t* rng, int mode, xqo_episode_stats* st) {
    const int maxMovesPerGame = 200;                                 /* :88 */
    int nout = L[ns - 1];
    xqo_board board;
    xqo_reset(&board);                                               /* :90 */
    int currentPlayer = XQO_RED;                                     /* :91 */
    double* state = (double*)malloc(sizeof(double) * 1260);
    double* next = (double*)malloc(sizeof(double) * 1260);
    double* q = (double*)malloc(sizeof(double) * (size_t)nout);
    double* tq = (double*)malloc(sizeof(double) * (size_t)nout);
    xqo_state_repr(&board, state);                                   /* :92 */
    int moveCount = 0, steps = 0, syncs = 0, rc = 0;
    uint16_t codes[XQO_MAX_MOVES];
    while (!xqo_check_game_over(&board) && moveCount < maxMovesPerGame) {    /* :96 */
        int n = xqo_all_valid_actions(&board, currentPlayer, codes);         /* :98 */
        if (n == 0) break;                                                   /* :100-103 */
        if (n > XQO_MAX_MOVES) n = XQO_MAX_MOVES;
        /* selectAction(state, 0.1, valid) :106 — rand() consumed once, twice when exploring */
        int r1 = xqo_rand(rng);
        int idx;
        if ((double)r1 / 2147483647.0 < eps) idx = xqo_rand(rng) % n;
        else {
            xqo_nn_forward(L, ns, w, b, state, q);
            idx = xqo_select_action(q, nout, codes, n, 0x7fffffff, 0, 0x7fffffff, -1.0);
        }
        int from = codes[idx] / 90, to = codes[idx] % 90;
        xqo_move_piece(&board, from / 9, from % 9, to / 9, to % 9);          /* :113 */
        moveCount = board.moveCount;                                         /* :115 */
        double reward = xqo_evaluate_board(&board, currentPlayer, moveCount);/* :116 */
        xqo_state_repr(&board, next);                                        /* :118 */
        int done = xqo_check_game_over(&board) || (moveCount + 1 >= maxMovesPerGame);   /* :119 */
        xqo_td_target(L, ns, w, b, state, next, to, reward, done, gamma, tq);           /* :122-128 */
        rc |= xqo_nn_backprop(L, ns, w, b, state, tq, lr, mode);                        /* :131 */
        memcpy(state, next, sizeof(double) * 1260);                                     /* :134 */
        currentPlayer = currentPlayer == XQO_RED ? XQO_BLACK : XQO_RED;                 /* :137 */
        if (moveCount % 100 == 0) ++syncs;   /* updateTargetNetwork(): copies STALE host weights upstream (:140) — no effect */
        ++steps;
    }
    if (st) {
        st->steps = steps; st->winner = xqo_get_winner(&board);
        st->redScore = board.redScore; st->blackScore = board.blackScore;
        st->moveCount = board.moveCount; st->target_syncs = syncs;
    }
    free(state); free(next); free(q); free(tq);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * ChessAI::startSelfPlay (chessai.cpp:191-266), getAIMove (:29-83), onGameCompleted (:370-393)
 * ---------------------------------------------------------------------------------------------- */
static int select_with_rand(const int* L, int ns, const double* w, const double* b, const double* state, double* q,
                            const uint16_t* codes, int n, xqo_rand_fn rnd, void* ctx, int rand_max, double eps) {
    /* DQN::selectAction, dqn.cpp:24-56: one rand() always, a second one only on the explore branch */
    const int r1 = rnd(ctx);
    if ((double)r1 / (double)rand_max < eps) return rnd(ctx) % n;
    xqo_nn_forward(L, ns, w, b, state, q);
    return xqo_select_action(q, L[ns - 1], codes, n, rand_max, 0, rand_max, -1.0);
}

int xqo_selfplay_game(const int* L, int ns, double* w, double* b, double lr, double gamma, double eps,
                      xqo_rand_fn rnd, void* ctx, int rand_max, int mode, xqo_board* board_out, xqo_episode_stats* st) {
    const int nout = L[ns - 1];
    xqo_board board;
    xqo_reset(&board);                                                       /* :193 */
    double* state = (double*)malloc(sizeof(double) * 1260);
    double* next = (double*)malloc(sizeof(double) * 1260);
    double* q = (double*)malloc(sizeof(double) * (size_t)nout);
    double* tq = (double*)malloc(sizeof(double) * (size_t)nout);
    xqo_state_repr(&board, state);                                           /* :194 */
    int steps = 0, syncs = 0, rc = 0;
    uint16_t codes[XQO_MAX_MOVES];
    while (!xqo_check_game_over(&board)) {                                   /* :196 — no local cap */
        const int currentPlayer = board.currentPlayer;                       /* :197 */
        int n = xqo_all_valid_actions(&board, currentPlayer, codes);         /* :200 */
        if (n == 0) break;                                                   /* :202-205 */
        if (n > XQO_MAX_MOVES) n = XQO_MAX_MOVES;
        const int idx = select_with_rand(L, ns, w, b, state, q, codes, n, rnd, ctx, rand_max, eps);   /* :208 */
        const int from = codes[idx] / 90, to = codes[idx] % 90;
        xqo_move_piece(&board, from / 9, from % 9, to / 9, to % 9);          /* :216 */
        const double reward = xqo_evaluate_board(&board, currentPlayer, board.moveCount);   /* :220 */
        xqo_state_repr(&board, next);                                        /* :223 */
        const int done = xqo_check_game_over(&board);                        /* :226 */
        xqo_td_target(L, ns, w, b, state, next, to, reward, done, gamma, tq);/* :229-235 */
        rc |= xqo_nn_backprop(L, ns, w, b, state, tq, lr, mode);             /* :238 */
        memcpy(state, next, sizeof(double) * 1260);                          /* :241 */
        if (board.moveCount % 100 == 0) ++syncs;                             /* :244-246 (stale copy upstream: no effect) */
        ++steps;
    }
    if (st) {
        st->steps = steps; st->winner = xqo_get_winner(&board);              /* :250 */
        st->redScore = board.redScore; st->blackScore = board.blackScore;    /* :251 */
        st->moveCount = board.moveCount; st->target_syncs = syncs;
    }
    if (board_out) *board_out = board;
    free(state); free(next); free(q); free(tq);
    return rc;
}

void xqo_get_ai_move(const xqo_board* board, int color, const int* L, int ns, const double* w, const double* b,
                     xqo_rand_fn rnd, void* ctx, int rand_max, int mv[4]) {
    double* state = (double*)malloc(sizeof(double) * 1260);
    double* q = (double*)malloc(sizeof(double) * (size_t)L[ns - 1]);
    uint16_t codes[XQO_MAX_MOVES];
    int moves[32];
    xqo_state_repr(board, state);                                            /* :31 */
    mv[0] = mv[1] = mv[2] = mv[3] = -1;
    for (int attempt = 0; attempt < 10; ++attempt) {                         /* :32-34 */
        int n = xqo_all_valid_actions(board, color, codes);                  /* :36 */
        if (n == 0) continue;                                                /* :38-40 */
        if (n > XQO_MAX_MOVES) n = XQO_MAX_MOVES;
        const int idx = select_with_rand(L, ns, w, b, state, q, codes, n, rnd, ctx, rand_max, 0.1);   /* :43 */
        const int from = codes[idx] / 90, to = codes[idx] % 90;
        const int fr = from / 9, fc = from % 9, tr = to / 9, tc = to % 9;
        const int m = xqo_get_valid_moves(board, fr, fc, moves);             /* :50 */
        const int pc = board->sq[from];
        const int pcolor = pc == 0 ? XQO_NONE : (pc > 7 ? XQO_BLACK : XQO_RED);
        if (pcolor == color && m > 0) {                                      /* :53 */
            for (int k = 0; k < m; ++k)
                if (moves[k] == to) { mv[0] = fr; mv[1] = fc; mv[2] = tr; mv[3] = tc; free(state); free(q); return; }   /* :55-64 */
        }
    }
    const int n = xqo_all_valid_actions(board, color, codes);                /* :69 */
    if (n > 0) {                                                             /* :70-73 */
        const int c = codes[rnd(ctx) % (n > XQO_MAX_MOVES ? XQO_MAX_MOVES : n)];   /* :75-76 (QRandomGenerator upstream) */
        mv[0] = (c / 90) / 9; mv[1] = (c / 90) % 9; mv[2] = (c % 90) / 9; mv[3] = (c % 90) % 9;
    }
    free(state); free(q);
}

int xqo_game_log_line(int gameNumber, int redScore, int blackScore, int numGames, char* out, size_t cap) {
    const char* result = redScore > blackScore ? "Red wins!" : blackScore > redScore ? "Black wins!" : "It's a draw!";   /* :376 */
    int n = snprintf(out, cap, "Game %d completed. Red Score: %d, Black Score: %d. %s\n", gameNumber, redScore, blackScore,
                     result);                                                /* :379-380 */
    if (gameNumber == numGames && n >= 0 && (size_t)n < cap)                 /* :383-385 */
        n += snprintf(out + n, cap - (size_t)n, "AI self-play session completed. Total games: %d\n\n", numGames);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11) — independent restatement for the checker
 * ---------------------------------------------------------------------------------------------- */
void xqo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------------------------------------
 * Build-defined batched self-play step for ONE game (DESIGN.md "VecEnv semantics"): the body of
 * chessai.cpp:96-143 minus the learning calls, with rand() replaced by Philox(ctr={step,0,game,0}, key=seed)
 * and auto-reset on termination.
 * ---------------------------------------------------------------------------------------------- */
void xqo_selfplay_step(xqo_board* b, const float* q90, uint64_t seed, uint32_t game_id, uint32_t step_id,
                       uint32_t eps_u32, xqo_step_out* o) {
    uint16_t codes[XQO_MAX_MOVES];
    memset(o, 0, sizeof *o);
    int player = b->currentPlayer;
    int n = xqo_all_valid_actions(b, player, codes);
    if (n > XQO_MAX_MOVES) n = XQO_MAX_MOVES;
    o->n_moves = n;
    o->winner = XQO_NONE;
    if (n == 0) {                                       /* chessai.cpp:100-103: break -> episode over */
        o->action_code = -1; o->done = 1; o->terminated = 1;
        o->winner = (uint8_t)xqo_get_winner(b);
        o->redScore = b->redScore; o->blackScore = b->blackScore; o->moveCount = b->moveCount;
        xqo_reset(b);
        return;
    }
    uint32_t ctr[4] = { step_id, 0u, game_id, 0u }, key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) }, r[4];
    xqo_philox4x32(ctr, key, r);
    int idx;
    if (q90 == NULL || r[0] < eps_u32) {
        idx = (int)(r[1] % (uint32_t)n);
        o->explored = 1;
    } else {
        float best = -INFINITY; idx = 0;
        for (int k = 0; k < n; ++k) {
            float v = q90[codes[k] % 90];
            if (v > best) { best = v; idx = k; }
        }
    }
    o->action_code = codes[idx];
    int from = codes[idx] / 90, to = codes[idx] % 90;
    xqo_move_piece(b, from / 9, from % 9, to / 9, to % 9);
    o->reward = xqo_evaluate_board(b, player, b->moveCount);
    int over = xqo_check_game_over(b);
    o->done = (uint8_t)(over || (b->moveCount + 1 >= 200));
    o->terminated = (uint8_t)over;
    o->redScore = b->redScore; o->blackScore = b->blackScore; o->moveCount = b->moveCount;
    if (over) {
        o->winner = (uint8_t)xqo_get_winner(b);
        xqo_reset(b);
    }
}
