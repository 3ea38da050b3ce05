// xq_env.hip — batched rules engine: one wavefront per board, thousands of boards per launch (gfx950).
//
// Replaces, for n_games boards at once, the per-ply work of ChessAI::train (reference chessai.cpp:96-119):
// getAllValidActions -> selectAction (epsilon-greedy over q[action.to]) -> movePiece -> evaluateBoard ->
// checkGameOver, plus the episode bookkeeping of chessai.cpp:146-162.  The ordered move list never leaves LDS on
// the self-play path.  HBM traffic per game and ply: board 48 B + meta 16 B read and written,
// 360 B of Q-values read, one transition record (48+48+4+4+1 B) written.
#include "xq_internal.h"
#include <hip/hip_ext.h>
#include "xq_rules.hip.h"

#include <algorithm>

namespace xq {

enum { MODE_LEGAL = 0, MODE_STEP = 1, MODE_SELFPLAY = 2 };

struct EnvParams {
    uint32_t* boards;
    uint4* meta;
    uint4* stats;
    int n_games;
    // MODE_LEGAL
    int player;
    uint16_t* codes_out;
    int32_t* counts_out;
    // MODE_STEP
    const int32_t* actions;
    int auto_reset;
    // MODE_SELFPLAY
    const float* q90;
    int q_stride;
    // ... or the k-slabs of the select head as the Q-net left them (QSource, xq_internal.h): q[j] = tanh(bias[j] + slabs summed in the
    // order of q_head_finish_kernel) — the kernel that would do exactly this for all 96 x n values is then not launched
    const float* q_slabs; long long q_slab_stride; int q_nslabs; const float* q_bias;
    uint32_t eps_u32, seed_lo, seed_hi, first_game_id;
    xq_step_result* results;
    // replay ring (optional)
    ReplayDev rp;
    int rp_write_base;
    // episode ring
    xq_episode_record* ep_ring;
    int ep_cap;
    unsigned long long* ep_head;
};

__constant__ uint32_t c_start_words[kBoardWords];

// meta.x = moveCount | player << 16 | flags.  META_TRACKED: the scores account for every piece missing from the board (own
// material = 1480 - the other side's score) and each living general stands in its own palace — true from the start position on and
// preserved by every move (movePiece credits the victim's value, chessboard.cpp:51-58; a general cannot leave its palace,
// :328-343), re-checked by xq_env_set_state.  Then evaluateBoard (chessai.cpp:311-345), checkGameOver and getWinner
// (chessboard.cpp:286-320) need no scan of the 90 squares: score(mover) = own score - other score, a general dies only by capture,
// and the first general in index order is Red's while it lives.
enum : uint32_t { META_TRACKED = 1u << 17, META_RED_GENERAL = 1u << 18, META_BLACK_GENERAL = 1u << 19,
                  META_START = META_TRACKED | META_RED_GENERAL | META_BLACK_GENERAL };
constexpr int kSideMaterial = 1000 + 2 * 20 + 2 * 20 + 2 * 40 + 2 * 90 + 2 * 45 + 5 * 10;      // 1480, chessboard.h:23-31

template <int MODE>
__global__ __launch_bounds__(256) void env_kernel(EnvParams P) {
    __shared__ WaveSlab slabs[4];
    // everything per game is wave-uniform: keeping the wave index, the game index and the meta record in SGPRs lets the compiler
    // run the bookkeeping and all ten Philox rounds on the scalar unit instead of on 64 identical lanes
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = lane_id();
    const int g = (int)blockIdx.x * 4 + wid;
    const bool active = g < P.n_games;
    WaveSlab& S = slabs[wid];

    uint32_t word = 0;
    uint4 m = make_uint4(0, 0, 0, 0);
    // The Q-values the select will need are requested NOW, beside the board and meta loads, whether or not this board will explore
    // (one board in ten does, and wastes them): behind move generation they were one exposed memory round trip per wave — 45 % of the
    // wave cycles of the Q-policy launch were waits (profiles/r04_a_pmc_hbm_traffic.json).  Up to 8 k-slabs of the select head
    // (last hidden width <= 512), two outputs per lane (lane, 64 + lane < 90).
    float qpre[2][8];
    float qbias[2] = {0.f, 0.f};
    const bool q_from_slabs = MODE == MODE_SELFPLAY && P.q_slabs != nullptr && P.q_nslabs <= 8;
    if (active) {
        if (lane < kBoardWords) word = P.boards[(size_t)g * kBoardWords + lane];
        m = P.meta[g];
        if (q_from_slabs) {
            const float* p0 = P.q_slabs + (size_t)g * 96 + lane;
            const float* p1 = P.q_slabs + (size_t)g * 96 + 64 + (lane < 26 ? lane : 0);
#pragma unroll
            for (int z = 0; z < 8; ++z) {
                const bool in = z < P.q_nslabs;
                qpre[0][z] = in ? p0[z * P.q_slab_stride] : 0.f;
                qpre[1][z] = in ? p1[z * P.q_slab_stride] : 0.f;
            }
            qbias[0] = P.q_bias[lane];
            qbias[1] = P.q_bias[64 + (lane < 26 ? lane : 0)];
        } else if (MODE == MODE_SELFPLAY && P.q90 != nullptr) {
            const float* qrow = P.q90 + (size_t)g * P.q_stride;
            qpre[0][0] = qrow[lane];
            qpre[1][0] = qrow[64 + (lane < 26 ? lane : 0)];
        }
    }
    m.x = __builtin_amdgcn_readfirstlane(m.x); m.y = __builtin_amdgcn_readfirstlane(m.y);
    m.z = __builtin_amdgcn_readfirstlane(m.z); m.w = __builtin_amdgcn_readfirstlane(m.w);
    unpack_to_slab(word, S.sq);
    wave_sync();

    int move_count = (int)(m.x & 0xFFFFu);
    int player = (int)((m.x >> 16) & 1u);
    uint32_t flags = m.x & (META_TRACKED | META_RED_GENERAL | META_BLACK_GENERAL);
    int red = (int)(m.y & 0xFFFFu), black = (int)(m.y >> 16);
    uint32_t plies = m.z, episodes = m.w;

    if (MODE == MODE_LEGAL) {
        const int pl = P.player < 0 ? player : P.player;
        const int n = gen_all_actions(S, pl);
        wave_sync();
        if (active) {
            uint16_t* out = P.codes_out + (size_t)g * kMaxMoves;
            out[lane] = lane < n ? S.moves[lane] : (uint16_t)0;
            out[lane + 64] = lane + 64 < n ? S.moves[lane + 64] : (uint16_t)0;
            if (lane == 0) P.counts_out[g] = n;
        }
        return;
    }

    // ---- choose the action ---------------------------------------------------------------------------------
    int from = 0, to = 0, n_moves = 0, action_code = -1;
    bool have_action = false, valid = false, explored = false;
    const int mover = player;

    if (MODE == MODE_SELFPLAY) {
        n_moves = gen_all_actions(S, player);                       // chessai.cpp:98
        wave_sync();
        if (n_moves > 0) {
            const Philox4 r = philox4x32_10(plies, 0u, P.first_game_id + (uint32_t)g, 0u, P.seed_lo, P.seed_hi);
            const bool have_q = P.q90 != nullptr || P.q_slabs != nullptr;
            explored = !have_q || (r.v[0] < P.eps_u32);             // dqn.cpp:30-31
            int idx;
            if (explored) {
                idx = (int)(r.v[1] % (uint32_t)n_moves);            // dqn.cpp:33
            } else {
                if (q_from_slabs) {
                    auto q_pre = [&](int e) {                                        // q_head_finish_kernel's association, on the
                        float sum = 0.f;                                             // values requested at the top of the kernel
#pragma unroll
                        for (int z = 0; z < 8; z += 4) {
                            if (z < P.q_nslabs) {
                                float t = qpre[e][z] + qpre[e][z + 1];
                                if (z + 3 < P.q_nslabs) t += qpre[e][z + 2] + qpre[e][z + 3];
                                sum = z == 0 ? t : sum + t;
                            }
                        }
                        return qbias[e] + sum;                                       // pre-activation: tanh below, on the candidates only
                    };
                    S.q[lane] = q_pre(0);
                    if (lane < 26) S.q[64 + lane] = q_pre(1);
                } else if (P.q_slabs != nullptr) {                                   // (more than 8 k-slabs: loaded here)
                    auto q_of = [&](int j) {
                        const float* p = P.q_slabs + (size_t)g * 96 + j;
                        float sum = 0.f;
                        for (int z = 0; z < P.q_nslabs; z += 4) {                    // q_head_finish_kernel's association
                            float t = p[z * P.q_slab_stride] + p[(z + 1) * P.q_slab_stride];
                            if (z + 3 < P.q_nslabs) t += p[(z + 2) * P.q_slab_stride] + p[(z + 3) * P.q_slab_stride];
                            sum = z == 0 ? t : sum + t;
                        }
                        return P.q_bias[j] + sum;
                    };
                    S.q[lane] = q_of(lane);
                    if (lane < 26) S.q[64 + lane] = q_of(64 + lane);
                } else {
                    S.q[lane] = qpre[0][0];                                          // dqn.cpp:37 (the row requested at the top)
                    if (lane < 26) S.q[64 + lane] = qpre[1][0];
                }
                wave_sync();
                const float NEG = -__builtin_inff();
                float v0 = NEG, v1 = NEG;
                // q[action.to], dqn.cpp:47.  From slabs S.q holds the head's PRE-activations and tanh (libm's, the output layer's: same
                // argument, same bits as a tanh per output) is taken of the candidates' values only: one pass over the wave for the usual
                // <= 64 moves instead of two passes over the 90 outputs (~40 vector instructions each)
                const bool pre = P.q_slabs != nullptr;
                if (lane < n_moves) { v0 = S.q[S.moves[lane] % 90]; if (pre) v0 = tanhf(v0); }
                if (n_moves > 64) {                                            // (wave-uniform)
                    if (lane + 64 < n_moves) { v1 = S.q[S.moves[lane + 64] % 90]; if (pre) v1 = tanhf(v1); }
                }
                if (!(v0 == v0)) v0 = NEG;                                     // NaN never wins `q > maxQ`
                if (!(v1 == v1)) v1 = NEG;
                const float ma = wave_max(v0), mb = wave_max(v1);
                if (mb > ma) {                                                 // strict >: first maximum wins (dqn.cpp:48)
                    const unsigned long long b = __ballot(lane + 64 < n_moves && v1 == mb);
                    idx = b ? 64 + (__ffsll((long long)b) - 1) : 0;
                } else {
                    const unsigned long long b = __ballot(lane < n_moves && v0 == ma);
                    idx = b ? (__ffsll((long long)b) - 1) : 0;                 // all -inf: validActions[0] (dqn.cpp:40)
                }
            }
            action_code = __builtin_amdgcn_readfirstlane((int)S.moves[idx]);
            from = action_code / 90;
            to = action_code - from * 90;
            have_action = true;
            valid = true;                                           // generated moves pass isValidMove by construction
        }
    } else {  // MODE_STEP: caller-chosen action, validated like movePiece() does (chessboard.cpp:39-41)
        const int a = active ? P.actions[g] : -1;
        action_code = a;
        if (a >= 0 && a < 8100) {
            from = a / 90;
            to = a - from * 90;
            valid = is_valid_move(S.sq, from / 9, from % 9, to / 9, to % 9);
        }
        have_action = true;
    }

    // ---- movePiece (chessboard.cpp:38-64) -------------------------------------------------------------------
    int captured = 0;
    if (valid) {
        const int moving = __builtin_amdgcn_readfirstlane((int)S.sq[from]);
        captured = __builtin_amdgcn_readfirstlane((int)S.sq[to]);
        wave_sync();
        if (lane == 0) {
            S.sq[to] = (uint8_t)moving;
            S.sq[from] = 0;
        }
        wave_sync();
        if (captured != 0) {
            const int sc = piece_value(captured);
            if (captured <= 7) black += sc; else red += sc;        // credited by the VICTIM's colour (:51-58)
            if (captured == 1) flags &= ~META_RED_GENERAL;
            if (captured == 8) flags &= ~META_BLACK_GENERAL;
        }
        move_count += 1;
        player ^= 1;
        plies += 1;
    }

    // ---- reward, terminal test (chessai.cpp:115-119) ------------------------------------------------------------
    int reward, winner_now;
    bool red_general, black_general;
    if (flags & META_TRACKED) {          // no board scan: see META_TRACKED
        const int score = mover == C_RED ? red - black : black - red;      // (1480 - lost own) - (1480 - lost enemy)
        {
#pragma clang fp contract(off)
            const double pen = __dmul_rn((double)move_count, 0.1);         // `score -= moveCount * 0.1` on an int, chessai.cpp:342
            reward = (int)__dsub_rn((double)score, pen);
        }
        red_general = (flags & META_RED_GENERAL) != 0;
        black_general = (flags & META_BLACK_GENERAL) != 0;
        winner_now = red_general ? C_RED : (black_general ? C_BLACK : C_NONE);
    } else {
        reward = evaluate_board_wave(S.sq, mover, move_count);
        const BoardStatus st = board_status_wave(S.sq);
        red_general = st.red_general; black_general = st.black_general;
        winner_now = st.first_general_color;
    }
    const bool over = move_count >= 200 || !red_general || !black_general;         // chessboard.cpp:286-309
    const bool no_action = (MODE == MODE_SELFPLAY) && !have_action;                 // chessai.cpp:100-103
    const bool done = over || no_action || (move_count + 1 >= 200);
    const bool terminated = over || no_action;
    // the episode ENDS on the ply that made it terminal; a rejected move on an already finished board (the facade's
    // checkGameOver() probe, a retried step) reports `terminated` but changes nothing: no stats, no episode record, no reset
    const bool ended_now = terminated && (MODE == MODE_SELFPLAY || valid);
    const bool do_reset = ended_now && (MODE == MODE_SELFPLAY || P.auto_reset != 0);
    const int winner = terminated ? winner_now : C_NONE;

    const uint32_t next_word = pack_from_slab(S.sq);               // s' = board after the move, before any reset

    if (!active) return;

    if (P.results != nullptr && lane == 0) {
        xq_step_result r;
        r.action = have_action ? action_code : -1;
        r.n_moves = n_moves;
        r.reward = reward;
        r.captured = (uint8_t)captured;
        r.valid = valid ? 1 : 0;
        r.done = done ? 1 : 0;
        r.terminated = terminated ? 1 : 0;
        r.winner = (uint8_t)winner;
        r.explored = explored ? 1 : 0;
        r.move_count = (uint16_t)move_count;
        r.red_score = (int16_t)red;
        r.black_score = (int16_t)black;
        P.results[g] = r;
    }

    if (MODE == MODE_SELFPLAY && P.rp.capacity > 0) {
        int slot = P.rp_write_base + g;
        slot -= (slot >= P.rp.capacity) ? P.rp.capacity : 0;
        if (lane < kBoardWords) {
            P.rp.boards[(size_t)slot * kBoardWords + lane] = word;
            P.rp.next_boards[(size_t)slot * kBoardWords + lane] = next_word;
        }
        if (lane == 0) {
            P.rp.action_to[slot] = have_action ? to : -1;
            P.rp.reward[slot] = (float)reward;
            P.rp.done[slot] = done ? 1 : 0;
            // prioritized replay: a new transition enters with the largest priority assigned so far (an empty one never gets sampled)
            if (P.rp.prio != nullptr) P.rp.prio[slot] = have_action ? __uint_as_float(*P.rp.pmax_snap) : 0.f;
        }
    }

    const bool policy_q = P.q90 != nullptr || P.q_slabs != nullptr;
    if ((captured != 0 || ended_now || (explored && policy_q)) && lane == 0) {
        uint4 s = P.stats[g];
        if (ended_now && winner == C_RED) s.x += 1;
        if (ended_now && winner == C_BLACK) s.y += 1;
        if (captured != 0) s.z += 1;
        if (explored && policy_q) s.w += 1;
        P.stats[g] = s;
    }

    uint32_t out_word = next_word;
    if (ended_now && lane == 0 && P.ep_ring != nullptr) {           // gameCompleted(game, red, black), chessai.cpp:162
        const unsigned long long h = atomicAdd(P.ep_head, 1ull);
        xq_episode_record e;
        e.game_id = P.first_game_id + (uint32_t)g;
        e.episode = episodes + 1;
        e.red_score = (int16_t)red;
        e.black_score = (int16_t)black;
        e.move_count = (uint16_t)move_count;
        e.winner = (uint8_t)winner;
        e.reserved = no_action ? 1 : 0;
        P.ep_ring[h % (unsigned long long)P.ep_cap] = e;
    }
    if (do_reset) {                                                 // board->reset(), chessai.cpp:90
        out_word = lane < kBoardWords ? c_start_words[lane] : 0u;
        move_count = 0; player = C_RED; red = 0; black = 0;
        flags = META_START;
        episodes += 1;
    }
    if (lane < kBoardWords) P.boards[(size_t)g * kBoardWords + lane] = out_word;
    if (lane == 0)
        P.meta[g] = make_uint4((uint32_t)move_count | ((uint32_t)player << 16) | flags, (uint32_t)red | ((uint32_t)black << 16),
                               plies, episodes);
}

// isValidMove for all 90x90 pairs of one game (parity tests of E4-E11 through the validator path)
__global__ __launch_bounds__(256) void valid_matrix_kernel(const uint32_t* boards, int g, uint8_t* out) {
    __shared__ uint8_t sq[96];
    if (threadIdx.x < kBoardWords) {
        const uint32_t w = boards[(size_t)g * kBoardWords + threadIdx.x];
        for (int k = 0; k < 8; ++k) sq[threadIdx.x * 8 + k] = (uint8_t)((w >> (4 * k)) & 15u);
    }
    __syncthreads();
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < 8100; i += (int)(gridDim.x * blockDim.x)) {
        const int f = i / 90, t = i - f * 90;
        out[i] = is_valid_move(sq, f / 9, f % 9, t / 9, t % 9) ? 1 : 0;
    }
}

// The seven public validators isValid{General..Soldier}Move (chessboard.h:50-56) of one game: out[(type-1)*8100 + f*90 + t]
// for in-board (from, to); n_query > 0 instead evaluates `n_query` explicit (type, fr, fc, tr, tc) tuples (any coordinates).
__global__ __launch_bounds__(256) void rule_matrix_kernel(const uint32_t* boards, int g, uint8_t* out, const int32_t* query, int n_query) {
    __shared__ uint8_t sq[96];
    if (threadIdx.x < kBoardWords) {
        const uint32_t w = boards[(size_t)g * kBoardWords + threadIdx.x];
        for (int k = 0; k < 8; ++k) sq[threadIdx.x * 8 + k] = (uint8_t)((w >> (4 * k)) & 15u);
    }
    __syncthreads();
    if (n_query > 0) {
        for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n_query; i += (int)(gridDim.x * blockDim.x)) {
            const int32_t* q = query + 5 * i;
            out[i] = piece_rule(sq, q[0], q[1], q[2], q[3], q[4]) ? 1 : 0;
        }
        return;
    }
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < 7 * 8100; i += (int)(gridDim.x * blockDim.x)) {
        const int type = 1 + i / 8100, ft = i % 8100;
        const int f = ft / 90, t = ft - f * 90;
        // from == to on a line piece runs into signed overflow upstream (loop `i != end` from start +- 1): reported as 0 here
        out[i] = (f != t || type < T_CHARIOT || type > T_CANNON) && piece_rule(sq, type, f / 9, f % 9, t / 9, t % 9) ? 1 : 0;
    }
}

// getWinner() for a range of games, one wave per game
__global__ __launch_bounds__(256) void winner_kernel(const uint32_t* boards, int first, int n, uint8_t* out) {
    __shared__ WaveSlab slabs[4];
    const int wid = (int)(threadIdx.x >> 6), lane = lane_id();
    const int i = (int)blockIdx.x * 4 + wid;
    uint32_t word = 0;
    if (i < n && lane < kBoardWords) word = boards[(size_t)(first + i) * kBoardWords + lane];
    unpack_to_slab(word, slabs[wid].sq);
    wave_sync();
    const BoardStatus st = board_status_wave(slabs[wid].sq);
    if (i < n && lane == 0) out[i] = (uint8_t)st.first_general_color;
}

__global__ void fill_start_kernel(uint32_t* boards, uint4* meta, uint4* stats, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n * kBoardWords) boards[i] = c_start_words[i % kBoardWords];
    if (i < n) {
        meta[i] = make_uint4(META_START, 0, 0, 0);
        stats[i] = make_uint4(0, 0, 0, 0);
    }
}

static int upload_start_words(hipStream_t stream) {
    uint8_t sq[96];
    uint32_t words[kBoardWords];
    start_position(sq);
    pack_board(sq, words);
    XQ_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_start_words), words, sizeof words, 0, hipMemcpyHostToDevice, stream));
    return XQ_OK;
}

static EnvParams base_params(xq_env* e) {
    EnvParams P;
    memset(&P, 0, sizeof P);
    P.boards = e->boards;
    P.meta = e->meta;
    P.stats = e->stats;
    P.n_games = e->n;
    P.seed_lo = (uint32_t)e->seed;
    P.seed_hi = (uint32_t)(e->seed >> 32);
    P.first_game_id = e->first_id;
    P.ep_ring = e->ep_ring;
    P.ep_cap = e->ep_cap;
    P.ep_head = e->ep_head;
    return P;
}

int env_selfplay_launch(xq_env* e, const float* q90_dev, int q_stride, uint32_t eps_u32, xq_step_result* results_dev,
                        xq_replay* replay, hipStream_t on, const QSource* qs, hipEvent_t ev_start, hipEvent_t ev_stop) {
    EnvParams P = base_params(e);
    P.q90 = q90_dev;
    P.q_stride = q_stride;
    if (qs && qs->slabs) {
        if (qs->nslabs < 2 || (qs->nslabs & 1)) return fail(XQ_ERR_INVALID_ARGUMENT, "select head slabs: even count >= 2");
        P.q90 = nullptr; P.q_slabs = qs->slabs; P.q_slab_stride = qs->slab_stride; P.q_nslabs = qs->nslabs; P.q_bias = qs->bias;
    }
    P.eps_u32 = eps_u32;
    P.results = results_dev;
    if (replay != nullptr) {
        if (replay->dev.capacity < e->n)
            return fail(XQ_ERR_INVALID_ARGUMENT, "replay capacity %d < n_games %d", replay->dev.capacity, e->n);
        P.rp = replay->dev;
        P.rp_write_base = replay->write_pos;
    }
    const int blocks = (e->n + 3) / 4;
    if (replay != nullptr) XQ_TRY(replay_writer_begin(replay, on ? on : e->stream));    // (costs nothing when the ring's users share this stream)
    if (ev_start || ev_stop) hipExtLaunchKernelGGL(env_kernel<MODE_SELFPLAY>, dim3(blocks), dim3(256), 0, on ? on : e->stream, ev_start, ev_stop, 0, P);
    else hipLaunchKernelGGL(env_kernel<MODE_SELFPLAY>, dim3(blocks), dim3(256), 0, on ? on : e->stream, P);
    XQ_HIP(hipGetLastError());
    if (replay != nullptr) {
        replay->write_pos = (replay->write_pos + e->n) % replay->dev.capacity;
        replay->size = std::min(replay->dev.capacity, replay->size + e->n);
        replay->total += (uint64_t)e->n;
    }
    return XQ_OK;
}

}  // namespace xq

using namespace xq;

// ================================================================================================================
// C ABI — env
// ================================================================================================================
extern "C" {

const char* xq_last_error(void) { return last_error_slot().c_str(); }
int xq_version(void) { return 100; }

int xq_device_count(int* n) {
    if (!n) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(XQ_ERR_NO_DEVICE, "hipGetDeviceCount: %s — libxqhip has no CPU fallback", hipGetErrorString(e)); }
    *n = c;
    return XQ_OK;
}
int xq_set_device(int device) {
    int c = 0;
    XQ_TRY(xq_device_count(&c));
    if (c == 0) return fail(XQ_ERR_NO_DEVICE, "no HIP device: libxqhip has no CPU fallback");
    XQ_HIP(hipSetDevice(device));
    return XQ_OK;
}
int xq_stream_synchronize(void* s) { XQ_HIP(hipStreamSynchronize((hipStream_t)s)); return XQ_OK; }
int xq_event_create(void** ev) {
    hipEvent_t e;
    XQ_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return XQ_OK;
}
int xq_event_destroy(void* ev) { XQ_HIP(hipEventDestroy((hipEvent_t)ev)); return XQ_OK; }
int xq_event_record(void* ev, void* s) { XQ_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)s)); return XQ_OK; }
int xq_event_elapsed_ms(void* a, void* b, float* ms) {
    XQ_HIP(hipEventSynchronize((hipEvent_t)b));
    XQ_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return XQ_OK;
}

static int env_init(xq_env* e, int n_games, uint64_t seed, uint32_t first_game_id, void* hip_stream) {
    e->n = n_games;
    e->seed = seed;
    e->first_id = first_game_id;
    if (hip_stream) e->stream = (hipStream_t)hip_stream;
    else { XQ_HIP(hipStreamCreate(&e->stream)); e->own_stream = true; }
    const size_t n = (size_t)n_games;
    XQ_HIP(hipMalloc(&e->boards, n * kBoardWords * sizeof(uint32_t)));
    XQ_HIP(hipMalloc(&e->meta, n * sizeof(uint4)));
    XQ_HIP(hipMalloc(&e->stats, n * sizeof(uint4)));
    XQ_HIP(hipMalloc(&e->results, n * sizeof(xq_step_result)));
    XQ_HIP(hipMalloc(&e->codes, n * kMaxMoves * sizeof(uint16_t)));
    XQ_HIP(hipMalloc(&e->counts, n * sizeof(int32_t)));
    XQ_HIP(hipMalloc(&e->actions, n * sizeof(int32_t)));
    XQ_HIP(hipMalloc(&e->q90, n * 96 * sizeof(float)));
    XQ_HIP(hipMalloc(&e->validmat, 8 * 8100));        // isValidMove matrix, or the 7 per-piece rule matrices (+ query scratch)
    e->ep_cap = std::max(4096, 4 * n_games);
    XQ_HIP(hipMalloc(&e->ep_ring, (size_t)e->ep_cap * sizeof(xq_episode_record)));
    XQ_HIP(hipMalloc(&e->ep_head, sizeof(unsigned long long)));
    XQ_TRY(upload_start_words(e->stream));
    return xq_env_reset(e);
}

int xq_env_create(int n_games, uint64_t seed, uint32_t first_game_id, void* hip_stream, xq_env** out) {
    if (!out || n_games <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_env_create: n_games must be > 0");
    int c = 0;
    XQ_TRY(xq_device_count(&c));
    if (c == 0) return fail(XQ_ERR_NO_DEVICE, "no HIP device: libxqhip has no CPU fallback");
    xq_env* e = new xq_env();
    const int rc = env_init(e, n_games, seed, first_game_id, hip_stream);
    if (rc != XQ_OK) { xq_env_destroy(e); return rc; }      // a failed allocation must not leak the ones before it
    *out = e;
    return XQ_OK;
}

int xq_env_destroy(xq_env* e) {
    if (!e) return XQ_OK;
    hipStreamSynchronize(e->stream);
    retire_stream(e->stream);        // synchronised above; unconditional: a caller-owned stream may be destroyed right after this call
    hipFree(e->boards); hipFree(e->meta); hipFree(e->stats); hipFree(e->results); hipFree(e->codes);
    hipFree(e->counts); hipFree(e->actions); hipFree(e->q90); hipFree(e->validmat); hipFree(e->ep_ring);
    hipFree(e->ep_head);
    if (e->own_stream) hipStreamDestroy(e->stream);
    delete e;
    return XQ_OK;
}

int xq_env_num_games(const xq_env* e, int* n) {
    if (!e || !n) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    *n = e->n;
    return XQ_OK;
}

int xq_env_reset(xq_env* e) {
    if (!e) return fail(XQ_ERR_INVALID_ARGUMENT, "null env");
    const int total = e->n * kBoardWords;
    hipLaunchKernelGGL(fill_start_kernel, dim3((total + 255) / 256), dim3(256), 0, e->stream, e->boards, e->meta,
                       e->stats, e->n);
    XQ_HIP(hipGetLastError());
    XQ_HIP(hipMemsetAsync(e->ep_head, 0, sizeof(unsigned long long), e->stream));
    e->ep_drained = 0;
    return XQ_OK;
}

int xq_env_set_state(xq_env* e, int first, int n, const uint8_t* boards90, const int32_t* meta4) {
    if (!e || !boards90 || first < 0 || n < 0 || first + n > e->n)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_env_set_state: bad range");
    std::vector<uint32_t> words((size_t)n * kBoardWords);
    std::vector<uint4> meta((size_t)n);
    for (int i = 0; i < n; ++i) {
        for (int s = 0; s < kSquares; ++s)
            if (boards90[(size_t)i * 90 + s] > 14) return fail(XQ_ERR_INVALID_ARGUMENT, "piece code > 14");
        pack_board(boards90 + (size_t)i * 90, &words[(size_t)i * kBoardWords]);
        uint32_t mc = 0, pl = 0, rs = 0, bs = 0;
        if (meta4) {
            mc = (uint32_t)meta4[i * 4 + 0] & 0xFFFFu; pl = (uint32_t)meta4[i * 4 + 1] & 1u;
            rs = (uint32_t)meta4[i * 4 + 2] & 0xFFFFu; bs = (uint32_t)meta4[i * 4 + 3] & 0xFFFFu;
        }
        // META_TRACKED when the scores explain the missing material and the generals (if alive) stand in their own palaces
        int mat[2] = {0, 0}, gen_n[2] = {0, 0};
        bool gen_home = true;
        for (int s = 0; s < kSquares; ++s) {
            const int c = boards90[(size_t)i * 90 + s];
            if (c == 0) continue;
            const int side = c > 7 ? 1 : 0, t = code_type(c);
            static const int value[8] = {0, 1000, 20, 20, 40, 90, 45, 10};
            mat[side] += value[t];
            if (t == T_GENERAL) {
                gen_n[side] += 1;
                const int r = s / 9, col = s % 9;
                gen_home = gen_home && col >= 3 && col <= 5 && (side == 0 ? r <= 2 : r >= 7);
            }
        }
        uint32_t fl = 0;
        if (gen_home && gen_n[0] <= 1 && gen_n[1] <= 1 && mat[0] == kSideMaterial - (int)bs && mat[1] == kSideMaterial - (int)rs)
            fl = META_TRACKED | (gen_n[0] ? META_RED_GENERAL : 0u) | (gen_n[1] ? META_BLACK_GENERAL : 0u);
        meta[i] = make_uint4(mc | (pl << 16) | fl, rs | (bs << 16), 0, 0);
    }
    XQ_HIP(hipStreamSynchronize(e->stream));
    // keep the per-slot RNG counter and episode count
    std::vector<uint4> old((size_t)n);
    XQ_HIP(hipMemcpy(old.data(), e->meta + first, (size_t)n * sizeof(uint4), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) { meta[i].z = old[i].z; meta[i].w = old[i].w; }
    XQ_HIP(hipMemcpy(e->boards + (size_t)first * kBoardWords, words.data(), words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    XQ_HIP(hipMemcpy(e->meta + first, meta.data(), meta.size() * sizeof(uint4), hipMemcpyHostToDevice));
    return XQ_OK;
}

int xq_env_get_state(xq_env* e, int first, int n, uint8_t* boards90, int32_t* meta4) {
    if (!e || first < 0 || n < 0 || first + n > e->n) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_env_get_state: bad range");
    std::vector<uint32_t> words((size_t)n * kBoardWords);
    std::vector<uint4> meta((size_t)n);
    XQ_HIP(hipStreamSynchronize(e->stream));
    XQ_HIP(hipMemcpy(words.data(), e->boards + (size_t)first * kBoardWords, words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    XQ_HIP(hipMemcpy(meta.data(), e->meta + first, meta.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        if (boards90) unpack_board(&words[(size_t)i * kBoardWords], boards90 + (size_t)i * 90);
        if (meta4) {
            meta4[i * 4 + 0] = (int32_t)(meta[i].x & 0xFFFFu);
            meta4[i * 4 + 1] = (int32_t)((meta[i].x >> 16) & 1u);
            meta4[i * 4 + 2] = (int32_t)(meta[i].y & 0xFFFFu);
            meta4[i * 4 + 3] = (int32_t)(meta[i].y >> 16);
        }
    }
    return XQ_OK;
}

int xq_env_legal_moves_dev(xq_env* e, int player, uint16_t* codes_dev, int32_t* counts_dev) {
    if (!e || !codes_dev || !counts_dev || player < -1 || player > 1)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_env_legal_moves: bad argument");
    EnvParams P = base_params(e);
    P.player = player;
    P.codes_out = codes_dev;
    P.counts_out = counts_dev;
    hipLaunchKernelGGL(env_kernel<MODE_LEGAL>, dim3((e->n + 3) / 4), dim3(256), 0, e->stream, P);
    XQ_HIP(hipGetLastError());
    return XQ_OK;
}

int xq_env_legal_moves(xq_env* e, int player, uint16_t* codes_host, int32_t* counts_host) {
    if (!e || !codes_host || !counts_host) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    XQ_TRY(xq_env_legal_moves_dev(e, player, e->codes, e->counts));
    XQ_HIP(hipMemcpyAsync(codes_host, e->codes, (size_t)e->n * kMaxMoves * sizeof(uint16_t), hipMemcpyDeviceToHost, e->stream));
    XQ_HIP(hipMemcpyAsync(counts_host, e->counts, (size_t)e->n * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    XQ_HIP(hipStreamSynchronize(e->stream));
    return XQ_OK;
}

int xq_env_valid_matrix(xq_env* e, int game, uint8_t* valid8100_host) {
    if (!e || !valid8100_host || game < 0 || game >= e->n) return fail(XQ_ERR_INVALID_ARGUMENT, "bad game index");
    hipLaunchKernelGGL(valid_matrix_kernel, dim3(8), dim3(256), 0, e->stream, e->boards, game, e->validmat);
    XQ_HIP(hipGetLastError());
    XQ_HIP(hipMemcpyAsync(valid8100_host, e->validmat, 8100, hipMemcpyDeviceToHost, e->stream));
    XQ_HIP(hipStreamSynchronize(e->stream));
    return XQ_OK;
}

int xq_env_rule_matrix(xq_env* e, int game, uint8_t* rules7x8100_host) {
    if (!e || !rules7x8100_host || game < 0 || game >= e->n) return fail(XQ_ERR_INVALID_ARGUMENT, "bad game index");
    hipLaunchKernelGGL(rule_matrix_kernel, dim3(64), dim3(256), 0, e->stream, e->boards, game, e->validmat, nullptr, 0);
    XQ_HIP(hipGetLastError());
    XQ_HIP(hipMemcpyAsync(rules7x8100_host, e->validmat, 7 * 8100, hipMemcpyDeviceToHost, e->stream));
    XQ_HIP(hipStreamSynchronize(e->stream));
    return XQ_OK;
}

int xq_env_rule_query(xq_env* e, int game, int piece_type, int fr, int fc, int tr, int tc, int* ok) {
    if (!e || !ok || game < 0 || game >= e->n) return fail(XQ_ERR_INVALID_ARGUMENT, "bad game index");
    if (piece_type < T_GENERAL || piece_type > T_SOLDIER) return fail(XQ_ERR_INVALID_ARGUMENT, "piece type must be 1..7");
    const int lim = 1024;
    if (fr < -lim || fr > lim || fc < -lim || fc > lim || tr < -lim || tr > lim || tc < -lim || tc > lim)
        return fail(XQ_ERR_INVALID_ARGUMENT, "coordinate out of the supported range [-1024, 1024]");
    if ((piece_type == T_CHARIOT || piece_type == T_CANNON) && fr == tr && fc == tc)
        return fail(XQ_ERR_UNDEFINED_UPSTREAM, "from == to on a line piece: the upstream path loop overflows (chessboard.cpp:390/:410)");
    const int32_t q[5] = {piece_type, fr, fc, tr, tc};
    int32_t* qd = reinterpret_cast<int32_t*>(e->validmat + 7 * 8100 + 12);      // 4-byte aligned scratch behind the matrices
    uint8_t* od = e->validmat + 7 * 8100;
    XQ_HIP(hipMemcpyAsync(qd, q, sizeof q, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(rule_matrix_kernel, dim3(1), dim3(64), 0, e->stream, e->boards, game, od, qd, 1);
    XQ_HIP(hipGetLastError());
    uint8_t r = 0;
    XQ_HIP(hipMemcpyAsync(&r, od, 1, hipMemcpyDeviceToHost, e->stream));
    XQ_HIP(hipStreamSynchronize(e->stream));
    *ok = r;
    return XQ_OK;
}

int xq_env_get_winner(xq_env* e, int first, int n, uint8_t* winners_host) {
    if (!e || !winners_host || first < 0 || n <= 0 || first + n > e->n) return fail(XQ_ERR_INVALID_ARGUMENT, "bad game range");
    uint8_t* out = reinterpret_cast<uint8_t*>(e->counts);          // n bytes of the [n] int32 scratch
    hipLaunchKernelGGL(winner_kernel, dim3((n + 3) / 4), dim3(256), 0, e->stream, e->boards, first, n, out);
    XQ_HIP(hipGetLastError());
    XQ_HIP(hipMemcpyAsync(winners_host, out, (size_t)n, hipMemcpyDeviceToHost, e->stream));
    XQ_HIP(hipStreamSynchronize(e->stream));
    return XQ_OK;
}

int xq_env_step(xq_env* e, const int32_t* actions_host, int auto_reset, xq_step_result* results_host) {
    if (!e || !actions_host) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    XQ_HIP(hipMemcpyAsync(e->actions, actions_host, (size_t)e->n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    EnvParams P = base_params(e);
    P.actions = e->actions;
    P.auto_reset = auto_reset;
    P.results = e->results;
    hipLaunchKernelGGL(env_kernel<MODE_STEP>, dim3((e->n + 3) / 4), dim3(256), 0, e->stream, P);
    XQ_HIP(hipGetLastError());
    if (results_host)
        XQ_HIP(hipMemcpyAsync(results_host, e->results, (size_t)e->n * sizeof(xq_step_result), hipMemcpyDeviceToHost, e->stream));
    XQ_HIP(hipStreamSynchronize(e->stream));
    return XQ_OK;
}

int xq_env_selfplay_step(xq_env* e, const float* q90_dev, int q_stride, uint32_t eps_u32, xq_step_result* results_dev,
                         xq_replay* replay) {
    if (!e) return fail(XQ_ERR_INVALID_ARGUMENT, "null env");
    if (q90_dev && q_stride < 90) return fail(XQ_ERR_INVALID_ARGUMENT, "q_stride must be >= 90");
    return env_selfplay_launch(e, q90_dev, q_stride, eps_u32, results_dev, replay);
}

int xq_env_selfplay_step_host(xq_env* e, const float* q90_host, uint32_t eps_u32, xq_step_result* results_host) {
    if (!e) return fail(XQ_ERR_INVALID_ARGUMENT, "null env");
    const float* qd = nullptr;
    if (q90_host) {
        // host rows are 90 floats; device rows are padded to 96
        XQ_HIP(hipMemcpy2DAsync(e->q90, 96 * sizeof(float), q90_host, 90 * sizeof(float), 90 * sizeof(float), (size_t)e->n,
                                hipMemcpyHostToDevice, e->stream));
        qd = e->q90;
    }
    XQ_TRY(env_selfplay_launch(e, qd, 96, eps_u32, e->results, nullptr));
    if (results_host)
        XQ_HIP(hipMemcpyAsync(results_host, e->results, (size_t)e->n * sizeof(xq_step_result), hipMemcpyDeviceToHost, e->stream));
    XQ_HIP(hipStreamSynchronize(e->stream));
    return XQ_OK;
}

int xq_env_drain_episodes(xq_env* e, xq_episode_record* records_host, int max_records, int* n_out, uint64_t* total) {
    if (!e || !n_out) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    unsigned long long head = 0;
    XQ_HIP(hipStreamSynchronize(e->stream));
    XQ_HIP(hipMemcpy(&head, e->ep_head, sizeof head, hipMemcpyDeviceToHost));
    if (total) *total = head;
    uint64_t begin = e->ep_drained;
    if (head - begin > (uint64_t)e->ep_cap) begin = head - (uint64_t)e->ep_cap;   // older records were overwritten
    int n = 0;
    while (begin < head && n < max_records && records_host) {
        const uint64_t idx = begin % (uint64_t)e->ep_cap;
        const uint64_t run = std::min<uint64_t>({head - begin, (uint64_t)e->ep_cap - idx, (uint64_t)(max_records - n)});
        XQ_HIP(hipMemcpy(records_host + n, e->ep_ring + idx, run * sizeof(xq_episode_record), hipMemcpyDeviceToHost));
        n += (int)run;
        begin += run;
    }
    e->ep_drained = begin;
    *n_out = n;
    return XQ_OK;
}

int xq_env_counters(xq_env* e, uint64_t c[6]) {
    if (!e || !c) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    std::vector<uint4> meta((size_t)e->n), stats((size_t)e->n);
    XQ_HIP(hipStreamSynchronize(e->stream));
    XQ_HIP(hipMemcpy(meta.data(), e->meta, meta.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    XQ_HIP(hipMemcpy(stats.data(), e->stats, stats.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) c[i] = 0;
    for (int i = 0; i < e->n; ++i) {
        c[0] += meta[i].z; c[1] += meta[i].w;
        c[2] += stats[i].x; c[3] += stats[i].y; c[4] += stats[i].z; c[5] += stats[i].w;
    }
    return XQ_OK;
}

const uint32_t* xq_env_boards_dev(const xq_env* e) { return e ? e->boards : nullptr; }
const uint32_t* xq_env_meta_dev(const xq_env* e) { return e ? (const uint32_t*)e->meta : nullptr; }

}  // extern "C"
