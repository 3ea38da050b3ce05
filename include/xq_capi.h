/* xq_capi.h — C ABI of libxqhip.so, the MI355X-native (gfx950) batched Xiangqi self-play + DQN hot path.
 *
 * This is the drop-in boundary for the training path of Qervas/cn_chess_ai (reference @ 2024-10-20).  The reference
 * has no FFI layer: its boundary is the C++ class surface ChessBoard / ChessAI / DQN / NeuralNetwork.  Every entry
 * point below names the reference interface it replaces (file:line into the reference tree); the C++ facade in
 * include/xq/ rebuilds those classes on top of this ABI, and INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *   - opaque handles, plain pointers and sizes, no C++ or torch types;
 *   - every function returns an xq_status (0 = ok) and never throws; xq_last_error() gives the message of the last
 *     failure on the calling thread;
 *   - pointers named *_host are caller-owned host memory, *_dev are device (HBM) pointers on the handle's device;
 *   - each handle runs on ONE hipStream_t (passed as void*; NULL = a stream the handle creates); calls are
 *     asynchronous on that stream unless they return data to the host, in which case they synchronise that stream;
 *   - handles that exchange DEVICE data are ordered by stream order when they were created on the same stream — what xq_trainer
 *     does, and the simplest way to use the ABI.  For handles on streams of their own see "Stream ordering" below: whatever is passed
 *     as a HANDLE the library orders itself, whatever is passed as a raw device pointer is the caller's to order;
 *   - handles are not thread-safe; one GPU per process (one rank per GPU under torch.distributed / RCCL).
 *
 * Stream ordering (no upstream analogue: the reference has one stream and a cudaDeviceSynchronize() after every launch).  Every pair of
 * entry points through which two handles hand each other device data, and what orders the consumer behind the producer when the two
 * handles run on different streams (same stream: stream order, nothing recorded).  "library" = the ring keeps, per resource, the
 * stream of the last write and of the reads since, and the consumer's stream waits on an event recorded on the producer's stream
 * (cn_chess_ai_amd/csrc/xq_internal.h, SharedResource); each class can be switched off with xq_debug_set_stream_ordering for the test
 * that shows it is needed (tests/test_stream_order_gpu.py: every row below has a sync-vs-nosync case that goes red without it).
 *
 *   producer -> consumer                                         device data                       ordered by
 *   xq_env_selfplay_step(replay) -> xq_dqn_td_grads_replay        ring slots (s, a, r, s', done)    library: XQ_ORDER_RING_CONTENTS
 *   xq_dqn_td_grads_replay -> xq_env_selfplay_step(replay)        the slots that step still reads   library: XQ_ORDER_RING_CONTENTS
 *   xq_env_selfplay_step(replay) -> xq_replay_get                 ring slots                        library: XQ_ORDER_RING_CONTENTS
 *   xq_env_selfplay_step(replay) -> xq_replay_per_rebuild         priorities of new transitions     library: XQ_ORDER_RING_PRIORITIES
 *   xq_dqn_td_grads_replay (prioritized) -> xq_replay_per_rebuild TD-error priorities               library: XQ_ORDER_RING_PRIORITIES
 *   xq_replay_per_rebuild -> xq_env_selfplay_step(replay)         maximum-priority snapshot         library: XQ_ORDER_RING_PRIORITIES
 *   xq_replay_per_rebuild -> xq_replay_sample_prioritized         sum tree                          one stream (the ring's): stream order
 *   xq_replay_sample* -> xq_dqn_td_grads_replay                   slot list, importance weights     library: XQ_ORDER_RING_DRAW
 *   xq_dqn_td_grads_replay -> next xq_replay_sample*              the list that step still reads    library: XQ_ORDER_RING_DRAW
 *   xq_dqn_apply_grads / set_params / load_model / update_target
 *     -> xq_dqn_forward* / select_q_dev / td_grads*               parameters                        one handle, one stream: stream order
 *     -> xq_trainer_collect (collect stream of overlap_collect)   parameters                        library: XQ_ORDER_TRAINER_PARAMS
 *   xq_dqn_select_q_dev / forward_boards_dev -> xq_env_selfplay_step   q90_dev (raw pointer)        CALLER: xq_stream_wait_stream(env stream, Q-net stream)
 *   xq_env_*step* -> xq_dqn_select_q_dev / forward_boards_dev / td_grads(boards_dev)  boards (raw pointer)   CALLER: xq_stream_wait_stream(Q-net stream, env stream)
 *   xq_env_selfplay_step(results_dev) / legal_moves_dev -> anything    raw pointers                 CALLER
 *   xq_comm_allreduce(buf_dev, stream), xq_allreduce_grads        caller's buffer / gradient buffer on the stream given / the handle's: stream order
 *   host-buffer entry points (*_host, set_state, get_*, push_host, set_priorities, set_params, backpropagate)   synchronise before they return
 * A stream handed to a handle must outlive every handle that exchanged data with it (events are recorded on it lazily).
 *
 * Encodings
 *   - square  s = row*9 + col, rows 0..9 (Red home rows 0-4, Red moves +row: chessboard.cpp:12-28), cols 0..8;
 *   - piece code: 0 empty, 1..7 = Red General,Advisor,Elephant,Horse,Chariot,Cannon,Soldier, 8..14 = Black
 *     (PieceType/PieceColor, chessboard.h:8-14; code-1 is the one-hot plane of chessai.cpp:278-282);
 *   - action code = from*90 + to  (Action{from,to}, action.h:4-11);
 *   - colours: 0 Red, 1 Black, 2 None (PieceColor, chessboard.h:12-14).
 */
#ifndef XQ_CAPI_H
#define XQ_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    XQ_OK = 0,
    XQ_ERR_INVALID_ARGUMENT = 1,  /* std::invalid_argument upstream (dqn.cu:17-19,200,324-329) */
    XQ_ERR_RUNTIME = 2,           /* std::runtime_error upstream: device error (dqn.h:16-24), empty action list (dqn.cpp:26-28) */
    XQ_ERR_NO_DEVICE = 3,         /* no gfx950 device / HIP runtime unusable: the product path has NO CPU fallback */
    XQ_ERR_IO = 4,                /* model file open/read/write failure or layer mismatch (dqn.cpp:80,115,147) */
    XQ_ERR_UNDEFINED_UPSTREAM = 5 /* bug-compatible backprop requested for a topology where the reference reads out of bounds */
} xq_status;

enum { XQ_MAX_MOVES = 128, XQ_BOARD_WORDS = 12, XQ_MAX_LAYERS = 8 };

const char* xq_last_error(void);
int xq_version(void);
/* Device plumbing (replaces the implicit cudaMalloc/cudaDeviceSynchronize device of dqn.cu). */
int xq_device_count(int* n);
int xq_set_device(int device);
int xq_stream_synchronize(void* hip_stream);
/* A HIP stream for callers that have no HIP headers of their own (the ctypes binding, the C++ facade): priority -1 urgent, 0 normal,
 * 1 background (streams of different priority never share a hardware queue; streams of one priority may, and then run in submission
 * order); nonblocking != 0: not ordered against the legacy null stream.  Destroy it after the handles that run on it. */
int xq_stream_create(int priority, int nonblocking, void** hip_stream);
int xq_stream_destroy(void* hip_stream);
/* Orders everything queued on waiting_stream from now on behind everything queued on producer_stream so far (one event record + one
 * wait; no host synchronisation) — for the device pointers the caller hands from one handle to another ("Stream ordering" above). */
int xq_stream_wait_stream(void* waiting_stream, void* producer_stream);
/* idle = 1 when everything queued on the stream so far has completed (hipStreamQuery), without blocking. */
int xq_stream_query(void* hip_stream, int* idle);
/* Diagnostics of the ordering tests — REFUSED (XQ_ERR_INVALID_ARGUMENT) unless the process runs with XQ_DEBUG_API=1 in its environment:
 * they stall streams and switch the library's synchronisation off.
 * xq_debug_stream_gate queues a one-wave kernel on hip_stream that waits until the HOST calls xq_debug_gate_release (or timeout_ms, 1..5000,
 * have passed: the kernel always ends): whatever is queued behind it is late by construction, not by a guess about durations.
 * xq_debug_gate_destroy after the stream has been synchronised.  xq_debug_stream_delay queues a kernel that spins for `microseconds`
 * (<= 200000).  xq_debug_set_stream_ordering: bit mask of the ordering classes the library provides (default XQ_ORDER_ALL);
 * process-wide; setting it back to XQ_ORDER_ALL is always allowed. */
enum { XQ_ORDER_RING_CONTENTS = 1, XQ_ORDER_RING_PRIORITIES = 2, XQ_ORDER_RING_DRAW = 4, XQ_ORDER_TRAINER_PARAMS = 8, XQ_ORDER_ALL = 15 };
int xq_debug_stream_gate(void* hip_stream, int timeout_ms, void** gate);
int xq_debug_gate_release(void* gate);
int xq_debug_gate_destroy(void* gate);
int xq_debug_stream_delay(void* hip_stream, int microseconds);
int xq_debug_set_stream_ordering(unsigned mask);
/* HIP-event timing on a given stream, for bench.py's roofline leg (ms between the two records). */
int xq_event_create(void** ev);
int xq_event_destroy(void* ev);
int xq_event_record(void* ev, void* hip_stream);
int xq_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);

/* ------------------------------------------------------------------------------------------------------------
 * xq_env — thousands of ChessBoard instances resident in HBM (chessboard.h:35-78), one wavefront per board.
 * HBM layout per game: board = 12 x u32 (90 squares x 4 bit), meta = 4 x u32
 *   {moveCount | player<<16, redScore | blackScore<<16, plies played by this slot (RNG counter), episodes finished}.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct xq_env xq_env;

/* ChessBoard::ChessBoard() x n_games (chessboard.cpp:4-6).  game ids are first_game_id .. +n_games-1 (RNG streams). */
int xq_env_create(int n_games, uint64_t seed, uint32_t first_game_id, void* hip_stream, xq_env** out);
int xq_env_destroy(xq_env* env);
int xq_env_num_games(const xq_env* env, int* n);
int xq_env_stream(const xq_env* env, void** hip_stream);      /* the stream the handle runs on (its own when it was created with NULL) */
/* ChessBoard::reset() on every game (chessboard.cpp:95-102); also zeroes the RNG counters and episode counts. */
int xq_env_reset(xq_env* env);
/* Test/interop access to the state (no upstream analogue: upstream board is private).  boards90: [n][90] piece codes;
 * meta4: [n][4] = moveCount, currentPlayer, redScore, blackScore (chessboard.h:62-77). */
int xq_env_set_state(xq_env* env, int first, int n, const uint8_t* boards90_host, const int32_t* meta4_host);
int xq_env_get_state(xq_env* env, int first, int n, uint8_t* boards90_host, int32_t* meta4_host);
/* ChessAI::getAllValidActions(player) for every game (chessai.cpp:347-368 over chessboard.cpp:112-283): canonical
 * order (from ascending, then generator order).  player: 0 Red, 1 Black, -1 = each game's side to move.
 * codes: [n_games][128] u16 action codes, counts: [n_games]. */
int xq_env_legal_moves(xq_env* env, int player, uint16_t* codes_host, int32_t* counts_host);
int xq_env_legal_moves_dev(xq_env* env, int player, uint16_t* codes_dev, int32_t* counts_dev);
/* ChessBoard::getWinner() (chessboard.cpp:312-320) for games first..first+n-1: colour of the first general in index order
 * (so Red while both are alive), 2 if none. */
int xq_env_get_winner(xq_env* env, int first, int n, uint8_t* winners_host);
/* ChessBoard::isValidMove for all 90x90 (from,to) pairs of game g (chessboard.cpp:66-93,328-440): valid8100[f*90+t]. */
int xq_env_valid_matrix(xq_env* env, int game, uint8_t* valid8100_host);

/* The seven public per-piece validators isValid{General,Advisor,Elephant,Horse,Chariot,Cannon,Soldier}Move
 * (chessboard.h:50-56, chessboard.cpp:328-440) of game g, evaluated on device as upstream writes them (geometry + occupancy
 * only; the piece on `from` is not consulted except for the soldier's colour): rules[(type-1)*8100 + f*90 + t] for all in-board
 * (from, to) pairs, type = PieceType 1..7.  xq_env_rule_query answers one call with arbitrary coordinates (upstream reads
 * squares outside the board as Empty); from == to on a chariot/cannon overflows a loop upstream => XQ_ERR_UNDEFINED_UPSTREAM. */
int xq_env_rule_matrix(xq_env* env, int game, uint8_t* rules7x8100_host);
int xq_env_rule_query(xq_env* env, int game, int piece_type, int from_row, int from_col, int to_row, int to_col, int* ok);

/* Per-game result of one ply.  Mirrors what chessai.cpp:113-119,146-162 derives after movePiece(). */
typedef struct {
    int32_t action;      /* action code played, -1 if the side to move had no action (chessai.cpp:100-103) */
    int32_t n_moves;     /* size of the legal list the action was chosen from (0 for xq_env_step) */
    int32_t reward;      /* evaluateBoard(mover, post-move count), chessai.cpp:311-345 */
    uint8_t captured;    /* piece code movePiece() returned (chessboard.cpp:38-64); 0 = none or invalid move */
    uint8_t valid;       /* 0: movePiece() rejected the move — no state change */
    uint8_t done;        /* checkGameOver() || moveCount+1 >= 200 (chessai.cpp:119) */
    uint8_t terminated;  /* episode ended: checkGameOver() (chessboard.cpp:286-309) or no action; board was auto-reset */
    uint8_t winner;      /* getWinner() (chessboard.cpp:312-320) when terminated, else 2 */
    uint8_t explored;    /* epsilon branch taken (dqn.cpp:31-34) */
    uint16_t move_count; /* getMoveCount() after the move (before auto-reset) */
    int16_t red_score, black_score; /* getRedScore()/getBlackScore() after the move (before auto-reset) */
} xq_step_result;

/* ChessBoard::movePiece with caller-chosen actions (chessboard.cpp:38-64) + evaluateBoard + checkGameOver, per game.
 * actions: [n_games] action codes (any int; out-of-board / invalid => rejected, no state change).
 * auto_reset != 0: a game whose checkGameOver() became true BY THIS MOVE is reset (the `for episode` loop of chessai.cpp:89-90).
 * A rejected move changes nothing — no counters, no episode record, no reset — also on an already finished board
 * (its result still reports terminated/winner: that is how the facade's checkGameOver() probes). */
int xq_env_step(xq_env* env, const int32_t* actions_host, int auto_reset, xq_step_result* results_host);

/* One self-play ply for every game, fully on device: legal moves (LDS) -> epsilon-greedy DQN::selectAction
 * (dqn.cpp:24-56; q90_dev = tanh Q-values of outputs 0..89 per game, row stride q_stride floats; NULL = uniform random
 * policy) -> movePiece -> evaluateBoard -> done -> auto-reset.  rand() is replaced by Philox4x32-10
 * (ctr = {ply counter of the slot, 0, game id, 0}, key = seed): explore iff r0 < eps_u32, index = r1 % n.
 * results_dev: optional [n_games] xq_step_result in HBM; replay: optional ring that receives the transition. */
typedef struct xq_replay xq_replay;
int xq_env_selfplay_step(xq_env* env, const float* q90_dev, int q_stride, uint32_t eps_u32,
                         xq_step_result* results_dev, xq_replay* replay);
/* Convenience for tests: same, copying results to the host. */
int xq_env_selfplay_step_host(xq_env* env, const float* q90_host, uint32_t eps_u32, xq_step_result* results_host);

/* Episode reporting — the gameCompleted(game, redScore, blackScore) signal (chessai.h:35, chessai.cpp:162). */
typedef struct {
    uint32_t game_id, episode;   /* episode = 1-based count for that game slot (signal's `gameNumber`) */
    int16_t red_score, black_score;
    uint16_t move_count;
    uint8_t winner, reserved;
} xq_episode_record;
/* Drains up to max_records finished-episode records (oldest first); *n_out = number written, *total = episodes so far. */
int xq_env_drain_episodes(xq_env* env, xq_episode_record* records_host, int max_records, int* n_out, uint64_t* total);
/* totals since create/reset: [0] plies, [1] episodes, [2] red wins, [3] black wins, [4] captures, [5] explored plies */
int xq_env_counters(xq_env* env, uint64_t counters6_host[6]);
const uint32_t* xq_env_boards_dev(const xq_env* env);   /* [n_games][12] packed boards in HBM */
const uint32_t* xq_env_meta_dev(const xq_env* env);     /* [n_games][4] */

/* ------------------------------------------------------------------------------------------------------------
 * xq_replay — ReplayBuffer of N7's 5-tuple (state, action.to, reward, nextState, done), dqn.cpp:157-172.
 * Not in the reference (SURVEY §8b "New"): ring in HBM, states stored as packed boards (48 B), never as 1260 floats.
 * ---------------------------------------------------------------------------------------------------------- */
int xq_replay_create(int capacity, uint64_t seed, void* hip_stream, xq_replay** out);
int xq_replay_destroy(xq_replay* r);
int xq_replay_size(xq_replay* r, int* size, int* capacity, uint64_t* total_pushed);
int xq_replay_stream(const xq_replay* r, void** hip_stream);
/* push(s, a, r, s', done) for n transitions given as 90-byte boards on the host (tests / interop). */
int xq_replay_push_host(xq_replay* r, int n, const uint8_t* boards90, const int32_t* action_to, const float* reward,
                        const uint8_t* done, const uint8_t* next_boards90);
/* sample(B): uniform with replacement, Philox(ctr = {draw, 0, sample call #, 1}, key = seed) % size.
 * Returns the chosen slots; xq_dqn_td_grads_replay consumes them on device.  Streams: the draw runs on the ring's stream, the
 * consumer on the Q-net's, env steps that fill the ring on the env's.  When those differ the library orders them itself ("Stream
 * ordering" at the top) — the TD step waits for the draw and for the env steps that wrote the ring, the next draw (it overwrites the
 * slot list) and the next env step (it overwrites slots) wait for the TD step that still reads them — so a caller may draw, queue the
 * step, play and draw again without synchronising.  (xq_trainer orders its own ring by hand: its collects write slots that the TD
 * step running beside them never samples, xq_replay_sample_window.) */
int xq_replay_sample(xq_replay* r, int batch, int32_t* slots_host /* optional */);
/* sample(B) restricted to the `count` ring slots that start at `start` (wrapping): slot = (start + Philox % count) % capacity.
 * The overlapped trainer uses it to leave out the slots a concurrent collect is writing; (0, size) == xq_replay_sample. */
int xq_replay_sample_window(xq_replay* r, int batch, int start, int count, int32_t* slots_host /* optional */);
int xq_replay_get(xq_replay* r, int slot, uint8_t* board90, int32_t* action_to, float* reward, uint8_t* done,
                  uint8_t* next_board90);
/* Prioritized replay, proportional variant (Schaul et al., ICLR 2016) — build-defined (BASELINE configs[4]), defined by
 * DESIGN.md §4:  p_i = (|TD error_i| + eps)^alpha, P(i) = p_i / sum p, stratified draw of one sample per segment of
 * the total mass (Philox ctr = {k, 0, sample call #, 2}), importance weight w_i = (N * P(i))^-beta / max_batch w applied to the
 * sample's gradient, new transitions enter with the largest priority assigned so far.  The masses live in a radix-32 sum tree in
 * HBM whose nodes are sequential fp32 sums (fixed association: sampled slots are reproducible bit for bit).
 *   xq_replay_enable_per      : once, before use (allocates the priority table and the tree)
 *   xq_replay_per_rebuild     : zeroes the priorities of `retire_count` slots from `retire_start` (wrapping; slots about to be
 *                               overwritten), rebuilds the tree from the priority table and snapshots the running maximum
 *   xq_replay_sample_prioritized: draws `batch` slots from the tree as of the last rebuild; xq_dqn_td_grads_replay then applies the
 *                               importance weights and writes the new priorities of the sampled slots back
 *   set/get_priorities, per_stats: tests / interop
 * Call order: rebuild -> sample_prioritized -> td_grads_replay.  A rebuild between a draw and its TD step is allowed (the batch maximum
 * of the draw's importance weights is left alone until the step has been queued). */
int xq_replay_enable_per(xq_replay* r, double alpha, double beta, double eps);
int xq_replay_per_rebuild(xq_replay* r, int retire_start, int retire_count);
int xq_replay_sample_prioritized(xq_replay* r, int batch, int32_t* slots_host /* optional */, float* weights_host /* optional, normalised */);
int xq_replay_set_priorities(xq_replay* r, int first, int n, const float* prio_host);
int xq_replay_get_priorities(xq_replay* r, int first, int n, float* prio_host);
int xq_replay_per_stats(xq_replay* r, float* total, float* max_priority, int* n_eligible);

/* ------------------------------------------------------------------------------------------------------------
 * xq_dqn — DQN (dqn.h:97-116) = qNetwork + targetNetwork (NeuralNetwork, dqn.h:42-95, dqn.cu), fp32 on MFMA.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct xq_dqn xq_dqn;

enum { XQ_NET_ONLINE = 0, XQ_NET_TARGET = 1 };
enum { XQ_BACKPROP_REFERENCE = 0,  /* bug-compatible hidden delta, dqn.cu:406-423 as written */
       XQ_BACKPROP_TEXTBOOK = 1 };
enum { XQ_TD_ONLINE_NET = 0,       /* max Q(s') from the ONLINE net — what ChessAI::train does (chessai.cpp:126-127) */
       XQ_TD_TARGET_NET = 1,       /* max Q(s') from the target net — DQN::train (dqn.cpp:166-167) */
       XQ_TD_DOUBLE = 2 };         /* Double DQN (build-defined, BASELINE configs[4]): a* = argmax_k Q_online(s')[k] over all outputs
                                    * (first maximum), y = r + gamma * Q_target(s')[a*] */
enum { XQ_PRECISION_F32 = 0,       /* fp32 MFMA everywhere (the reference computes in fp64; north_star: fp32, Q within 1e-4) */
       XQ_PRECISION_BF16 = 1,      /* bf16 Q-net (build-defined, BASELINE configs[4]): forward passes on bf16 MFMA — weights and hidden
                                    * activations rounded to bf16 (RNE), fp32 accumulation, biases and outputs fp32; backward and SGD
                                    * in fp32 on the master weights */
       XQ_PRECISION_BF16_FULL = 2 };/* the same forward, and the dense products of the backward pass on bf16 MFMA too: the lower hidden deltas
                                    * take the bf16 weights and the upstream delta rounded to bf16, the hidden weight gradients the delta
                                    * rounded to bf16 and the (already bf16) activations; fp32 accumulation, fp32 deltas for the bias /
                                    * layer-0 / output-layer gradients, fp32 master weights and SGD (defined in DESIGN.md section 4) */

enum { XQ_QMAX_FULL = 0,           /* max_a' Q(s',a') of the TD target (chessai.cpp:126-127, dqn.cpp:166-167): every output in fp32 */
       XQ_QMAX_SCREENED = 1 };     /* the same fp32 maximum, found by exact screening: all outputs once on the bf16 matrix pipe with a
                                    * rigorous error bound, then only the outputs within the bound of the screened maximum again in
                                    * fp32.  The value returned is the maximum of fp32-evaluated outputs. */

/* DQN::DQN(layerSizes, lr, gamma) (dqn.cpp:12-20) -> NeuralNetwork ctor (dqn.cu:14-57): W ~ U(-0.05,0.05) from a
 * seeded generator, biases 0; target = copy of online.  layer_sizes[0] must be 1260 for the board-input fast path;
 * any sizes work through the dense-state entry points. */
int xq_dqn_create(const int* layer_sizes, int n_sizes, double learning_rate, double gamma, uint64_t seed,
                  void* hip_stream, xq_dqn** out);
int xq_dqn_destroy(xq_dqn* d);
int xq_dqn_stream(const xq_dqn* d, void** hip_stream);
/* XQ_PRECISION_*: arithmetic of the forward passes on packed boards (action select, TD targets, Q(s,a)).  The dense-state entry
 * points (xq_dqn_forward / xq_dqn_backpropagate: the reference's std::vector<double> API) always compute in fp32. */
int xq_dqn_set_precision(xq_dqn* d, int precision);
/* Layer 0 of the s' chain of a TD step (online TD rule, fp32 net; no upstream analogue — the reference evaluates every state from
 * scratch, dqn.cu:199-260).  0 (default): the sum over the occupied squares of s' in ascending square order, the reference's
 * i-ascending accumulation with the zeros skipped.  1: z1(s') = z1(s) - rows of the squares that changed + rows of what stands there
 * now (two squares for a move; boards more than 8 squares apart are gathered in full): one gather instead of two, a DIFFERENT
 * summation order (differences ~1e-7 on the activations, far inside the 1e-4 budget on Q).  bench.py switches it on.
 * The same switch covers the select chain of the self-play loop (xq_dqn_select_q_dev / xq_trainer_collect, fp32 net): its layer-0
 * sums are kept per game, and while the online parameters do not change — the plies of one update — the next ply's sums are
 * derived from the kept ones the same way (used when an update period has several plies; a game that ended is summed in full). */
int xq_dqn_set_l0_derive(xq_dqn* d, int on);
/* How the gradient half of xq_dqn_td_grads is queued (no upstream analogue: dqn.cu:323-467 launches one kernel per layer and waits
 * for each).  1 (default): as fused launches on the handle's stream — per hidden layer below the top one ONE grid that holds the blocks
 * of the delta product, of the weight-gradient product above it and (first time) of the output-layer sums, then ONE grid with the
 * layer-0 sums and the bias column sums.  Taken for every fp32 net whose backward products fit 64 x 64 tiles (all BASELINE nets at
 * 8192-16384 samples); bf16 nets and larger products run the same kernels one by one on two streams (critical chain + side stream,
 * three event records and a join), as does 0.  What happens to the partial sums the fused launches leave: with
 * xq_dqn_set_fused_apply on and no communicator they stay pending and xq_dqn_apply_grads adds them; otherwise ONE launch reduces them
 * into the gradient buffer on the handle's stream, and an attached communicator all-reduces that buffer in ONE collective behind it
 * (xq_dqn_set_comm).  The same switch covers the other fusion of the step's second half: with exact screening on a net whose last
 * hidden layer is 256 wide (uniform replay), the TD target / output delta / top hidden delta are computed inside the blocks of the
 * kernel that re-evaluates the screened maxima instead of by a launch of their own.  Results are bitwise identical either way. */
int xq_dqn_set_td_tail(xq_dqn* d, int on);
/* How the layer-0 weight gradient of a TD step (updateWeightsBiasesKernel on layer 0, dqn.cu:310-319, fed by the one-hot of
 * chessai.cpp:268-289) is computed.  1 (default; first hidden width a multiple of 64 and >= 256 samples, other shapes take 0): the dense
 * product one-hot^T x delta_0 on the bf16 matrix pipe, exact — the one-hot operand is 0 / 1 and delta_0 is split into three bf16 values per
 * fp32 (hi + mid + lo, every residual exact), products exact, fp32 accumulation.  0: per-(square, piece) segmented sums of delta rows on
 * the vector ALU.  Same value up to the summation order (both within a few fp32 ulp of an fp64 evaluation).  Measured at 8192 x 256
 * (round 5): 17 against 37 us alone, the headline step 0.1755 against 0.1833 ms. */
int xq_dqn_set_l0_grad_mode(xq_dqn* d, int mode);
/* Exact screening, second pass: groups whose two largest screened values both reach the threshold are re-evaluated WHOLE (32 rows).  Nets whose
 * trained rows dominate can ask for one such group for nearly every sample; the refine kernel then serves the groups that many samples of a block
 * share from LDS (activation rows of the block + the rows of up to three groups staged in one sweep; 145 KB of LDS, so only for launches of at most
 * one block per CU and a 256-wide last hidden layer).  mode -1 (default): on while the screen's own counters — read back every 32 screened steps —
 * show more than 0.25 whole groups per sample, off below 0.10; 0: never; 1: whenever the launch has room.  A speed switch: same bits either way. */
int xq_dqn_set_refine_stage(xq_dqn* d, int mode);
/* XQ_QMAX_*: how the TD step finds max_a' Q(s',a').  XQ_QMAX_SCREENED applies to fp32 nets with XQ_TD_ONLINE_NET / XQ_TD_TARGET_NET
 * whose last hidden width is a multiple of 64 and whose product is large enough for the persistent GEMM (>= 512 tiles of 128 x 128);
 * every other case silently keeps the full fp32 product.  Guard: every 32 screened steps the candidate counters are read back
 * asynchronously; when the screen leaves more than 24 candidate groups (or 1 whole group) per sample — a net whose outputs all lie
 * within the bf16 bound of each other — the next 512 TD steps run the full product, then the screen is tried again.  stats[4]
 * (xq_dqn_qmax_stats, synchronises): TD steps screened, samples, candidate (sample, 32-output group) pairs re-evaluated in fp32,
 * pairs whose whole group was re-evaluated. */
int xq_dqn_set_qmax_mode(xq_dqn* d, int mode);
int xq_dqn_qmax_stats(xq_dqn* d, uint64_t stats[4]);
/* The guard's own counters: how often it has switched a run of TD steps to the full product so far, and how many steps of the current
 * run are left (0 = the screen is on).  No synchronisation. */
int xq_dqn_qmax_guard(xq_dqn* d, uint64_t* fallbacks, int* hold_steps_left);
int xq_dqn_num_params(const xq_dqn* d, size_t* n_weights, size_t* n_biases);
/* host_weights / host_biases in the REFERENCE flat layout (row-major [out][in] per layer, layers concatenated,
 * dqn.cu:112-140), fp64 like upstream.  set = copyToDevice() (dqn.cu:480-485), get = copyFromDevice() (:487-492). */
int xq_dqn_set_params(xq_dqn* d, int which_net, const double* weights_host, const double* biases_host);
int xq_dqn_get_params(xq_dqn* d, int which_net, double* weights_host, double* biases_host);
/* DQN::getQValues / NeuralNetwork::forward (dqn.cpp:65-68, dqn.cu:199-260) for n dense states [n][layer_sizes[0]]
 * -> q [n][layer_sizes[last]], fp64 at the boundary like upstream (computed in fp32). */
int xq_dqn_forward(xq_dqn* d, int which_net, const double* states_host, int n, double* q_host);
/* Same for n packed boards already in HBM ([n][12] u32): the one-hot of chessai.cpp:268-289 is never materialised.
 * q_dev: [n][ldq] fp32, first `n_out` outputs per row (n_out <= layer_sizes[last]). */
int xq_dqn_forward_boards_dev(xq_dqn* d, int which_net, const uint32_t* boards_dev, int n, int n_out,
                              float* q_dev, int ldq);
/* Q(s)[0..95] of the online net for n packed boards in HBM, through the SELECT chain of the self-play loop (what xq_trainer_collect
 * feeds the env kernel; DQN::selectAction reads q[action.to] only, dqn.cpp:47) -> q_dev [n][96] fp32.  Differs from
 * xq_dqn_forward_boards_dev(.., n_out = 96, ..) only in how it gets there: from 2048 boards on the head rides on the last hidden
 * product, and with xq_dqn_set_l0_derive the layer-0 sums of these boards are KEPT — while the online parameters do not change
 * (the plies of one update), the next call for the same board array derives them from the kept ones (rows of the squares that
 * changed out, rows of what stands there now in; boards more than 8 squares apart are summed in full). */
int xq_dqn_select_q_dev(xq_dqn* d, const uint32_t* boards_dev, int n, float* q_dev);
/* DQN::backpropagate(state, target, lr) / NeuralNetwork::backpropagate (dqn.cpp:59-62, dqn.cu:323-467) for a batch of
 * n (state, target) pairs: gradients of all n samples are taken at the pre-update weights, summed, scaled by
 * grad_scale and applied once (n = 1, grad_scale = 1 is exactly the upstream call).  mode = XQ_BACKPROP_*. */
int xq_dqn_backpropagate(xq_dqn* d, const double* states_host, const double* targets_host, int n,
                         double learning_rate, double grad_scale, int mode);
/* DQN::updateTargetNetwork() (dqn.cpp:71-73) — copies the TRAINED device weights (upstream copies stale host
 * vectors; documented divergence). */
int xq_dqn_update_target(xq_dqn* d);
/* DQN::saveModel / loadModel (dqn.cpp:76-154): raw LE fp64 weights, raw fp64 biases, BE u64 count, BE i32 sizes. */
int xq_dqn_save_model(xq_dqn* d, const char* path);
int xq_dqn_load_model(xq_dqn* d, const char* path);

/* The TD step of ChessAI::train (chessai.cpp:122-131) / DQN::train (dqn.cpp:157-172) on a batch of transitions held
 * in HBM:  y = done ? r : r + gamma * max_k Q_net(s')[k]  (max over ALL outputs);  target = Q(s) with entry
 * action.to replaced by y;  backprop of 0.5*|Q(s)-target|^2.  Split in two so a multi-GPU caller can all-reduce
 * the gradient buffer in between:
 *   xq_dqn_td_grads  : forward + deltas + gradient reduction over the batch  -> compact gradient buffer (HBM)
 *   xq_dqn_apply_grads: params -= lr * grad_scale * grads
 * boards/next_boards: [n][12] u32, optionally gathered through slots_dev ([n] row indices, NULL = identity).
 * td_net = XQ_TD_*.  loss_out_dev: optional, sum over the batch of 0.5*(Q(s,a)-y)^2. */
int xq_dqn_td_grads(xq_dqn* d, const uint32_t* boards_dev, const uint32_t* next_boards_dev,
                    const int32_t* action_to_dev, const float* reward_dev, const uint8_t* done_dev,
                    const int32_t* slots_dev, int n, int td_net, int mode);
int xq_dqn_apply_grads(xq_dqn* d, double learning_rate, double grad_scale);
/* The flat gradient buffer xq_dqn_td_grads fills (fp32, *n_floats long) — what RCCL all-reduces over xGMI. */
int xq_dqn_grad_buffer(xq_dqn* d, float** grads_dev, size_t* n_floats);
/* on = 1: xq_dqn_apply_grads may sum the layer-0 gradient's partial sums itself (same order, bit-identical update, one
 * kernel fewer); the layer-0 segment of the gradient buffer is then NOT filled by xq_dqn_td_grads.  Leave 0 (default) when
 * anything reads the buffer between the two calls, e.g. a multi-GPU all-reduce. */
int xq_dqn_set_fused_apply(xq_dqn* d, int on);
/* Convenience: sample-free TD update straight from a replay ring (slots from the last xq_replay_sample). */
int xq_dqn_td_grads_replay(xq_dqn* d, xq_replay* r, int batch, int td_net, int mode);
/* Host-buffer TD step for tests: n transitions as 90-byte boards. Returns Q(s,a) and y per sample if non-NULL. */
int xq_dqn_td_update_host(xq_dqn* d, int n, const uint8_t* boards90, const uint8_t* next_boards90,
                          const int32_t* action_to, const float* reward, const uint8_t* done, int td_net, int mode,
                          double learning_rate, double grad_scale, float* q_sa_out, float* y_out);
/* Sum-of-squared TD error of the last xq_dqn_td_grads (synchronises). */
int xq_dqn_last_loss(xq_dqn* d, double* loss);
/* Q(s,a) and the TD target y of the first n samples of the last xq_dqn_td_grads* (tests / interop; synchronises).  Either pointer
 * may be NULL. */
int xq_dqn_last_td_values(xq_dqn* d, int n, float* q_sa_host, float* y_host);
/* Per-kernel HIP-event timing on the handle stream, for bench.py (name -> summed ms, launches, algorithmic flops/bytes).
 * enable: -1 leave, 0 off, 1 on, 2 on + clear, 3 on + clear but bracket only the TD step's dominant GEMM (gemm_qmax_rowmax /
 * gemm_qmax_screen) and env_selfplay_step, 4 = 3 with the GEMM bracketed on every 4th launch only (a pair of event records
 * drains the stream's queue: ~10 us per bracket). */
/* exact_launches: how many of `launches` were timed by the kernel's OWN start / stop events (single-kernel brackets, launched with
 * hipExtLaunchKernelGGL): their ms is the kernel's duration as a profiler reports it; the others are pairs of recorded events around the
 * launch(es), which read ~6.5 us more per bracket. */
typedef struct { char name[48]; float ms; int launches; double flops; double bytes; int exact_launches; int reserved; } xq_kernel_stat;
int xq_dqn_kernel_stats(xq_dqn* d, int enable, xq_kernel_stat* stats, int max_stats, int* n_stats);
/* Which launches enable = 3 / 4 bracket: a comma-separated list of bracket names as xq_dqn_kernel_stats reports them (launches of the
 * self-play loop's select chain on the trainer's collect stream carry the suffix "@select"; "rccl_allreduce_grads" = the gradient
 * all-reduce of a data-parallel step).  NULL / "" = the default: gemm_qmax_rowmax, gemm_qmax_screen, env_selfplay_step.  bench.py
 * walks the kernels of the step with it one at a time (roofline_chain), so that each is measured with only its own bracket in the loop. */
int xq_dqn_kernel_filter(xq_dqn* d, const char* names_csv);
/* Live timeline of the brackets gathered by the last xq_dqn_kernel_stats call (enable 1/2 sessions): start/end of every
 * bracketed launch in ms relative to the first one, across the handle's streams — what a kernel trace shows, without a
 * profiler attached.  Diagnostic. */
typedef struct { char name[48]; float start_ms; float end_ms; } xq_kernel_span;
int xq_dqn_kernel_timeline(xq_dqn* d, xq_kernel_span* spans, int max_spans, int* n_spans);

/* ------------------------------------------------------------------------------------------------------------
 * xq_comm — the one exchange step of the data-parallel loop (no upstream analogue: the reference is
 * single-GPU): games are sharded over ranks by contiguous game-id ranges, one process per GPU, and each update sums
 * the compact TD gradient buffer over all ranks with an RCCL all-reduce over xGMI; every replica then applies the same SGD
 * step (xq_trainer_learn_apply scales by 1/(minibatch*world)).  RCCL is loaded on first use.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct xq_comm xq_comm;
enum { XQ_COMM_ID_BYTES = 128 };
/* rank 0 draws the communicator id (ncclGetUniqueId) and ships the 128 bytes to the other ranks out of band. */
int xq_comm_unique_id(uint8_t* id128);
/* ncclCommInitRank on the calling process's current device (xq_set_device).  Collective: every rank calls it. */
int xq_comm_create(int rank, int world, const uint8_t* id128, xq_comm** out);
/* Same, with the id exchanged through a file every rank can see (rank 0 writes, the others poll up to timeout_s).  `path` must
 * not exist beforehand (rank 0 refuses a leftover, XQ_ERR_IO: it would hand the other ranks a dead id); rank 0 removes the file
 * again once every rank has joined, so one path serves run after run. */
int xq_comm_create_from_file(int rank, int world, const char* path, double timeout_s, xq_comm** out);
int xq_comm_destroy(xq_comm* c);
int xq_comm_info(const xq_comm* c, int* rank, int* world, uint64_t* collectives_issued, uint64_t* floats_reduced);
/* Sum of one host counter over all ranks (blocking; control decisions every rank must take alike, e.g. "enough episodes"). */
int xq_comm_sum_u64(xq_comm* c, uint64_t* inout_host);
/* In-place sum of n_floats fp32 over all ranks on hip_stream (NULL = the communicator's own stream). */
int xq_comm_allreduce(xq_comm* c, float* buf_dev, size_t n_floats, void* hip_stream);
/* Attach (or detach with NULL) a communicator: every xq_dqn_td_grads* then all-reduces the gradient buffer itself, on the stream of
 * its producer right behind it (no communicator stream, no extra events).  fp32 nets (fused launches, xq_dqn_set_td_tail): ONE
 * collective over the whole buffer on the handle's stream, behind the launch that reduces the step's partial sums into it.  bf16
 * nets and xq_dqn_set_td_tail(0): two buckets — [hidden + output-layer weights, biases] on the library's side stream, issued first,
 * the layer-0 segment (the largest and last) on the handle's stream behind the layer-0 kernel.  Either way the call returns with
 * the handle's stream ordered behind the exchange; xq_dqn_apply_grads therefore sees the global sum.  world = 1 is bit-identical
 * to no communicator. */
int xq_dqn_set_comm(xq_dqn* d, xq_comm* comm);
/* Where the event the trainer's select chain waits for is recorded in a data-parallel TD step (fp32 nets, fused launches).  0: behind
 * the max pass, as on one GPU (the select chain runs beside the gradient kernels; the all-reduce is exposed).  1: behind the gradient
 * kernels, in front of the all-reduce (the select chain runs beside the exchange and hides it; costs ~41 us of overlap with the
 * gradient kernels).  -1 (default): decided by MEASUREMENT — xq_dqn_set_comm with a communicator of more than one rank times 20
 * all-reduces of the gradient buffer (xq_dqn_calibrate_exchange) and takes 1 iff their mean exceeds 41 us; one rank: 0.  No effect
 * without a communicator.  Same results either way. */
int xq_dqn_set_exchange_overlap(xq_dqn* d, int mode);
/* COLLECTIVE (every rank of the attached communicator, same point of the program): times 4 + 20 all-reduces of a scratch buffer of the
 * gradient buffer's size on the handle's stream, averages the mean over the ranks and sets the -1 (auto) choice above to "late" iff
 * it exceeds threshold_us (< 0: the library's 41 us).  Works with one rank too (tests force both branches with the threshold).
 * xq_dqn_exchange_calibration reads back what the last calibration measured and chose (calibrated = 0: none yet). */
int xq_dqn_calibrate_exchange(xq_dqn* d, double threshold_us, double* allreduce_us, int* late);
int xq_dqn_exchange_calibration(const xq_dqn* d, int* calibrated, double* allreduce_us, double* threshold_us, int* late);
/* One all-reduce of the whole gradient buffer on the handle's stream, for callers that do not attach a communicator. */
int xq_allreduce_grads(xq_dqn* d, xq_comm* comm);

/* ------------------------------------------------------------------------------------------------------------
 * xq_trainer — the ChessAI::train() loop (chessai.cpp:85-170) for n_games boards at once, on device.
 *   collect : Q(s)[0..89] for every game -> xq_env_selfplay_step -> transitions into the replay ring
 *   learn   : sample minibatch -> xq_dqn_td_grads  [caller may all-reduce xq_dqn_grad_buffer] -> apply
 *   target sync every target_sync_interval learn steps (chessai.cpp:140 uses moveCount % 100).
 * With overlap_collect the same three calls are issued in the same order; collect is queued on a second stream and
 * learn_apply joins it, so env stepping hides behind the TD step (and behind the caller's gradient all-reduce).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct xq_trainer xq_trainer;
typedef struct {
    int n_games;
    int layer_sizes[XQ_MAX_LAYERS + 1];
    int n_sizes;
    double learning_rate, gamma, epsilon;   /* chessai.h:48-49, chessai.cpp:106 */
    int replay_capacity;                    /* 0 => on-policy: learn on the n_games transitions just collected (reference semantics) */
    int minibatch;                          /* transitions per learn step (<= replay capacity; on-policy: n_games) */
    int td_net;                             /* XQ_TD_* */
    int backprop_mode;                      /* XQ_BACKPROP_* */
    int target_sync_interval;               /* learn steps between updateTargetNetwork(); 0 = never */
    int mean_gradient;                      /* 1: grad_scale = 1/(minibatch*world), 0: sum (grad_scale = 1) */
    uint64_t seed;
    uint32_t first_game_id;
    int collects_per_update;                /* plies played in every game per learn step (0 or 1 = one; BASELINE configs[3] uses 4) */
    int overlap_collect;                    /* 1: collect runs on its own HIP stream beside learn_grads of the same iteration; both
                                             * read the same parameters, and the minibatch is drawn from the ring minus the slots the
                                             * collect is writing (they become eligible one iteration later).  Needs a replay ring. */
    /* build-defined modes of BASELINE configs[4] (td_net = XQ_TD_DOUBLE selects Double DQN): */
    int prioritized;                        /* 1: proportional prioritized replay (xq_replay_enable_per); the minibatch never contains
                                             * the slots this iteration's collects write: they are retired from the tree at learn_apply */
    double per_alpha, per_beta, per_eps;    /* 0 => 0.6, 0.4, 1e-3 */
    int precision;                          /* XQ_PRECISION_* of the Q-network's forward passes */
} xq_trainer_config;

int xq_trainer_create(const xq_trainer_config* cfg, void* hip_stream, xq_trainer** out);
int xq_trainer_destroy(xq_trainer* t);
int xq_trainer_env(xq_trainer* t, xq_env** env);
int xq_trainer_dqn(xq_trainer* t, xq_dqn** dqn);
int xq_trainer_replay(xq_trainer* t, xq_replay** replay);
/* xq_dqn_set_comm on the trainer's network: learn_grads then carries the bucketed all-reduce, learn_apply(world) the mean. */
int xq_trainer_set_comm(xq_trainer* t, xq_comm* comm);
/* n_plies uniform-random plies in every game (no Q-network, nothing written to the replay ring, not counted as env steps):
 * desynchronises the games so that a measurement or a training run starts from a spread of game phases rather than from
 * n_games copies of the opening.  Call between iterations. */
int xq_trainer_random_plies(xq_trainer* t, int n_plies);
/* Which rule gives the TD target from the next learn step on: XQ_TD_ONLINE_NET (ChessAI::train, chessai.cpp:126), XQ_TD_TARGET_NET
 * (DQN::train, dqn.cpp:166) or XQ_TD_DOUBLE.  Call between iterations. */
int xq_trainer_set_td_net(xq_trainer* t, int td_net);
int xq_trainer_collect(xq_trainer* t);                       /* one ply in every game */
int xq_trainer_learn_grads(xq_trainer* t);                   /* sample + gradients into the grad buffer */
int xq_trainer_learn_apply(xq_trainer* t, int world_size);   /* SGD apply (+ target sync bookkeeping) */
int xq_trainer_step(xq_trainer* t, int n_iterations);        /* (collect x collects_per_update) + learn_grads + learn_apply(world of the attached communicator, else 1), n times */
int xq_trainer_counters(xq_trainer* t, uint64_t* env_steps, uint64_t* updates, uint64_t* episodes);

#ifdef __cplusplus
}
#endif
#endif /* XQ_CAPI_H */
