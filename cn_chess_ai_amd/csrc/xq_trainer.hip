// xq_trainer.hip — the ChessAI::train() loop (reference chessai.cpp:85-170) for n_games boards at once, on device.
//
// Per iteration (one ply in every game + one minibatch update):
//   collect : Q(s)[0..89] of every game (selectAction reads q[action.to] only, dqn.cpp:47) -> fused self-play
//             kernel (legal moves in LDS, epsilon-greedy select, movePiece, evaluateBoard, done, auto-reset) ->
//             transition written straight into the replay ring;
//   learn   : sample -> TD gradients (xq_dqn_td_grads) -> [caller all-reduces the gradient buffer over RCCL] -> SGD;
//   target  : updateTargetNetwork() every target_sync_interval updates (chessai.cpp:140 uses moveCount % 100).
// No host round trip inside an iteration; everything is queued on one HIP stream — or, with cfg.overlap_collect, collect
// goes to a second stream: collect(t) and learn_grads(t) both read theta_t, learn_grads(t) samples the ring minus the
// slots collect(t) is writing, learn_apply(t) joins both before theta_{t+1} is written.  Preferred call order there:
// learn_grads, collect, learn_apply — collect then starts behind the column-max GEMM (which wants the chip to itself) and
// runs beside the gradient chain, which is a string of small latency-bound kernels.  collect-first is also accepted.
#include "xq_internal.h"

struct xq_trainer {
    xq_trainer_config cfg;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    xq_env* env = nullptr;
    xq_dqn* dqn = nullptr;
    xq_replay* replay = nullptr;
    uint64_t env_steps = 0, updates = 0;
    uint32_t eps_u32 = 0;
    // overlap_collect
    hipStream_t cstream = nullptr;          // collect stream
    hipEvent_t ev_params = nullptr;         // main: parameters of the next iteration are final (after learn_apply)
    hipEvent_t ev_collect = nullptr;        // cstream: the collects of this iteration are done
    bool collect_event_stale = false;       // plies queued on cstream since ev_collect was last recorded (recorded when someone waits:
                                            // a record per ply costs the collect stream ~6 us each, wait_collects())
    int inflight = 0;                       // ring slots written by collects since the last learn_apply
    bool grads_queued = false;              // learn_grads already queued in this iteration: collect starts behind its big GEMM
    int prepaid_collects = 0;               // collects learn_grads had to run itself (empty ring), owed to the next collect() calls
    int excluded = 0;                       // ring slots the queued learn_grads left out of its minibatch (from write_pos - inflight)
    hipEvent_t ev_grads = nullptr;          // main: the queued learn_grads has finished reading the ring
    bool per_ready = false;                 // prioritized replay: the sum tree holds the ring as of the last learn_apply
    // an event record costs the recording stream ~6 us (rocprofv3 trace): ev_params / ev_grads are only recorded when a collect
    // actually has to wait for them (collect-first or mixed call orders), not on every iteration of the learn_grads -> collect loop
    bool params_event_stale = true;         // ev_params has not been recorded since the last parameter update
    bool grads_event_stale = true;          // ev_grads has not been recorded since the queued learn_grads
    bool early_collect = false;             // collects_per_update > 1: collects start behind the previous parameter update (see collect_impl)
    uint64_t params_version = 0;            // dqn_params_version() when ev_params was last recorded: anything that rewrote parameters on the
                                            // handle's stream since (set_params on a bf16 net, an apply_grads of the caller's own) lies
                                            // BEHIND that record, and the select chain must not start in front of it
};

using namespace xq;

extern "C" {

static int trainer_init(xq_trainer* t, const xq_trainer_config* cfg, void* hip_stream);

int xq_trainer_create(const xq_trainer_config* cfg, void* hip_stream, xq_trainer** out) {
    if (!cfg || !out) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_trainer_create: null pointer");
    if (cfg->n_games <= 0 || cfg->n_sizes < 3 || cfg->n_sizes > XQ_MAX_LAYERS + 1)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_trainer_create: bad n_games / layer sizes");
    if (cfg->replay_capacity != 0 && cfg->replay_capacity < cfg->n_games)
        return fail(XQ_ERR_INVALID_ARGUMENT, "replay_capacity must be 0 (on-policy) or >= n_games");
    if (cfg->replay_capacity != 0 && cfg->minibatch <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "minibatch must be > 0");
    if (cfg->overlap_collect && cfg->replay_capacity == 0)
        return fail(XQ_ERR_INVALID_ARGUMENT, "overlap_collect needs a replay ring (on-policy learns on the ply just played)");
    int ndev = 0;
    XQ_TRY(xq_device_count(&ndev));
    if (ndev == 0) return fail(XQ_ERR_NO_DEVICE, "no HIP device: libxqhip has no CPU fallback");
    xq_trainer* t = new xq_trainer();
    const int rc = trainer_init(t, cfg, hip_stream);
    if (rc != XQ_OK) { xq_trainer_destroy(t); return rc; }  // a failed allocation must not leak the handles before it
    *out = t;
    return XQ_OK;
}

static int trainer_init(xq_trainer* t, const xq_trainer_config* cfg, void* hip_stream) {
    t->cfg = *cfg;
    if (hip_stream) t->stream = (hipStream_t)hip_stream;
    else { XQ_HIP(hipStreamCreate(&t->stream)); t->own_stream = true; }
    XQ_TRY(xq_env_create(cfg->n_games, cfg->seed, cfg->first_game_id, t->stream, &t->env));
    XQ_TRY(xq_dqn_create(cfg->layer_sizes, cfg->n_sizes, cfg->learning_rate, cfg->gamma, cfg->seed ^ 0x9E3779B97F4A7C15ull,
                         t->stream, &t->dqn));
    const int cap = cfg->replay_capacity > 0 ? cfg->replay_capacity : cfg->n_games;
    XQ_TRY(xq_replay_create(cap, cfg->seed + 0x1234567ull + cfg->first_game_id, t->stream, &t->replay));
    t->replay->caller_orders = true;         // env, ring and Q-net share t->stream; the collect stream is ordered below, by hand
    if (cfg->precision != XQ_PRECISION_F32) XQ_TRY(xq_dqn_set_precision(t->dqn, cfg->precision));
    if (cfg->prioritized) {
        if (cfg->replay_capacity == 0) return fail(XQ_ERR_INVALID_ARGUMENT, "prioritized replay needs a replay ring (replay_capacity > 0)");
        XQ_TRY(xq_replay_enable_per(t->replay, cfg->per_alpha > 0 ? cfg->per_alpha : 0.6, cfg->per_beta > 0 ? cfg->per_beta : 0.4,
                                    cfg->per_eps > 0 ? cfg->per_eps : 1e-3));
    }
    const double e = cfg->epsilon < 0 ? 0 : cfg->epsilon;
    const double v = e * 4294967296.0;
    t->eps_u32 = v >= 4294967295.0 ? 4294967295u : (uint32_t)v;
    if (cfg->overlap_collect) {
        int lo = 0, hi = 0;
        XQ_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));        // lo = least urgent: collect has slack until learn_apply,
        // one ply per update: the TD step is the critical path, the select chain has slack until learn_apply => least urgent.
        // Several plies per update: the plies ARE the critical path (each waits for the move of the one before, bench --config 4:
        // 4 x 155 us alone, 375 + 314 us for the two that share the chip with the TD step) => most urgent: 1.096 -> 1.054 ms per step
        // (one ply per update with the select chain most urgent: no difference on the headline bench, -0.5 % on bench --config 5)
        const int prio = cfg->collects_per_update > 1 ? hi : lo;
        XQ_HIP(hipStreamCreateWithPriority(&t->cstream, hipStreamNonBlocking, prio));
        XQ_HIP(hipEventCreateWithFlags(&t->ev_params, stream_event_flags()));
        XQ_HIP(hipEventCreateWithFlags(&t->ev_collect, stream_event_flags()));
        XQ_HIP(hipEventCreateWithFlags(&t->ev_grads, stream_event_flags()));
        XQ_HIP(hipEventRecord(t->ev_params, t->stream));      // orders the first collect after the handles' initialisation
        t->params_event_stale = false;
        t->params_version = dqn_params_version(t->dqn);
        t->early_collect = cfg->collects_per_update > 1;
    }
    return XQ_OK;
}

int xq_trainer_destroy(xq_trainer* t) {
    if (!t) return XQ_OK;
    if (t->cstream) hipStreamSynchronize(t->cstream);
    hipStreamSynchronize(t->stream);
    if (t->cstream) { retire_stream(t->cstream); hipStreamDestroy(t->cstream); }
    if (t->ev_params) hipEventDestroy(t->ev_params);
    if (t->ev_collect) hipEventDestroy(t->ev_collect);
    if (t->ev_grads) hipEventDestroy(t->ev_grads);
    retire_stream(t->stream);        // synchronised above; unconditional: a caller-owned stream may be destroyed right after this call
    xq_env_destroy(t->env);
    xq_dqn_destroy(t->dqn);
    xq_replay_destroy(t->replay);
    if (t->own_stream) hipStreamDestroy(t->stream);
    delete t;
    return XQ_OK;
}

int xq_trainer_env(xq_trainer* t, xq_env** e) { if (!t || !e) return fail(XQ_ERR_INVALID_ARGUMENT, "null"); *e = t->env; return XQ_OK; }
int xq_trainer_dqn(xq_trainer* t, xq_dqn** d) { if (!t || !d) return fail(XQ_ERR_INVALID_ARGUMENT, "null"); *d = t->dqn; return XQ_OK; }
int xq_trainer_replay(xq_trainer* t, xq_replay** r) { if (!t || !r) return fail(XQ_ERR_INVALID_ARGUMENT, "null"); *r = t->replay; return XQ_OK; }

// the handle's stream waits for every ply queued on the collect stream so far
static int wait_collects(xq_trainer* t) {
    if (t->collect_event_stale) { XQ_HIP(hipEventRecord(t->ev_collect, t->cstream)); t->collect_event_stale = false; }
    XQ_HIP(hipStreamWaitEvent(t->stream, t->ev_collect, 0));
    return XQ_OK;
}

static int collect_impl(xq_trainer* t) {
    float* q90 = nullptr;
    int stride = 0;
    hipStream_t on = nullptr;
    if (t->cstream) {
        on = t->cstream;
        // has anything rewritten parameters on the handle's stream since ev_params was recorded?  (ORD_TRAINER_PARAMS off: the round-3
        // behaviour, for the test that shows the check is needed)
        const bool params_moved = (order_mask() & ORD_TRAINER_PARAMS) && dqn_params_version(t->dqn) != t->params_version;
        if (t->grads_queued && !t->early_collect) {
            // queued behind the column-max GEMM (which wants the chip to itself); that event lies behind every parameter update
            if (t->inflight == 0) XQ_HIP(hipStreamWaitEvent(on, dqn_qmax_event(t->dqn), 0));
        } else if (t->grads_queued) {
            // several plies per update: the collect chain is the long pole of the iteration (each ply waits for the move of the one
            // before) — it starts behind the previous parameter update, recorded there (learn_apply), and runs beside the whole TD step.
            // If parameters were rewritten behind that record, the event of the queued TD step (behind everything on the handle's
            // stream) takes its place.
            if (t->inflight == 0 && !t->params_event_stale && !params_moved) XQ_HIP(hipStreamWaitEvent(on, t->ev_params, 0));
            else if (t->inflight == 0) XQ_HIP(hipStreamWaitEvent(on, dqn_qmax_event(t->dqn), 0));     // (no record yet: first iteration)
        } else if (t->inflight == 0) {
            if (t->params_event_stale || params_moved) {
                XQ_HIP(hipEventRecord(t->ev_params, t->stream));
                t->params_event_stale = false;
                t->params_version = dqn_params_version(t->dqn);
            }
            XQ_HIP(hipStreamWaitEvent(on, t->ev_params, 0));
        }
        // more plies than the queued learn_grads excluded from its minibatch (caller mixed the orders): these slots may be
        // among the ones it samples, so they can only be overwritten once it has read them
        if (t->grads_queued && t->inflight + t->env->n > t->excluded) {
            if (t->grads_event_stale) { XQ_HIP(hipEventRecord(t->ev_grads, t->stream)); t->grads_event_stale = false; }
            XQ_HIP(hipStreamWaitEvent(on, t->ev_grads, 0));
        }
    }
    hipStream_t s = on ? on : t->stream;
    QSource qs;
    XQ_TRY(dqn_q90_boards(t->dqn, t->env->boards, t->env->n, &q90, &stride, on, &qs));
    if (t->cfg.replay_capacity == 0) {           // on-policy: the ring is exactly one batch, refilled every ply
        t->replay->write_pos = 0;
        t->replay->size = 0;
    }
    Profiler* p = dqn_profiler(t->dqn);
    const int h = p->begin("env_selfplay_step", s, true);       // one kernel: timed by its own start / stop events
    XQ_TRY(env_selfplay_launch(t->env, q90, stride, t->eps_u32, nullptr, t->replay, on, &qs, h >= 0 ? p->recs[h].a : nullptr,
                               h >= 0 ? p->recs[h].b : nullptr));
    // algorithmic HBM bytes per game and ply: board+meta in/out (2*(48+16)), Q row 360, transition 48+48+4+4+1
    p->end(h, s, 0.0, (double)t->env->n * (2.0 * (48 + 16) + 360 + 105));
    t->env_steps += (uint64_t)t->env->n;
    if (on) {
        t->inflight += t->env->n;
        t->collect_event_stale = true;
    }
    return XQ_OK;
}

int xq_trainer_collect(xq_trainer* t) {
    if (!t) return fail(XQ_ERR_INVALID_ARGUMENT, "null trainer");
    if (t->prepaid_collects > 0) { t->prepaid_collects--; return XQ_OK; }     // already played by learn_grads (empty ring)
    return collect_impl(t);
}

int xq_trainer_set_comm(xq_trainer* t, xq_comm* comm) {
    if (!t) return fail(XQ_ERR_INVALID_ARGUMENT, "null trainer");
    if (t->grads_queued) return fail(XQ_ERR_RUNTIME, "xq_trainer_set_comm: call between iterations (after learn_apply)");
    return xq_dqn_set_comm(t->dqn, comm);
}

int xq_trainer_set_td_net(xq_trainer* t, int td_net) {
    if (!t || (td_net != XQ_TD_ONLINE_NET && td_net != XQ_TD_TARGET_NET && td_net != XQ_TD_DOUBLE)) return fail(XQ_ERR_INVALID_ARGUMENT, "bad td_net");
    if (t->grads_queued) return fail(XQ_ERR_RUNTIME, "xq_trainer_set_td_net: call between iterations (after learn_apply)");
    t->cfg.td_net = td_net;
    return XQ_OK;
}

int xq_trainer_random_plies(xq_trainer* t, int n_plies) {
    if (!t || n_plies < 0) return fail(XQ_ERR_INVALID_ARGUMENT, "bad argument");
    if (t->inflight > 0 || t->grads_queued) return fail(XQ_ERR_RUNTIME, "xq_trainer_random_plies: call between iterations (after learn_apply)");
    for (int i = 0; i < n_plies; ++i) XQ_TRY(env_selfplay_launch(t->env, nullptr, 0, 0u, nullptr, nullptr, t->stream));
    t->params_event_stale = true;                // the next collect (own stream) starts behind these plies
    return XQ_OK;
}

// Prioritized replay: the minibatch comes from the sum tree as of the last learn_apply — it holds every transition collected
// before this iteration except the slots this iteration's collects overwrite (retired there), in both call orders and with or
// without overlap_collect.  An empty tree (first iteration) is filled by playing this iteration's plies first.
static int learn_grads_prioritized(xq_trainer* t) {
    const int plies = t->cfg.collects_per_update > 1 ? t->cfg.collects_per_update : 1;
    if (!t->per_ready) {
        if (t->inflight == 0 && t->replay->size == 0) {
            for (int c = 0; c < plies; ++c) XQ_TRY(collect_impl(t));
            t->prepaid_collects = plies;
        }
        if (t->cstream) { XQ_TRY(wait_collects(t)); t->inflight = 0; }
        if (t->replay->size == 0) return fail(XQ_ERR_RUNTIME, "replay is empty");
        XQ_TRY(replay_per_rebuild(t->replay, 0, 0, t->stream));
        t->per_ready = true;
        t->excluded = 0;
    } else {
        t->excluded = plies * t->cfg.n_games;       // retired at the last learn_apply: the collects may run beside this step
    }
    XQ_TRY(replay_per_sample(t->replay, t->cfg.minibatch, t->stream));
    XQ_TRY(xq_dqn_td_grads_replay(t->dqn, t->replay, t->cfg.minibatch, t->cfg.td_net, t->cfg.backprop_mode));
    t->grads_event_stale = true;
    t->grads_queued = true;
    return XQ_OK;
}

int xq_trainer_learn_grads(xq_trainer* t) {
    if (!t) return fail(XQ_ERR_INVALID_ARGUMENT, "null trainer");
    if (t->cfg.prioritized) return learn_grads_prioritized(t);
    int batch = 0;
    if (t->cfg.replay_capacity > 0) {
        batch = t->cfg.minibatch;
        int start = 0, count = -1;
        if (t->cstream) {
            // the ring minus the window the collects of this iteration write: [write_pos - inflight, write_pos) when they are
            // already queued, else the next `plies * n_games` slots from write_pos (they will be queued behind the GEMM)
            const xq_replay* r = t->replay;
            const int cap = r->dev.capacity;
            const int plies = t->cfg.collects_per_update > 1 ? t->cfg.collects_per_update : 1;
            const int m = t->inflight > 0 ? t->inflight : plies * t->cfg.n_games;
            const int size_after = t->inflight > 0 ? r->size : std::min(cap, r->size + m);
            const int wpos_after = t->inflight > 0 ? r->write_pos : (r->write_pos + m) % cap;
            if (size_after < cap) { start = 0; count = size_after - m; }
            else { start = wpos_after; count = cap - m; }
            if (count <= 0 || m > cap) {
                // nothing older than this iteration's plies (first iteration): play them now, wait, learn on those
                if (t->inflight == 0) {
                    for (int c = 0; c < plies; ++c) XQ_TRY(collect_impl(t));
                    t->prepaid_collects = plies;
                }
                XQ_TRY(wait_collects(t));
                t->inflight = 0;
                start = 0; count = -1;
                t->excluded = 0;
            } else {
                t->excluded = m;
            }
        }
        XQ_TRY(replay_sample_implicit(t->replay, batch, start, count));     // no sampling kernel: the consumers recompute the slots
    }
    XQ_TRY(xq_dqn_td_grads_replay(t->dqn, t->replay, batch, t->cfg.td_net, t->cfg.backprop_mode));
    t->grads_event_stale = true;
    t->grads_queued = true;
    return XQ_OK;
}

int xq_trainer_learn_apply(xq_trainer* t, int world_size) {
    if (!t || world_size < 1) return fail(XQ_ERR_INVALID_ARGUMENT, "bad argument");
    const int batch = t->cfg.replay_capacity > 0 ? t->cfg.minibatch : t->cfg.n_games;
    const double scale = t->cfg.mean_gradient ? 1.0 / ((double)batch * world_size) : 1.0;
    if (t->cstream && t->inflight > 0) XQ_TRY(wait_collects(t));   // the select chain reads theta_t
    XQ_TRY(xq_dqn_apply_grads(t->dqn, t->cfg.learning_rate, scale));
    t->updates += 1;
    if (t->cfg.target_sync_interval > 0 && t->updates % (uint64_t)t->cfg.target_sync_interval == 0)
        XQ_TRY(xq_dqn_update_target(t->dqn));
    if (t->cfg.prioritized) {
        // the tree of the next iteration: this iteration's TD-error priorities and new transitions in, the slots the next
        // collects will overwrite out (only once the ring is full — before that they are fresh slots with priority 0)
        const int plies = t->cfg.collects_per_update > 1 ? t->cfg.collects_per_update : 1;
        const int m = plies * t->cfg.n_games;
        const xq_replay* r = t->replay;
        const bool full = r->size + m > r->dev.capacity;
        XQ_TRY(replay_per_rebuild(t->replay, r->write_pos, full ? std::min(m, r->dev.capacity) : 0, t->stream));
        t->per_ready = true;
    }
    t->params_event_stale = true;
    if (t->cstream && t->early_collect) {       // the next iteration's collects start here, whatever the caller queues first
        XQ_HIP(hipEventRecord(t->ev_params, t->stream));
        t->params_event_stale = false;
        t->params_version = dqn_params_version(t->dqn);
    }
    t->inflight = 0;
    t->grads_queued = false;
    return XQ_OK;
}

int xq_trainer_step(xq_trainer* t, int n_iterations) {
    if (!t || n_iterations < 0) return fail(XQ_ERR_INVALID_ARGUMENT, "bad argument");
    const int plies = t->cfg.collects_per_update > 1 ? t->cfg.collects_per_update : 1;
    if (plies > 1 && t->cfg.replay_capacity == 0)
        return fail(XQ_ERR_INVALID_ARGUMENT, "collects_per_update > 1 needs a replay ring (on-policy keeps one ply)");
    const int world = dqn_comm(t->dqn) ? comm_world(dqn_comm(t->dqn)) : 1;     // data-parallel when a communicator is attached
    const int fused_before = dqn_fused_apply(t->dqn);   // restored on every exit path: the caller's choice survives step()
    XQ_TRY(xq_dqn_set_fused_apply(t->dqn, 1));      // single-GPU loop: nothing reads the gradient buffer between grads and apply
    int rc = XQ_OK;
    for (int i = 0; i < n_iterations && rc == XQ_OK; ++i) {
        if (t->cstream) {
            rc = xq_trainer_learn_grads(t);
            for (int c = 0; c < plies && rc == XQ_OK; ++c) rc = xq_trainer_collect(t);
        } else {
            for (int c = 0; c < plies && rc == XQ_OK; ++c) rc = xq_trainer_collect(t);
            if (rc == XQ_OK) rc = xq_trainer_learn_grads(t);
        }
        if (rc == XQ_OK) rc = xq_trainer_learn_apply(t, world);
    }
    const int rc_off = xq_dqn_set_fused_apply(t->dqn, fused_before);     // also after a failed iteration
    return rc != XQ_OK ? rc : rc_off;
}

int xq_trainer_counters(xq_trainer* t, uint64_t* env_steps, uint64_t* updates, uint64_t* episodes) {
    if (!t) return fail(XQ_ERR_INVALID_ARGUMENT, "null trainer");
    if (env_steps) *env_steps = t->env_steps;
    if (updates) *updates = t->updates;
    if (episodes) {
        unsigned long long head = 0;
        if (t->cstream) XQ_HIP(hipStreamSynchronize(t->cstream));
        XQ_HIP(hipStreamSynchronize(t->stream));
        XQ_HIP(hipMemcpy(&head, t->env->ep_head, sizeof head, hipMemcpyDeviceToHost));
        *episodes = head;
    }
    return XQ_OK;
}

}  // extern "C"
