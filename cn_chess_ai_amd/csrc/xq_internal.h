// xq_internal.h — handle layouts shared by the translation units of libxqhip (not part of the ABI).
#pragma once

#include "xq_common.h"

#include <vector>

namespace xq {

// ---- ordering of device data shared between handles that run on different HIP streams ---------------------------------------------
// Classes of ordering the library provides (include/xq_capi.h, "Stream ordering"); xq_debug_set_stream_ordering switches them off one
// by one so that a test can show each of them is needed.
enum : unsigned { ORD_RING_CONTENTS = 1u, ORD_RING_PRIORITIES = 2u, ORD_RING_DRAW = 4u, ORD_TRAINER_PARAMS = 8u, ORD_ALL = 15u };
inline unsigned& order_mask() { static unsigned m = ORD_ALL; return m; }

// Flags of an event that only ever orders one device stream behind another (hipStreamWaitEvent; never hipEventSynchronize / Query):
// no timing and no system-scope fence when it completes — the consumer is a kernel on the same device, which acquires at device scope
// like any kernel behind another.  Measured (tools/sync_probe.hip): a record costs the recording stream 6.9 instead of 9.4 us between
// 6-us kernels; same-box A/B of the step 0.1739 -> 0.1715 ms.  XQ_EVENT_SYSFENCE=1 restores the fence (A/B knob).
inline unsigned stream_event_flags() {
    static const unsigned f = [] { const char* e = getenv("XQ_EVENT_SYSFENCE"); return (unsigned)hipEventDisableTiming | ((e && e[0] == '1') ? 0u : (unsigned)hipEventDisableSystemFence); }();
    return f;
}

// One device resource and the streams that touched it last.  read(s) orders s behind the last write; write(s) orders s behind the
// last write and behind every read since.  The event is recorded LAZILY — on the producer's stream at the moment a consumer on another
// stream shows up (streams run in order, so a record made later still lies behind the producer's work) — because a record costs the
// recording stream ~6 us and most producers are never waited for.  Streams the library destroys are struck
// from every resource first (retire_stream), and every xq_*_destroy strikes the handle's stream — its own or the caller's — after
// synchronising it; a caller that destroys a lent stream while a handle still uses it gets its entries dropped at the next access
// (order_behind: a record on a dead stream fails, a dead stream has nothing left to wait for).
struct SharedResource {
    unsigned cls;
    bool has_writer = false;
    hipStream_t writer = nullptr;
    hipStream_t readers[4] = {nullptr, nullptr, nullptr, nullptr};
    int n_readers = 0;
    hipEvent_t ev = nullptr;
    explicit SharedResource(unsigned c);
    ~SharedResource();
    SharedResource(const SharedResource&) = delete;
    SharedResource& operator=(const SharedResource&) = delete;
    int order_behind(hipStream_t waiter, hipStream_t producer);
    int read(hipStream_t s);
    int write(hipStream_t s);
    void forget(hipStream_t s);        // s was synchronised and goes away (caller holds the registry's mutex: retire_stream, order_behind)
    void host_synchronised();          // the whole device was synchronised: nothing left to wait for
};
void retire_stream(hipStream_t s);     // after hipStreamSynchronize(s), before hipStreamDestroy(s)

}  // namespace xq

// replay ring in HBM: structure of arrays, states as packed boards (48 B) — never 1260 floats
struct ReplayDev {
    uint32_t* boards = nullptr;       // [capacity][12]
    uint32_t* next_boards = nullptr;  // [capacity][12]
    int32_t* action_to = nullptr;     // [capacity]   action.to (0..89), -1 = empty slot (contributes no gradient)
    float* reward = nullptr;          // [capacity]
    uint8_t* done = nullptr;          // [capacity]
    int capacity = 0;
    // prioritized replay (xq_replay_enable_per; nullptr = uniform): priority of every slot = leaves of the sum tree, and the
    // largest priority assigned so far as it stood at the last rebuild (what a new transition enters with), as float bits
    float* prio = nullptr;            // [capacity rounded up to 32]
    const unsigned* pmax_snap = nullptr;
};

struct xq_replay {
    ReplayDev dev;
    int size = 0;
    int write_pos = 0;
    uint64_t total = 0;
    uint64_t seed = 0;
    uint64_t sample_calls = 0;
    int32_t* slots_dev = nullptr;     // last sample()
    int slots_cap = 0;
    int last_batch = 0;
    bool implicit = false;            // last sample() was "virtual": consumers derive slot i = philox(i, call) % size themselves
    uint32_t implicit_call = 0;
    int implicit_size = 0;
    int implicit_start = 0;           // windowed sample: slot = (start + philox % size) % capacity
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // Device data of the ring that other handles read or write, possibly on streams of their own — each a SharedResource (above):
    //   contents : boards / next_boards / action_to / reward / done   written by env steps (xq_env_selfplay_step) and push_host,
    //              read by TD steps (xq_dqn_td_grads_replay)
    //   prio     : live priority table + running maximum, and the tree / snapshot a rebuild makes of them
    //              written by env steps (new transitions enter with the snapshot maximum), by TD steps (TD-error priorities) and by
    //              a rebuild; every one of them also reads what the one before wrote, so they are all writers.  (Draws read the tree on
    //              the stream the rebuild ran on — the ring's — and need no entry.)
    //   draw     : slot list, importance weights, batch maximum          written by a draw, read by the TD step that consumes it
    // Every access goes through read() / write(), which order the accessing stream behind the accesses it depends on when those ran
    // on another stream; on one stream (the trainer's layout) they cost nothing.  `caller_orders`: the owner orders the accesses
    // itself (xq_trainer: its collects write ring slots that the TD step running beside them never samples).
    xq::SharedResource contents{xq::ORD_RING_CONTENTS}, prio_res{xq::ORD_RING_PRIORITIES}, draw{xq::ORD_RING_DRAW};
    bool caller_orders = false;
    // prioritized replay (build-defined, BASELINE configs[4]): radix-32 sum tree, level 0 = dev.prio
    struct Per {
        bool enabled = false;
        float alpha = 0.6f, beta = 0.4f, eps = 1e-3f;
        int nlv = 0, n[8] = {0}, p[8] = {0};
        float* leaves = nullptr;          // level 0 as of the last rebuild (copy of dev.prio; the sampler reads this, never the live table)
        float* upper = nullptr;           // levels 1.. concatenated (padded), root last
        size_t off[8] = {0};              // offset of level lv inside `upper` (lv >= 1)
        unsigned* scalars = nullptr;      // [0] max priority, live (atomicMax of float bits)  [1] its snapshot at the last rebuild
                                          // [2] max raw importance weight of the last sample (float bits)  [3] eligible slots (p > 0)
        unsigned* wave_counts = nullptr;  // per-wave counts of non-zero leaves of the last rebuild (summed into scalars[3] by per_upper_kernel)
        bool wmax_clean = false;          // scalars[2] is zero (set by a rebuild, consumed by the next draw)
        bool draw_unconsumed = false;     // a prioritized draw whose TD step has not been queued yet: a rebuild must leave scalars[2] alone
        float* is_w = nullptr;            // [slots_cap] raw importance weights of the last prioritized sample
        bool last_prioritized = false;
    } per;
};

struct xq_env {
    int n = 0;
    uint64_t seed = 0;
    uint32_t first_id = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t* boards = nullptr;        // [n][12]
    uint4* meta = nullptr;             // [n] {moveCount|player<<16, red|black<<16, plies, episodes}
    uint4* stats = nullptr;            // [n] {red wins, black wins, captures, explored}
    xq_step_result* results = nullptr; // [n]
    uint16_t* codes = nullptr;         // [n][128] scratch for legal_moves
    int32_t* counts = nullptr;         // [n]
    int32_t* actions = nullptr;        // [n] scratch for step()
    float* q90 = nullptr;              // [n][96] scratch for selfplay_step_host
    uint8_t* validmat = nullptr;       // [8100]
    xq_episode_record* ep_ring = nullptr;
    int ep_cap = 0;
    unsigned long long* ep_head = nullptr;   // device counter of finished episodes
    uint64_t ep_drained = 0;
};

namespace xq {

// HIP-event profiler: brackets individual launches on the handle's stream (bench.py roofline leg).  Disabled = free.
struct Profiler {
    struct Cat { char name[48]; double flops = 0, bytes = 0; int launches = 0; int exact = 0; float ms = 0; };   // exact: launches timed by their own events
    struct Rec { int cat; hipEvent_t a, b; bool attached; };   // attached: a / b are the start / stop events of the kernel's own launch
    struct Span { char name[48]; float start_ms, end_ms; };      // relative to the first bracket of the batch (live timeline)
    std::vector<Span> spans;
    bool enabled = false;
    bool roofline_only = false;   // bracket only the kernels named in `only` (keeps the timed region undisturbed)
    int sample_period = 1;        // roofline_only: bracket every sample_period-th launch of a named kernel (an event record drains
                                  // the recording queue: ~10 us per bracket on a stream of 10-50 us kernels, tools/sync_probe.hip)
    struct Only { char name[48]; int phase; int period_add; };
    std::vector<Only> only;       // xq_dqn_kernel_filter; default = the three kernels bench.py priced in rounds 1-3
    void set_only(const char* csv) {
        only.clear();
        for (const char* p = csv; p && *p;) {
            const char* q = strchr(p, ',');
            const size_t len = q ? (size_t)(q - p) : strlen(p);
            Only o; memset(&o, 0, sizeof o);
            memcpy(o.name, p, len < sizeof o.name - 1 ? len : sizeof o.name - 1);
            // the env kernel on a period of its own, sample_period + 1: with 4 plies per update a shared period of 4 would always
            // pick the same ply
            o.period_add = strncmp(o.name, "env_selfplay_step", 17) == 0 ? 1 : 0;
            if (len) only.push_back(o);
            p = q ? q + 1 : nullptr;
        }
    }
    std::vector<Cat> cats;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    int cat_id(const char* name) {
        for (size_t i = 0; i < cats.size(); ++i) if (!strcmp(cats[i].name, name)) return (int)i;
        Cat c; memset(c.name, 0, sizeof c.name); strncpy(c.name, name, sizeof c.name - 1);
        cats.push_back(c);
        return (int)cats.size() - 1;
    }
    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    // attach: the caller hands recs[h].a / .b to hipExtLaunchKernelGGL as the launch's start / stop events — the elapsed time is then the
    // kernel's own (what rocprofv3 reports), with no marker packets on the stream; otherwise the pair is recorded around the scope
    // (one or more launches; reads ~6.5 us more than a single kernel, tools/sync_probe.hip)
    int begin(const char* name, hipStream_t s, bool attach = false) {
        if (!enabled) return -1;
        if (roofline_only) {
            if (only.empty()) set_only("gemm_qmax_rowmax,gemm_qmax_screen,env_selfplay_step");
            Only* o = nullptr;
            for (auto& x : only) if (!strcmp(x.name, name)) { o = &x; break; }
            if (!o) return -1;
            const int period = sample_period > 1 ? sample_period + o->period_add : 1;
            if (period > 1 && (o->phase++ % period) != 0) return -1;
        }
        Rec r; r.cat = cat_id(name); r.a = get_event(); r.b = get_event(); r.attached = attach;
        if (!attach) (void)hipEventRecord(r.a, s);
        recs.push_back(r);
        return (int)recs.size() - 1;
    }
    void end(int h, hipStream_t s, double flops, double bytes) {
        if (h < 0) return;
        if (!recs[h].attached) (void)hipEventRecord(recs[h].b, s);
        Cat& c = cats[recs[h].cat];
        c.flops += flops; c.bytes += bytes; c.launches += 1; c.exact += recs[h].attached ? 1 : 0;
    }
    void collect() {
        if (!recs.empty()) spans.clear();
        for (auto& r : recs) {
            float ms = 0;
            (void)hipEventSynchronize(r.b);
            (void)hipEventElapsedTime(&ms, r.a, r.b);
            cats[r.cat].ms += ms;
            if (!roofline_only && spans.size() < 8192) {
                Span sp; memcpy(sp.name, cats[r.cat].name, sizeof sp.name); sp.start_ms = 0; sp.end_ms = 0;
                (void)hipEventElapsedTime(&sp.start_ms, recs[0].a, r.a);
                (void)hipEventElapsedTime(&sp.end_ms, recs[0].a, r.b);
                spans.push_back(sp);
            }
        }
        for (auto& r : recs) { pool.push_back(r.a); pool.push_back(r.b); }
        recs.clear();
    }
    void reset() { collect(); cats.clear(); }
    ~Profiler() { collect(); for (auto e : pool) (void)hipEventDestroy(e); }
};

// communicator internals (xq_comm.hip)
int comm_allreduce_on(xq_comm* c, float* buf, size_t n_floats, hipStream_t stream);   // in-place sum over all ranks
int comm_world(const xq_comm* c);

// dqn-side entry points used by the trainer (defined in xq_dqn.hip)
Profiler* dqn_profiler(xq_dqn* d);
xq_comm* dqn_comm(const xq_dqn* d);         // communicator attached with xq_dqn_set_comm, or nullptr
int dqn_fused_apply(const xq_dqn* d);      // current xq_dqn_set_fused_apply setting
// Q[0..95] of the select chain still as the k-slabs of the head ([nslabs][n][96], nslabs even): the env kernel adds them, the bias and
// the tanh itself, for the boards that exploit (dqn_q90_boards fills it when the head rode on the last hidden product)
struct QSource { const float* slabs; long long slab_stride; int nslabs; const float* bias; };
// qs != nullptr: when the select head rides on the last hidden product, its k-slabs are handed over as they are (qs->slabs != nullptr,
// *q90_dev = nullptr) for the env kernel to finish; otherwise qs->slabs = nullptr and *q90_dev holds the finished values as always
int dqn_q90_boards(xq_dqn* d, const uint32_t* boards_dev, int n, float** q90_dev, int* q_stride, hipStream_t on = nullptr, QSource* qs = nullptr);
hipStream_t dqn_stream(xq_dqn* d);
hipEvent_t dqn_qmax_event(xq_dqn* d);      // recorded behind the column-max GEMM of the last xq_dqn_td_grads*
uint64_t dqn_params_version(const xq_dqn* d);   // counts the operations that rewrote parameters (either net, fp32 or the bf16 shadow)

// "virtual" replay sample: same slots as xq_replay_sample would write (Philox ctr = {i, 0, call, 1}, key = seed, % size),
// but no kernel and no slot buffer — the consumer kernels recompute them.  Used by the trainer's hot loop.
// (start, count) restricts the draw to `count` ring slots from `start` (count < 0: the whole filled part).
int replay_sample_implicit(xq_replay* r, int batch, int start = 0, int count = -1);

// cross-stream ordering of the ring's users (SharedResource; no-ops on one stream or when the owner orders: xq_replay::caller_orders)
int replay_consumer_begin(xq_replay* r, hipStream_t consumer, bool listed, bool prioritized);   // a TD step that reads the ring
int replay_writer_begin(xq_replay* r, hipStream_t writer);                                      // an env step that writes it
// prioritized replay internals used by the trainer (xq_replay.hip)
int replay_per_rebuild(xq_replay* r, int retire_start, int retire_count, hipStream_t on);
int replay_per_sample(xq_replay* r, int batch, hipStream_t on);

// env-side launchers used by the trainer
int env_selfplay_launch(xq_env* env, const float* q90_dev, int q_stride, uint32_t eps_u32, xq_step_result* results_dev,
                        xq_replay* replay, hipStream_t on = nullptr, const QSource* qs = nullptr, hipEvent_t ev_start = nullptr,
                        hipEvent_t ev_stop = nullptr);      // ev_*: the kernel's own start / stop events (profiler, kernel-exact timing)
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
}  // namespace xq
