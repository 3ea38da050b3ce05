// xq_replay.hip — ReplayBuffer: ring of (state, action.to, reward, nextState, done) in HBM.
//
// The reference has no replay buffer; the element is the argument list of the never-called
// DQN::train(state, action, reward, nextState, done) (reference dqn.cpp:157-172).  States are stored as packed boards
// (12 x u32 = 48 B) instead of 1260 doubles (10 080 B): 1 M transitions = 105 MB instead of 20 GB.
#include "xq_internal.h"

#include <algorithm>

namespace xq {

__global__ void replay_sample_kernel(int32_t* slots, int batch, uint32_t start, uint32_t size, uint32_t cap, uint32_t call,
                                     uint32_t seed_lo, uint32_t seed_hi) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= batch) return;
    const Philox4 r = philox4x32_10((uint32_t)i, 0u, call, 1u, seed_lo, seed_hi);
    uint32_t v = start + r.v[0] % size;
    if (v >= cap) v -= cap;
    slots[i] = (int32_t)v;
}

// ---- prioritized replay (build-defined) -------------------------------------------------------------------
// Radix-32 sum tree in fp32.  Every node is the SEQUENTIAL sum of its 32 children in index order — a fixed association, so
// the tree (and therefore every sampled slot) is reproducible bit for bit, on any device and from run to run; no float atomics.

// parent[i] = ((c[32i] + c[32i+1]) + ...) + c[32i+31]; also counts the non-zero children (eligible slots, level 1 only)
// `snapshot` (level 1 only): the children are also copied there — the sampler descends through this copy of the priority table,
// so collects and TD steps may rewrite the live table while a minibatch is being drawn
__global__ __launch_bounds__(256) void per_level_kernel(const float* __restrict__ child, float* __restrict__ parent, int n_parent,
                                                        unsigned* __restrict__ count_nonzero, float* __restrict__ snapshot) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int nz = 0;
    if (i < n_parent) {
        const float4* c = reinterpret_cast<const float4*>(child + (size_t)i * 32);
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = c[k];
        if (snapshot) {
#pragma unroll
            for (int k = 0; k < 8; ++k) reinterpret_cast<float4*>(snapshot + (size_t)i * 32)[k] = v[k];
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s += v[k].x; s += v[k].y; s += v[k].z; s += v[k].w;
            nz += (v[k].x > 0.f) + (v[k].y > 0.f) + (v[k].z > 0.f) + (v[k].w > 0.f);
        }
        parent[i] = s;
    }
    if (count_nonzero) {         // one count per wave, summed by per_upper_kernel: nothing to zero beforehand, no atomics
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nz += __shfl_xor(nz, off, 64);
        if ((threadIdx.x & 63) == 0) count_nonzero[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (unsigned)nz;
    }
}

struct PerTree {
    const float* leaves;      // level 0
    float* upper;             // levels 1..
    int nlv, n[8];
    long long off[8];
};
// levels 2.. in one block (they hold <= a few thousand nodes), then the snapshot of the running maximum priority, the number of eligible
// slots (sum of per_level_kernel's per-wave counts) and — unless a prioritized draw is still waiting for its TD step, which reads the
// slot — a clean batch-maximum slot for the next draw
__global__ __launch_bounds__(1024) void per_upper_kernel(PerTree T, unsigned* scalars, const unsigned* __restrict__ wave_counts, int n_counts,
                                                         int zero_wmax) {
    __shared__ unsigned csum[16];
    unsigned c = 0;
    for (int i = (int)threadIdx.x; i < n_counts; i += (int)blockDim.x) c += wave_counts[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) csum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += csum[w];
        if (wave_counts) scalars[3] = t;
        if (zero_wmax) scalars[2] = 0u;
    }
    for (int lv = 2; lv < T.nlv; ++lv) {
        const float* child = T.upper + T.off[lv - 1];
        float* parent = T.upper + T.off[lv];
        for (int i = (int)threadIdx.x; i < T.n[lv]; i += (int)blockDim.x) {
            float s = 0.f;
            for (int k = 0; k < 32; ++k) s += child[(size_t)i * 32 + k];
            parent[i] = s;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) scalars[1] = scalars[0];
}

__global__ void per_fill_kernel(float* prio, int start, int count, int cap, const unsigned* value_bits, float fixed) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= count) return;
    int s = start + i;
    if (s >= cap) s -= cap;
    prio[s] = value_bits ? __uint_as_float(*value_bits) : fixed;
}

// stratified draw: u_k = (k + r_k) * (total / B), r_k = (philox(k, 0, call, 2).v[0] >> 8) * 2^-24; then the descent of DESIGN.md §4
__global__ __launch_bounds__(256) void per_sample_kernel(PerTree T, int batch, uint32_t call, uint32_t seed_lo, uint32_t seed_hi,
                                                         float beta, unsigned* scalars, int32_t* __restrict__ slots,
                                                         float* __restrict__ is_w) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= batch) return;
    const float total = T.nlv > 1 ? T.upper[T.off[T.nlv - 1]] : T.leaves[0];
    const float seg = __fdiv_rn(total, (float)batch);
    const Philox4 rr = philox4x32_10((uint32_t)k, 0u, call, 2u, seed_lo, seed_hi);
    const float r = (float)(rr.v[0] >> 8) * (1.0f / 16777216.0f);
    float u = __fmul_rn(__fadd_rn((float)k, r), seg);
    int node = 0;
    for (int lv = T.nlv - 2; lv >= 0; --lv) {
        const float* v = (lv == 0 ? T.leaves : T.upper + T.off[lv]) + (size_t)node * 32;
        float c[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 x = reinterpret_cast<const float4*>(v)[q];
            c[4 * q] = x.x; c[4 * q + 1] = x.y; c[4 * q + 2] = x.z; c[4 * q + 3] = x.w;
        }
        int pick = -1, last = -1;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            if (pick < 0) {
                if (c[q] > 0.f) last = q;
                if (u < c[q]) pick = q;
                else u = __fsub_rn(u, c[q]);
            }
        }
        if (pick < 0) { pick = last < 0 ? 0 : last; u = 0.f; }
        node = node * 32 + pick;
    }
    // an empty tree (every priority zero) cannot be sampled proportionally: slot 0, weight 1 (callers check the ring is not empty)
    const float prob = total > 0.f ? __fdiv_rn(T.leaves[node], total) : 0.f;
    const float w = prob > 0.f ? powf((float)scalars[3] * prob, -beta) : 1.f;
    slots[k] = node;
    is_w[k] = w;
    // batch maximum: one atomic per wave, not per sample (order-independent, so still reproducible)
    float wm = w;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&scalars[2], __float_as_uint(wm));
}

static PerTree per_tree(const xq_replay* r) {
    PerTree T; memset(&T, 0, sizeof T);
    T.leaves = r->per.leaves; T.upper = r->per.upper; T.nlv = r->per.nlv;
    for (int lv = 0; lv < 8; ++lv) { T.n[lv] = r->per.n[lv]; T.off[lv] = (long long)r->per.off[lv]; }
    return T;
}

int replay_per_rebuild(xq_replay* r, int retire_start, int retire_count, hipStream_t on) {
    if (!r || !r->per.enabled) return fail(XQ_ERR_INVALID_ARGUMENT, "prioritized replay is not enabled on this ring");
    hipStream_t s = on ? on : r->stream;
    // the batch-maximum slot belongs to the last draw until its TD step has been queued (sample -> rebuild -> td_grads must
    // not divide by a zeroed maximum); the next draw then clears it itself
    const bool zero_wmax = !r->per.draw_unconsumed;
    if (!r->caller_orders) {
        // reads the priorities env steps and TD steps wrote on their own streams, retires slots in the same table, and snapshots the
        // maximum the next env steps read
        XQ_TRY(r->prio_res.write(s));
        // (the batch-maximum slot it may zero is read only by a prioritized TD step, which also WRITES priorities: the line above already
        // waits for it)
    }
    if (retire_count > 0) {      // slots the next collects overwrite: out of the tree before anything samples them
        hipLaunchKernelGGL(per_fill_kernel, dim3((retire_count + 255) / 256), dim3(256), 0, s, r->dev.prio, retire_start, retire_count,
                           r->dev.capacity, nullptr, 0.f);
        XQ_HIP(hipGetLastError());
    }
    PerTree T = per_tree(r);
    int n_counts = 0;
    if (T.nlv > 1) {
        const int blocks = (T.n[1] + 255) / 256;
        n_counts = blocks * 4;                                       // one count per wave of the level-1 grid
        hipLaunchKernelGGL(per_level_kernel, dim3(blocks), dim3(256), 0, s, r->dev.prio, T.upper + T.off[1], T.n[1],
                           r->per.wave_counts, r->per.leaves);
        XQ_HIP(hipGetLastError());
    } else {
        XQ_HIP(hipMemsetAsync(r->per.scalars + 3, 0, sizeof(unsigned), s));        // (a ring of <= 32 slots: counted by the sampler's callers)
        XQ_HIP(hipMemcpyAsync(r->per.leaves, r->dev.prio, 32 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(per_upper_kernel, dim3(1), dim3(1024), 0, s, T, r->per.scalars, n_counts ? r->per.wave_counts : nullptr, n_counts,
                       zero_wmax ? 1 : 0);
    XQ_HIP(hipGetLastError());
    r->per.wmax_clean = zero_wmax;
    return XQ_OK;
}

int replay_per_sample(xq_replay* r, int batch, hipStream_t on) {
    if (!r || !r->per.enabled) return fail(XQ_ERR_INVALID_ARGUMENT, "prioritized replay is not enabled on this ring");
    if (batch <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_replay_sample: batch must be > 0");
    hipStream_t s = on ? on : r->stream;
    if (batch > r->slots_cap) {
        XQ_HIP(hipDeviceSynchronize());
        if (r->slots_dev) XQ_HIP(hipFree(r->slots_dev));
        if (r->per.is_w) XQ_HIP(hipFree(r->per.is_w));
        r->per.is_w = nullptr;
        XQ_HIP(hipMalloc(&r->slots_dev, (size_t)batch * sizeof(int32_t)));
        r->slots_cap = batch;
    }
    if (!r->per.is_w) XQ_HIP(hipMalloc(&r->per.is_w, (size_t)r->slots_cap * sizeof(float)));
    if (!r->caller_orders) {
        XQ_TRY(r->draw.write(s));        // the TD step that still reads the previous list / weights / batch maximum (its own stream) first
    }
    if (!r->per.wmax_clean) XQ_HIP(hipMemsetAsync(r->per.scalars + 2, 0, sizeof(unsigned), s));     // a second draw from the same tree
    r->per.wmax_clean = false;
    hipLaunchKernelGGL(per_sample_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, per_tree(r), batch, (uint32_t)r->sample_calls,
                       (uint32_t)r->seed, (uint32_t)(r->seed >> 32), r->per.beta, r->per.scalars, r->slots_dev, r->per.is_w);
    XQ_HIP(hipGetLastError());
    r->sample_calls++;
    r->last_batch = batch;
    r->implicit = false;
    r->per.last_prioritized = true;
    r->per.draw_unconsumed = true;
    return XQ_OK;
}

// ---- a TD step that takes its minibatch from the ring (xq_dqn_td_grads_replay, on the Q-net's stream) ------------------------------
int replay_consumer_begin(xq_replay* r, hipStream_t consumer, bool listed, bool prioritized) {
    r->per.draw_unconsumed = false;
    if (r->caller_orders) return XQ_OK;
    XQ_TRY(r->contents.read(consumer));                      // behind the env steps that wrote the transitions
    if (listed) XQ_TRY(r->draw.read(consumer));              // behind the draw; the next draw waits for this step in turn
    if (prioritized) XQ_TRY(r->prio_res.write(consumer));    // it writes the TD-error priorities of its samples: the next rebuild waits
    return XQ_OK;
}
// ---- an env step that writes its transitions into the ring (xq_env_selfplay_step, on the env's stream) -----------------------------
int replay_writer_begin(xq_replay* r, hipStream_t writer) {
    if (r->caller_orders) return XQ_OK;
    XQ_TRY(r->contents.write(writer));                       // behind the TD steps that still read the slots it overwrites
    if (r->per.enabled) {
        XQ_TRY(r->prio_res.write(writer));                   // new transitions enter with the maximum the last rebuild snapshot (behind
                                                             // it), and the next rebuild reads their priorities (behind this step)
    }
    return XQ_OK;
}

int replay_sample_implicit(xq_replay* r, int batch, int start, int count) {
    if (!r || batch <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "replay_sample_implicit: batch must be > 0");
    if (count < 0) { start = 0; count = r->size; }
    if (count <= 0 || count > r->size || start < 0 || start >= r->dev.capacity)
        return fail(XQ_ERR_RUNTIME, "xq_replay_sample: buffer is empty");
    r->implicit = true;
    r->per.last_prioritized = false;
    r->per.draw_unconsumed = false;
    r->implicit_call = (uint32_t)r->sample_calls;
    r->implicit_size = count;
    r->implicit_start = start;
    r->sample_calls++;
    r->last_batch = batch;
    return XQ_OK;
}

}  // namespace xq

using namespace xq;

extern "C" {

static int replay_init(xq_replay* r, int capacity, uint64_t seed, void* hip_stream);

int xq_replay_create(int capacity, uint64_t seed, void* hip_stream, xq_replay** out) {
    if (!out || capacity <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_replay_create: capacity must be > 0");
    int c = 0;
    XQ_TRY(xq_device_count(&c));
    if (c == 0) return fail(XQ_ERR_NO_DEVICE, "no HIP device: libxqhip has no CPU fallback");
    xq_replay* r = new xq_replay();
    const int rc = replay_init(r, capacity, seed, hip_stream);
    if (rc != XQ_OK) { xq_replay_destroy(r); return rc; }   // a failed allocation must not leak the ones before it
    *out = r;
    return XQ_OK;
}

static int replay_init(xq_replay* r, int capacity, uint64_t seed, void* hip_stream) {
    r->seed = seed;
    if (hip_stream) r->stream = (hipStream_t)hip_stream;
    else { XQ_HIP(hipStreamCreate(&r->stream)); r->own_stream = true; }
    const size_t n = (size_t)capacity;
    r->dev.capacity = capacity;
    XQ_HIP(hipMalloc(&r->dev.boards, n * kBoardWords * sizeof(uint32_t)));
    XQ_HIP(hipMalloc(&r->dev.next_boards, n * kBoardWords * sizeof(uint32_t)));
    XQ_HIP(hipMalloc(&r->dev.action_to, n * sizeof(int32_t)));
    XQ_HIP(hipMalloc(&r->dev.reward, n * sizeof(float)));
    XQ_HIP(hipMalloc(&r->dev.done, n));
    XQ_HIP(hipMemsetAsync(r->dev.action_to, 0xFF, n * sizeof(int32_t), r->stream));   // -1: empty slot
    XQ_HIP(hipMemsetAsync(r->dev.boards, 0, n * kBoardWords * sizeof(uint32_t), r->stream));
    XQ_HIP(hipMemsetAsync(r->dev.next_boards, 0, n * kBoardWords * sizeof(uint32_t), r->stream));
    XQ_HIP(hipMemsetAsync(r->dev.reward, 0, n * sizeof(float), r->stream));
    XQ_HIP(hipMemsetAsync(r->dev.done, 0, n, r->stream));
    return XQ_OK;
}

int xq_replay_destroy(xq_replay* r) {
    if (!r) return XQ_OK;
    hipStreamSynchronize(r->stream);
    retire_stream(r->stream);        // synchronised above; unconditional: a caller-owned stream may be destroyed right after this call
    hipFree(r->dev.boards); hipFree(r->dev.next_boards); hipFree(r->dev.action_to); hipFree(r->dev.reward);
    hipFree(r->dev.done); hipFree(r->slots_dev);
    hipFree(r->dev.prio); hipFree(r->per.leaves); hipFree(r->per.upper); hipFree(r->per.scalars); hipFree(r->per.wave_counts); hipFree(r->per.is_w);
    if (r->own_stream) hipStreamDestroy(r->stream);
    delete r;
    return XQ_OK;
}

int xq_replay_size(xq_replay* r, int* size, int* capacity, uint64_t* total) {
    if (!r) return fail(XQ_ERR_INVALID_ARGUMENT, "null replay");
    if (size) *size = r->size;
    if (capacity) *capacity = r->dev.capacity;
    if (total) *total = r->total;
    return XQ_OK;
}

int xq_replay_push_host(xq_replay* r, int n, const uint8_t* boards90, const int32_t* action_to, const float* reward,
                        const uint8_t* done, const uint8_t* next_boards90) {
    if (!r || n < 0 || !boards90 || !action_to || !reward || !done || !next_boards90)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_replay_push_host: null pointer");
    if (n > r->dev.capacity) return fail(XQ_ERR_INVALID_ARGUMENT, "push of %d exceeds capacity %d", n, r->dev.capacity);
    for (size_t i = 0; i < (size_t)n * 90; ++i)       // code 15 would index one-hot plane 14 of 14 in the layer-0 kernels
        if (boards90[i] > 14 || next_boards90[i] > 14) return fail(XQ_ERR_INVALID_ARGUMENT, "piece code > 14");
    uint32_t w[kBoardWords], nw[kBoardWords];
    XQ_HIP(hipDeviceSynchronize());              // host-side writes: nothing may still be reading or writing the ring on any stream
    r->contents.host_synchronised(); r->prio_res.host_synchronised();
    for (int i = 0; i < n; ++i) {
        const int slot = r->write_pos;
        pack_board(boards90 + (size_t)i * 90, w);
        pack_board(next_boards90 + (size_t)i * 90, nw);
        XQ_HIP(hipMemcpy(r->dev.boards + (size_t)slot * kBoardWords, w, sizeof w, hipMemcpyHostToDevice));
        XQ_HIP(hipMemcpy(r->dev.next_boards + (size_t)slot * kBoardWords, nw, sizeof nw, hipMemcpyHostToDevice));
        XQ_HIP(hipMemcpy(r->dev.action_to + slot, action_to + i, sizeof(int32_t), hipMemcpyHostToDevice));
        XQ_HIP(hipMemcpy(r->dev.reward + slot, reward + i, sizeof(float), hipMemcpyHostToDevice));
        XQ_HIP(hipMemcpy(r->dev.done + slot, done + i, 1, hipMemcpyHostToDevice));
        if (r->per.enabled) {       // a new transition enters with the largest priority assigned so far (0 for an empty one)
            hipLaunchKernelGGL(per_fill_kernel, dim3(1), dim3(64), 0, r->stream, r->dev.prio, slot, 1, r->dev.capacity,
                               action_to[i] >= 0 ? r->per.scalars + 0 : nullptr, 0.f);
            XQ_HIP(hipGetLastError());
        }
        r->write_pos = (r->write_pos + 1) % r->dev.capacity;
        if (r->size < r->dev.capacity) r->size++;
        r->total++;
    }
    if (r->per.enabled) XQ_HIP(hipStreamSynchronize(r->stream));
    return XQ_OK;
}

// ---- prioritized replay: C ABI -------------------------------------------------------------------------------------------
int xq_replay_enable_per(xq_replay* r, double alpha, double beta, double eps) {
    if (!r) return fail(XQ_ERR_INVALID_ARGUMENT, "null replay");
    if (!(alpha >= 0) || !(beta >= 0) || !(eps > 0)) return fail(XQ_ERR_INVALID_ARGUMENT, "prioritized replay needs alpha >= 0, beta >= 0, eps > 0");
    if (r->per.enabled) return fail(XQ_ERR_RUNTIME, "prioritized replay is already enabled");
    xq_replay::Per& P = r->per;
    P.alpha = (float)alpha; P.beta = (float)beta; P.eps = (float)eps;
    int cnt = r->dev.capacity, lv = 0;
    size_t upper = 0;
    for (;;) {                                   // level sizes: 32 children per node up to a single root
        if (lv >= 8) return fail(XQ_ERR_INVALID_ARGUMENT, "capacity too large for the priority tree");
        P.n[lv] = cnt; P.p[lv] = round_up(cnt, 32);
        if (lv >= 1) { P.off[lv] = upper; upper += (size_t)P.p[lv]; }
        ++lv;
        if (cnt <= 1) break;
        cnt = P.p[lv - 1] / 32;
    }
    P.nlv = lv;
    XQ_HIP(hipStreamSynchronize(r->stream));
    XQ_HIP(hipMalloc(&r->dev.prio, (size_t)P.p[0] * sizeof(float)));
    XQ_HIP(hipMemset(r->dev.prio, 0, (size_t)P.p[0] * sizeof(float)));
    XQ_HIP(hipMalloc(&P.leaves, (size_t)P.p[0] * sizeof(float)));
    XQ_HIP(hipMemset(P.leaves, 0, (size_t)P.p[0] * sizeof(float)));
    XQ_HIP(hipMalloc(&P.upper, std::max<size_t>(upper, 32) * sizeof(float)));
    XQ_HIP(hipMemset(P.upper, 0, std::max<size_t>(upper, 32) * sizeof(float)));
    XQ_HIP(hipMalloc(&P.scalars, 4 * sizeof(unsigned)));
    XQ_HIP(hipMalloc(&P.wave_counts, (size_t)((P.n[1] + 255) / 256 * 4 + 4) * sizeof(unsigned)));
    const float one = 1.0f;                      // initial maximum priority (Schaul et al.: new transitions get the maximum, 1 at start)
    unsigned init[4];
    memcpy(&init[0], &one, 4); init[1] = init[0]; init[2] = 0; init[3] = 0;
    XQ_HIP(hipMemcpy(P.scalars, init, sizeof init, hipMemcpyHostToDevice));
    r->dev.pmax_snap = P.scalars + 1;
    P.enabled = true;
    return XQ_OK;
}

int xq_replay_set_priorities(xq_replay* r, int first, int n, const float* prio_host) {
    if (!r || !r->per.enabled || !prio_host || first < 0 || n < 0 || first + n > r->dev.capacity)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_replay_set_priorities: bad argument (is prioritized replay enabled?)");
    for (int i = 0; i < n; ++i)
        if (!(prio_host[i] >= 0.f)) return fail(XQ_ERR_INVALID_ARGUMENT, "priorities must be >= 0");
    XQ_HIP(hipDeviceSynchronize());              // (a TD step or an env step on another stream may be writing the table)
    r->prio_res.host_synchronised();
    XQ_HIP(hipMemcpy(r->dev.prio + first, prio_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    float mx = 0.f;
    for (int i = 0; i < n; ++i) mx = std::max(mx, prio_host[i]);
    unsigned cur = 0;
    XQ_HIP(hipMemcpy(&cur, r->per.scalars, 4, hipMemcpyDeviceToHost));
    float curf; memcpy(&curf, &cur, 4);
    if (mx > curf) { memcpy(&cur, &mx, 4); XQ_HIP(hipMemcpy(r->per.scalars, &cur, 4, hipMemcpyHostToDevice)); }
    return XQ_OK;
}

int xq_replay_get_priorities(xq_replay* r, int first, int n, float* prio_host) {
    if (!r || !r->per.enabled || !prio_host || first < 0 || n < 0 || first + n > r->dev.capacity)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_replay_get_priorities: bad argument (is prioritized replay enabled?)");
    XQ_HIP(hipDeviceSynchronize());
    XQ_HIP(hipMemcpy(prio_host, r->dev.prio + first, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return XQ_OK;
}

int xq_replay_per_rebuild(xq_replay* r, int retire_start, int retire_count) {
    if (!r) return fail(XQ_ERR_INVALID_ARGUMENT, "null replay");
    if (retire_count < 0 || retire_count > r->dev.capacity || retire_start < 0 || retire_start >= r->dev.capacity)
        return fail(XQ_ERR_INVALID_ARGUMENT, "bad retire window");
    return replay_per_rebuild(r, retire_start, retire_count, nullptr);
}

int xq_replay_per_stats(xq_replay* r, float* total, float* max_priority, int* n_eligible) {
    if (!r || !r->per.enabled) return fail(XQ_ERR_INVALID_ARGUMENT, "prioritized replay is not enabled on this ring");
    XQ_HIP(hipDeviceSynchronize());
    unsigned sc[4];
    XQ_HIP(hipMemcpy(sc, r->per.scalars, sizeof sc, hipMemcpyDeviceToHost));
    if (total) {
        const float* root = r->per.nlv > 1 ? r->per.upper + r->per.off[r->per.nlv - 1] : r->per.leaves;
        XQ_HIP(hipMemcpy(total, root, sizeof(float), hipMemcpyDeviceToHost));
    }
    if (max_priority) memcpy(max_priority, &sc[0], 4);
    if (n_eligible) *n_eligible = (int)sc[3];
    return XQ_OK;
}

int xq_replay_sample_prioritized(xq_replay* r, int batch, int32_t* slots_host, float* weights_host) {
    if (!r) return fail(XQ_ERR_INVALID_ARGUMENT, "null replay");
    XQ_TRY(replay_per_sample(r, batch, nullptr));
    if (slots_host || weights_host) {
        XQ_HIP(hipStreamSynchronize(r->stream));
        if (slots_host) XQ_HIP(hipMemcpy(slots_host, r->slots_dev, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (weights_host) {
            XQ_HIP(hipMemcpy(weights_host, r->per.is_w, (size_t)batch * sizeof(float), hipMemcpyDeviceToHost));
            unsigned wm = 0;
            XQ_HIP(hipMemcpy(&wm, r->per.scalars + 2, 4, hipMemcpyDeviceToHost));
            float wmax; memcpy(&wmax, &wm, 4);
            for (int i = 0; i < batch; ++i) weights_host[i] /= wmax;        // what td_grads applies: w_i / max_batch w
        }
    }
    return XQ_OK;
}

int xq_replay_sample(xq_replay* r, int batch, int32_t* slots_host) {
    if (!r) return fail(XQ_ERR_INVALID_ARGUMENT, "null replay");
    return xq_replay_sample_window(r, batch, 0, r->size, slots_host);
}

int xq_replay_sample_window(xq_replay* r, int batch, int start, int count, int32_t* slots_host) {
    if (!r || batch <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_replay_sample: batch must be > 0");
    if (r->size <= 0 || count <= 0) return fail(XQ_ERR_RUNTIME, "xq_replay_sample: buffer is empty");
    if (count > r->size || start < 0 || start >= r->dev.capacity)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_replay_sample_window: window (%d, %d) outside the %d filled slots", start, count, r->size);
    if (batch > r->slots_cap) {
        if (r->slots_dev) { XQ_HIP(hipStreamSynchronize(r->stream)); XQ_HIP(hipFree(r->slots_dev)); }
        XQ_HIP(hipMalloc(&r->slots_dev, (size_t)batch * sizeof(int32_t)));
        r->slots_cap = batch;
    }
    if (!r->caller_orders) XQ_TRY(r->draw.write(r->stream));     // the TD step that still reads the previous list (its own stream) first
    hipLaunchKernelGGL(replay_sample_kernel, dim3((batch + 255) / 256), dim3(256), 0, r->stream, r->slots_dev, batch,
                       (uint32_t)start, (uint32_t)count, (uint32_t)r->dev.capacity, (uint32_t)r->sample_calls, (uint32_t)r->seed, (uint32_t)(r->seed >> 32));
    XQ_HIP(hipGetLastError());
    r->sample_calls++;
    r->last_batch = batch;
    r->implicit = false;
    r->per.last_prioritized = false;
    r->per.draw_unconsumed = false;
    if (slots_host) {
        XQ_HIP(hipMemcpyAsync(slots_host, r->slots_dev, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToHost, r->stream));
        XQ_HIP(hipStreamSynchronize(r->stream));
    }
    return XQ_OK;
}

int xq_replay_get(xq_replay* r, int slot, uint8_t* board90, int32_t* action_to, float* reward, uint8_t* done,
                  uint8_t* next_board90) {
    if (!r || slot < 0 || slot >= r->dev.capacity) return fail(XQ_ERR_INVALID_ARGUMENT, "bad slot");
    uint32_t w[kBoardWords];
    if (!r->caller_orders) XQ_TRY(r->contents.read(r->stream));      // behind an env step that wrote the slot on a stream of its own
    else XQ_HIP(hipDeviceSynchronize());
    XQ_HIP(hipStreamSynchronize(r->stream));
    if (board90) {
        XQ_HIP(hipMemcpy(w, r->dev.boards + (size_t)slot * kBoardWords, sizeof w, hipMemcpyDeviceToHost));
        unpack_board(w, board90);
    }
    if (next_board90) {
        XQ_HIP(hipMemcpy(w, r->dev.next_boards + (size_t)slot * kBoardWords, sizeof w, hipMemcpyDeviceToHost));
        unpack_board(w, next_board90);
    }
    if (action_to) XQ_HIP(hipMemcpy(action_to, r->dev.action_to + slot, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (reward) XQ_HIP(hipMemcpy(reward, r->dev.reward + slot, sizeof(float), hipMemcpyDeviceToHost));
    if (done) XQ_HIP(hipMemcpy(done, r->dev.done + slot, 1, hipMemcpyDeviceToHost));
    return XQ_OK;
}

}  // extern "C"
