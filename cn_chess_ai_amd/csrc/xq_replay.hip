// xq_replay.hip — ReplayBuffer: ring of (state, action.to, reward, nextState, done) in HBM.
//
// The reference has no replay buffer; the element is the argument list of the never-called
// DQN::train(state, action, reward, nextState, done) (reference dqn.cpp:157-172).  States are stored as packed boards
// (12 x u32 = 48 B) instead of 1260 doubles (10 080 B): 1 M transitions = 105 MB instead of 20 GB.
#include "xq_internal.h"

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

int replay_sample_implicit(xq_replay* r, int batch, int start, int count) {
    if (!r || batch <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "replay_sample_implicit: batch must be > 0");
    if (count < 0) { start = 0; count = r->size; }
    if (count <= 0 || count > r->size || start < 0 || start >= r->dev.capacity)
        return fail(XQ_ERR_RUNTIME, "xq_replay_sample: buffer is empty");
    r->implicit = true;
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
    hipFree(r->dev.boards); hipFree(r->dev.next_boards); hipFree(r->dev.action_to); hipFree(r->dev.reward);
    hipFree(r->dev.done); hipFree(r->slots_dev);
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
    XQ_HIP(hipStreamSynchronize(r->stream));
    for (int i = 0; i < n; ++i) {
        const int slot = r->write_pos;
        pack_board(boards90 + (size_t)i * 90, w);
        pack_board(next_boards90 + (size_t)i * 90, nw);
        XQ_HIP(hipMemcpy(r->dev.boards + (size_t)slot * kBoardWords, w, sizeof w, hipMemcpyHostToDevice));
        XQ_HIP(hipMemcpy(r->dev.next_boards + (size_t)slot * kBoardWords, nw, sizeof nw, hipMemcpyHostToDevice));
        XQ_HIP(hipMemcpy(r->dev.action_to + slot, action_to + i, sizeof(int32_t), hipMemcpyHostToDevice));
        XQ_HIP(hipMemcpy(r->dev.reward + slot, reward + i, sizeof(float), hipMemcpyHostToDevice));
        XQ_HIP(hipMemcpy(r->dev.done + slot, done + i, 1, hipMemcpyHostToDevice));
        r->write_pos = (r->write_pos + 1) % r->dev.capacity;
        if (r->size < r->dev.capacity) r->size++;
        r->total++;
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
    hipLaunchKernelGGL(replay_sample_kernel, dim3((batch + 255) / 256), dim3(256), 0, r->stream, r->slots_dev, batch,
                       (uint32_t)start, (uint32_t)count, (uint32_t)r->dev.capacity, (uint32_t)r->sample_calls, (uint32_t)r->seed, (uint32_t)(r->seed >> 32));
    XQ_HIP(hipGetLastError());
    r->sample_calls++;
    r->last_batch = batch;
    r->implicit = false;
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
