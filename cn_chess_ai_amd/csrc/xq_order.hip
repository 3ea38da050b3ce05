// xq_order.hip — ordering of device data that handles on different HIP streams hand to each other (SharedResource, xq_internal.h),
// the stream utilities of the C ABI, and the two diagnostics the ordering tests use (a delay kernel, a switch per ordering class).
//
// No upstream analogue: the reference runs one stream and calls cudaDeviceSynchronize() after every launch (dqn.cu:233-236, 359-362).
#include "xq_internal.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace xq {

namespace {
std::mutex& registry_mutex() { static std::mutex m; return m; }
std::vector<SharedResource*>& registry() { static std::vector<SharedResource*> v; return v; }
}  // namespace

SharedResource::SharedResource(unsigned c) : cls(c) {
    std::lock_guard<std::mutex> lock(registry_mutex());
    registry().push_back(this);
}
SharedResource::~SharedResource() {
    {
        std::lock_guard<std::mutex> lock(registry_mutex());
        auto& v = registry();
        v.erase(std::remove(v.begin(), v.end(), this), v.end());
    }
    if (ev) (void)hipEventDestroy(ev);
}

// (callers hold registry_mutex())
int SharedResource::order_behind(hipStream_t waiter, hipStream_t producer) {
    if (waiter == producer || !(order_mask() & cls)) return XQ_OK;
    if (!ev) XQ_HIP(hipEventCreateWithFlags(&ev, stream_event_flags()));
    // lazily: behind everything the producer's stream holds by now.  A record that fails means the producer's stream is gone (a caller
    // destroyed a stream it had lent to a handle without xq_stream_destroy / the handle's destroy, which strike it from here): a
    // destroyed stream has finished its work, so its entry is dropped and there is nothing left to wait for — the error is not kept.
    if (hipEventRecord(ev, producer) != hipSuccess) {
        (void)hipGetLastError();
        forget(producer);
        return XQ_OK;
    }
    XQ_HIP(hipStreamWaitEvent(waiter, ev, 0));
    return XQ_OK;
}

// read() / write() / host_synchronised() take the registry's mutex: retire_stream() — any handle or stream being destroyed on another
// thread — edits the same fields of EVERY resource.  Uncontended in the trainer (one thread): ~20 ns per access.
int SharedResource::read(hipStream_t s) {
    std::lock_guard<std::mutex> lock(registry_mutex());
    for (int i = 0; i < n_readers; ++i)
        if (readers[i] == s) return XQ_OK;           // already ordered behind the last write
    if (has_writer) XQ_TRY(order_behind(s, writer));
    if (n_readers == 4) {                            // a fifth reading stream takes the place of the fourth and waits for it, so that a
        XQ_TRY(order_behind(s, readers[3]));         // later writer that waits for s is behind both
        if (n_readers == 4) readers[3] = s; else readers[n_readers++] = s;     // (order_behind may have dropped a dead reader)
    } else {
        readers[n_readers++] = s;
    }
    return XQ_OK;
}

int SharedResource::write(hipStream_t s) {
    std::lock_guard<std::mutex> lock(registry_mutex());
    bool waited = false;
    hipStream_t rd[4];
    const int nr = n_readers;
    for (int i = 0; i < nr; ++i) rd[i] = readers[i];                  // (a copy: order_behind may strike a dead stream from readers[])
    for (int i = 0; i < nr; ++i)
        if (rd[i] != s) { XQ_TRY(order_behind(s, rd[i])); waited = true; }
    // (every reader is itself behind the last writer: waiting for one of them covers it)
    if (has_writer && writer != s && !waited) XQ_TRY(order_behind(s, writer));
    has_writer = true; writer = s; n_readers = 0;
    return XQ_OK;
}

void SharedResource::host_synchronised() {
    std::lock_guard<std::mutex> lock(registry_mutex());
    has_writer = false; n_readers = 0;
}

void SharedResource::forget(hipStream_t s) {
    if (has_writer && writer == s) has_writer = false;
    int k = 0;
    for (int i = 0; i < n_readers; ++i)
        if (readers[i] != s) readers[k++] = readers[i];
    n_readers = k;
}

void retire_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lock(registry_mutex());
    for (SharedResource* r : registry()) r->forget(s);
}

// spins until `ticks` of the 100 MHz wall clock have passed (s_memrealtime): bounded by construction, one wave
__global__ void delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// waits until the host raises *flag (xq_debug_gate_release) — or `ticks` of the 100 MHz wall clock have passed: an exit every wave
// reaches whatever the host does
__global__ void gate_kernel(const unsigned* flag, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// the diagnostics below switch synchronisation off / stall streams: refused unless the process asked for them
static int debug_api_enabled(const char* what) {
    const char* e = getenv("XQ_DEBUG_API");
    if (e && e[0] == '1') return XQ_OK;
    return fail(XQ_ERR_INVALID_ARGUMENT, "%s: the debug entry points are disabled (set XQ_DEBUG_API=1 in the environment of a test process)", what);
}

}  // namespace xq

using namespace xq;

struct xq_debug_gate { unsigned* host = nullptr; unsigned* dev = nullptr; };

extern "C" {

int xq_stream_wait_stream(void* waiting_stream, void* producer_stream) {
    if (waiting_stream == producer_stream) return XQ_OK;
    hipEvent_t ev = nullptr;
    XQ_HIP(hipEventCreateWithFlags(&ev, stream_event_flags()));
    hipError_t e = hipEventRecord(ev, (hipStream_t)producer_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiting_stream, ev, 0);
    (void)hipEventDestroy(ev);                       // released once the wait has passed
    if (e != hipSuccess) return fail(XQ_ERR_RUNTIME, "xq_stream_wait_stream: %s", hipGetErrorString(e));
    return XQ_OK;
}

int xq_stream_create(int priority, int nonblocking, void** out) {
    if (!out || priority < -1 || priority > 1) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_stream_create: priority -1 (urgent), 0 or 1 (background)");
    int least = 0, greatest = 0;                     // numerically: greatest priority = the smaller number
    XQ_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int p = priority < 0 ? greatest : priority > 0 ? least : (least + greatest) / 2;
    hipStream_t s = nullptr;
    XQ_HIP(hipStreamCreateWithPriority(&s, nonblocking ? hipStreamNonBlocking : hipStreamDefault, p));
    *out = (void*)s;
    return XQ_OK;
}

int xq_stream_destroy(void* hip_stream) {
    if (!hip_stream) return XQ_OK;
    XQ_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    retire_stream((hipStream_t)hip_stream);          // no handle's bookkeeping may name it afterwards
    XQ_HIP(hipStreamDestroy((hipStream_t)hip_stream));
    return XQ_OK;
}

int xq_stream_query(void* hip_stream, int* idle) {
    if (!idle) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    const hipError_t e = hipStreamQuery((hipStream_t)hip_stream);
    if (e != hipSuccess && e != hipErrorNotReady) return fail(XQ_ERR_RUNTIME, "xq_stream_query: %s", hipGetErrorString(e));
    *idle = e == hipSuccess ? 1 : 0;
    return XQ_OK;
}

int xq_debug_stream_gate(void* hip_stream, int timeout_ms, void** gate_out) {
    XQ_TRY(debug_api_enabled("xq_debug_stream_gate"));
    if (!gate_out || timeout_ms < 1 || timeout_ms > 5000) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_debug_stream_gate: timeout 1..5000 ms");
    xq_debug_gate* g = new xq_debug_gate();
    if (hipHostMalloc((void**)&g->host, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) { delete g; return fail(XQ_ERR_RUNTIME, "hipHostMalloc failed"); }
    *g->host = 0u;
    if (hipHostGetDevicePointer((void**)&g->dev, g->host, 0) != hipSuccess) { (void)hipHostFree(g->host); delete g; return fail(XQ_ERR_RUNTIME, "hipHostGetDevicePointer failed"); }
    int rate_khz = 0;
    unsigned long long per_ms = 100000;              // s_memrealtime: 100 MHz on gfx9
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0) == hipSuccess && rate_khz > 0) per_ms = (unsigned long long)rate_khz;
    hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, g->dev, per_ms * (unsigned long long)timeout_ms);
    if (hipGetLastError() != hipSuccess) { (void)hipHostFree(g->host); delete g; return fail(XQ_ERR_RUNTIME, "gate launch failed"); }
    *gate_out = g;
    return XQ_OK;
}

int xq_debug_gate_release(void* gate) {
    if (!gate) return fail(XQ_ERR_INVALID_ARGUMENT, "null gate");
    __atomic_store_n(((xq_debug_gate*)gate)->host, 1u, __ATOMIC_RELEASE);
    return XQ_OK;
}

int xq_debug_gate_destroy(void* gate) {             // after the gated stream has been synchronised
    if (!gate) return XQ_OK;
    xq_debug_gate* g = (xq_debug_gate*)gate;
    __atomic_store_n(g->host, 1u, __ATOMIC_RELEASE);
    (void)hipHostFree(g->host);
    delete g;
    return XQ_OK;
}

int xq_debug_stream_delay(void* hip_stream, int microseconds) {
    XQ_TRY(debug_api_enabled("xq_debug_stream_delay"));
    if (microseconds < 0 || microseconds > 200000) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_debug_stream_delay: 0..200000 us");
    if (microseconds == 0) return XQ_OK;
    int rate_khz = 0;
    unsigned long long per_us = 100;                 // s_memrealtime: 100 MHz on gfx9
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0) == hipSuccess && rate_khz > 0)
        per_us = std::max(1ull, (unsigned long long)rate_khz / 1000ull);
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, per_us * (unsigned long long)microseconds);
    XQ_HIP(hipGetLastError());
    return XQ_OK;
}

int xq_debug_set_stream_ordering(unsigned mask) {
    if ((mask & ORD_ALL) != ORD_ALL) XQ_TRY(debug_api_enabled("xq_debug_set_stream_ordering"));      // switching ordering back ON is always allowed
    order_mask() = mask & ORD_ALL;
    return XQ_OK;
}

}  // extern "C"

// the stream a handle runs on (its own when it was created with NULL): what a caller needs for xq_stream_wait_stream
extern "C" int xq_env_stream(const xq_env* e, void** s) {
    if (!e || !s) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    *s = (void*)e->stream;
    return XQ_OK;
}
extern "C" int xq_replay_stream(const xq_replay* r, void** s) {
    if (!r || !s) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    *s = (void*)r->stream;
    return XQ_OK;
}
