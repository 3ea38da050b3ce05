// xq_comm.hip — the one exchange step of the data-parallel loop: sum all-reduce of the TD gradient buffer over RCCL / xGMI.
//
// The reference is single-process, single-GPU; multi-GPU is build-defined: games are
// sharded by contiguous game-id ranges, one process per GPU, and every update all-reduces the compact gradient buffer
// (1.65 MB for 1260-256-256-8100), then applies identical SGD on every replica.  RCCL is opened with dlopen on first use
// (librccl is > 500 MB and a single-GPU user never needs it); inside a process that has torch loaded this resolves to the
// very librccl.so.1 torch bundles, so both share one HIP runtime.
//
// Bucketing (xq_dqn_set_comm): the gradient chains of a TD step end on two streams — the side stream finishes the
// output-layer / hidden-layer / bias segments (0.36 MB, one contiguous range behind the layer-0 segment) while the handle's
// stream is still inside the layer-0 segmented sum (1.29 MB, the last and largest piece).  Each range is all-reduced ON THE
// STREAM OF ITS PRODUCER, right behind it (the small bucket is issued first, so RCCL's per-communicator ordering never holds
// it behind the late one); only the layer-0 bucket is exposed.  There is no communicator stream in the TD step: a process
// with a fifth busy stream shares hardware queues between streams that wait for each other.  The
// communicator's own stream exists only for the stand-alone calls (xq_comm_sum_u64, xq_comm_allreduce without a stream) and
// is created on their first use.
#include "xq_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <thread>

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* error = nullptr;
};

RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) { a.error = "librccl.so.1 not found (dlopen)"; return a; }
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString)
            a.error = "librccl.so.1 lacks an expected nccl* symbol";
        return a;
    }();
    return api;
}

}  // namespace

struct xq_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;          // stand-alone collectives only (created on first use); the TD step uses its producers' streams
    uint64_t collectives = 0, floats = 0;  // issued so far (tests, bench)
    unsigned long long* scalar = nullptr;  // device scratch of xq_comm_sum_u64
};

#define XQ_RCCL(call)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (call);                                                                                  \
        if (r_ != ncclSuccess)                                                                                     \
            return ::xq::fail(XQ_ERR_RUNTIME, "RCCL error: %s at %s:%d (%s)", rccl().GetErrorString(r_), __FILE__, __LINE__, #call); \
    } while (0)

namespace xq {

int comm_allreduce_on(xq_comm* c, float* buf, size_t n_floats, hipStream_t stream) {
    if (n_floats == 0) return XQ_OK;
    XQ_RCCL(rccl().AllReduce(buf, buf, n_floats, ncclFloat32, ncclSum, c->comm, stream));
    c->collectives += 1;
    c->floats += n_floats;
    return XQ_OK;
}
int comm_world(const xq_comm* c) { return c->world; }
static int comm_own_stream(xq_comm* c) {
    if (!c->stream) XQ_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    return XQ_OK;
}

}  // namespace xq

using namespace xq;

extern "C" {

int xq_comm_unique_id(uint8_t* id128) {
    if (!id128) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    if (rccl().error) return fail(XQ_ERR_RUNTIME, "%s", rccl().error);
    static_assert(sizeof(ncclUniqueId) == XQ_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    XQ_RCCL(rccl().GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return XQ_OK;
}

int xq_comm_create(int rank, int world, const uint8_t* id128, xq_comm** out) {
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_comm_create: bad rank/world");
    int ndev = 0;
    XQ_TRY(xq_device_count(&ndev));
    if (ndev == 0) return fail(XQ_ERR_NO_DEVICE, "no HIP device: libxqhip has no CPU fallback");
    if (rccl().error) return fail(XQ_ERR_RUNTIME, "%s", rccl().error);
    xq_comm* c = new xq_comm();
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);     // binds the CURRENT device (xq_set_device)
    if (r != ncclSuccess) {
        delete c;
        return fail(XQ_ERR_RUNTIME, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl().GetErrorString(r));
    }
    hipError_t e = hipMalloc(&c->scalar, sizeof(unsigned long long));
    if (e != hipSuccess) { xq_comm_destroy(c); return fail(XQ_ERR_RUNTIME, "HIP error: %s (xq_comm_create)", hipGetErrorString(e)); }
    *out = c;
    return XQ_OK;
}

// Rendezvous through a file both sides can see (one node: any local path): rank 0 writes the unique id to `path`.tmp and
// renames it into place; the other ranks poll for it.  For callers without a launcher that can ship 128 bytes (the C++ facade).
int xq_comm_create_from_file(int rank, int world, const char* path, double timeout_s, xq_comm** out) {
    if (!path || !out) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    uint8_t id[XQ_COMM_ID_BYTES];
    if (rank == 0) {
        // a file left over from an earlier run would hand the other ranks a dead id (they may read it before this rank replaces it)
        if (FILE* old = fopen(path, "rb")) { fclose(old); return fail(XQ_ERR_IO, "rendezvous file %s already exists: use a fresh path per run", path); }
        XQ_TRY(xq_comm_unique_id(id));
        const std::string tmp = std::string(path) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f) return fail(XQ_ERR_IO, "cannot write %s", tmp.c_str());
        const bool ok = fwrite(id, 1, sizeof id, f) == sizeof id;
        fclose(f);
        if (!ok || rename(tmp.c_str(), path) != 0) return fail(XQ_ERR_IO, "cannot publish %s", path);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            FILE* f = fopen(path, "rb");
            if (f) {
                const size_t n = fread(id, 1, sizeof id, f);
                fclose(f);
                if (n == sizeof id) break;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                return fail(XQ_ERR_IO, "rank %d: no unique id at %s after %.0f s", rank, path, timeout_s);
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    const int rc = xq_comm_create(rank, world, id, out);
    // ncclCommInitRank returns once every rank has joined, i.e. after every rank has read the id: rank 0 takes the file away
    // again, so the same path serves the next run (a leftover from a crashed run is still refused above — never a dead id)
    if (rank == 0) remove(path);
    return rc;
}

int xq_comm_destroy(xq_comm* c) {
    if (!c) return XQ_OK;
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->scalar) hipFree(c->scalar);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return XQ_OK;
}

int xq_comm_info(const xq_comm* c, int* rank, int* world, uint64_t* collectives, uint64_t* floats) {
    if (!c) return fail(XQ_ERR_INVALID_ARGUMENT, "null comm");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (collectives) *collectives = c->collectives;
    if (floats) *floats = c->floats;
    return XQ_OK;
}

int xq_comm_sum_u64(xq_comm* c, uint64_t* inout_host) {
    if (!c || !inout_host) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    unsigned long long v = *inout_host;
    XQ_TRY(comm_own_stream(c));
    XQ_HIP(hipMemcpyAsync(c->scalar, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
    XQ_RCCL(rccl().AllReduce(c->scalar, c->scalar, 1, ncclUint64, ncclSum, c->comm, c->stream));
    XQ_HIP(hipMemcpyAsync(&v, c->scalar, sizeof v, hipMemcpyDeviceToHost, c->stream));
    XQ_HIP(hipStreamSynchronize(c->stream));
    *inout_host = v;
    return XQ_OK;
}

int xq_comm_allreduce(xq_comm* c, float* buf_dev, size_t n_floats, void* hip_stream) {
    if (!c || !buf_dev) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    if (!hip_stream) XQ_TRY(comm_own_stream(c));
    return comm_allreduce_on(c, buf_dev, n_floats, hip_stream ? (hipStream_t)hip_stream : c->stream);
}

}  // extern "C"
