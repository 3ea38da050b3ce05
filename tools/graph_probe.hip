// tools/graph_probe.hip — does a captured hipGraph carry the fork / join structure of the TD step's gradient tail more cheaply than
// event records and waits on eagerly launched streams?  (diagnostic, not part of libxqhip)
//
// Model of the tail (durations as in profiles/r03_c_step_timeline_config2.txt): critical stream K1 10 us -> K2 17 -> K3 40 -> K4 7,
// side stream S1 15 -> S2 23 behind K1 and S3 10 behind K2, K4 behind the side stream too.  Kernels are register spin loops on 256
// blocks (no memory traffic), so only the synchronisation differs between the variants:
//   eager : hipEventRecord / hipStreamWaitEvent exactly as xq_dqn.hip issues them, `reps` iterations queued back to back
//   graph : the same iteration captured once from the two streams (hipStreamBeginCapture), instantiated, hipGraphLaunch'ed `reps` times
//   chain : all seven kernels on ONE stream (no fork at all): what the structure costs if nothing overlaps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d: %s\n", hipGetErrorString(e_), __LINE__, #x); exit(1); } } while (0)

__global__ void spin_kernel(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
}

int main() {
    hipStream_t M, S;
    CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    float* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
    hipEvent_t t0, t1, ef, ed, ej;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ed, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    // calibrate: iterations per microsecond of a 256-block spin kernel
    auto run = [&](int it, hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, p, it); };
    float ms = 0;
    run(100000, M); CK(hipStreamSynchronize(M));
    CK(hipEventRecord(t0, M)); run(400000, M); CK(hipEventRecord(t1, M)); CK(hipStreamSynchronize(M)); CK(hipEventElapsedTime(&ms, t0, t1));
    const double per_us = 400000.0 / (ms * 1e3);
    auto us = [&](double u) { return (int)(u * per_us); };
    printf("calibration: %.0f spin iterations per us\n", per_us);
    const int k1 = us(10), k2 = us(17), k3 = us(40), k4 = us(7), s1 = us(15), s2 = us(23), s3 = us(10);
    const int reps = 300;
    auto iteration = [&]() {
        run(k1, M);
        CK(hipEventRecord(ef, M)); CK(hipStreamWaitEvent(S, ef, 0));
        run(k2, M);
        CK(hipEventRecord(ed, M));
        run(k3, M);
        run(s1, S); run(s2, S);
        CK(hipStreamWaitEvent(S, ed, 0));
        run(s3, S);
        CK(hipEventRecord(ej, S)); CK(hipStreamWaitEvent(M, ej, 0));
        run(k4, M);
    };
    for (int round = 0; round < 3; ++round) {
        // eager
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, M));
        for (int r = 0; r < reps; ++r) iteration();
        CK(hipEventRecord(t1, M)); CK(hipStreamSynchronize(M)); CK(hipStreamSynchronize(S)); CK(hipEventElapsedTime(&ms, t0, t1));
        const float eager = 1e3f * ms / reps;
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(M, hipStreamCaptureModeGlobal));
        iteration();
        CK(hipStreamEndCapture(M, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, M));
        CK(hipStreamSynchronize(M));
        CK(hipEventRecord(t0, M));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, M));
        CK(hipEventRecord(t1, M)); CK(hipStreamSynchronize(M)); CK(hipEventElapsedTime(&ms, t0, t1));
        const float graph = 1e3f * ms / reps;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        // one stream
        CK(hipEventRecord(t0, M));
        for (int r = 0; r < reps; ++r) { run(k1, M); run(k2, M); run(k3, M); run(s1, M); run(s2, M); run(s3, M); run(k4, M); }
        CK(hipEventRecord(t1, M)); CK(hipStreamSynchronize(M)); CK(hipEventElapsedTime(&ms, t0, t1));
        const float chain = 1e3f * ms / reps;
        printf("round %d: critical path 74 us of kernels; eager (events) %.1f us per iteration, graph replay %.1f us, one stream (122 us of kernels) %.1f us\n",
               round, eager, graph, chain);
    }
    return 0;
}
